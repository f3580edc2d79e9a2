"""Receiver side of the channel on the GPU: host mirror of the reference's
function-level API over the C ABI of include/ssf.h (ssf_fir_filter, ssf_delay_signal, ssf_decimate,
ssf_rx_run).  numpy in, numpy out, same names / argument meaning / error behaviour as

    firFilter, lowPassFIR, decimate, delaySignal, iqMixing      optic/dsp/core.py:87, 352, 435, 880, 925
    pbs, photodiode, balancedPD, opticalHybrid2x4,
    coherentReceiver, pdmCoherentReceiver                       optic/models/devices.py:223-668

Every call uploads its inputs once, runs all stages in device memory (PBS, polarisation delay,
hybrid + photodiodes, low-pass FIR, IQ imbalance, skew filters are one enqueued sequence) and
downloads the result.  There is no CPU fallback: without the HIP library the first call raises.

Host-side glue that stays in numpy on purpose: the FIR tap formulas (lowPassFIR: a few hundred
values), the 2x2 / 4x4 constant matrices of ``pbs`` and ``opticalHybrid2x4`` when they are called
on their own, dtype / shape handling.

Noise: the reference seeds numpy's global generator per photodiode (devices.py:368-389); the
device draws from counter-based Philox streams instead (one per pair of photodiodes, keyed by
``param.seed``; Box-Muller in single precision), so noisy runs agree with the reference statistically, not sample by sample
(same policy as the EDFA).  ``_unit_normals`` feeds host-supplied standard
normals through the same arithmetic for exact checks."""
import copy
import ctypes as C
import logging as logg

import numpy as np

from . import _lib
from . import device as _device
from .utils import parameters

_MODE = {"photodiode": 0, "balancedPD": 1, "coherentReceiver": 2, "pdmCoherentReceiver": 3, "iqMixing": 4}


class _HipBackend:
    """Calls into libssf_hip.so (tests swap in the CPU emulator of the same kernels).  All array
    arguments are raw pointers: host (numpy) or device (DeviceArray) memory alike."""

    def _dev(self):
        from .models import _state
        return _state["device"]

    def fir(self, N, ncols, ntaps, taps, x, out):
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_fir_filter(self._dev(), N, ncols, ntaps, taps, x, out))

    def fir_long(self, inLen, outLen, ncols, ntaps, taps, shift, x, out):
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_fir_long(self._dev(), int(inLen), int(outLen), int(ncols), int(ntaps), taps, int(shift), x, out))

    def delay(self, N, delay, Fs, x, out):
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_delay_signal(self._dev(), N, float(delay), float(Fs), x, out))

    def decimate(self, N, ncols, SpSin, dec, x, out):
        lib = _lib.load()
        sd = (C.c_int32 * ncols)()
        _lib.raise_for(lib, None, lib.ssf_decimate(self._dev(), N, ncols, int(SpSin), int(dec), x, out, sd))
        return list(sd)

    def pbs(self, N, ncols, theta, E, Ex, Ey):
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_pbs(self._dev(), N, ncols, float(theta), E, Ex, Ey))

    def hybrid(self, N, Es, Elo, Eo):
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_optical_hybrid_2x4(self._dev(), N, Es, Elo, Eo))

    def rx_chain(self, N, p, Es, Elo, taps, SpSin, dec, H, K, nfft, out):
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_rx_chain(self._dev(), N, C.byref(p), Es, Elo, taps.ctypes.data_as(C.c_void_p), len(taps),
                                                   SpSin, dec, H.ctypes.data_as(C.c_void_p), int(K), int(nfft), out, None))

    def rx(self, mode, N, nmodes, p, in0, lo, un, out):
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_rx_run(self._dev(), mode, N, nmodes, C.byref(p), in0, lo,
                                                 C.cast(un, C.POINTER(C.c_double)) if un is not None else None, out))


_backend = _HipBackend()
_dev = _device


def _fs(param, default=None):
    try:
        return param.Fs
    except AttributeError:
        if default is not None:
            return default
        logg.error("Simulation sampling frequency (Fs) not provided.")
        raise AttributeError("Simulation sampling frequency (Fs) not provided: set param.Fs") from None


def _is_complex(x):
    return x.dtype.kind == "c"


# ------------------------------------------------------------------------------------ filters
def lowPassFIR(fc, fs, N, typeF="rect"):
    """FIR coefficients of a low-pass filter (optic/dsp/core.py:352-392).  Host arithmetic: N values."""
    fu = fc / fs
    d = (N - 1) / 2
    n = np.arange(0, N)
    if typeF == "rect":
        h = (2 * fu) * np.sinc(2 * fu * (n - d))
    elif typeF == "gauss":
        h = np.sqrt(2 * np.pi / np.log(2)) * fu * np.exp(-(2 / np.log(2)) * (np.pi * fu * (n - d)) ** 2)
    else:
        raise ValueError("typeF must be 'rect' or 'gauss'")
    return h / np.sum(h)


def firFilter(h, x, prec=None):
    """FIR filtering with the filter delay compensated: 'same'-mode convolution of every column of x
    with h (optic/dsp/core.py:87-125; ``prec`` as in the cupy twin optic/dsp/coreGPU.py:27-78).  One
    overlap-save launch for all columns up to 4096 taps, ceil(taps / 4096) launches beyond (ssf_fir_long).  A complex128
    DeviceArray stays on the device."""
    on_dev = _dev.is_device(x)
    if not on_dev:
        x = np.asarray(x)
    taps = np.ascontiguousarray(h, dtype=np.complex128)
    input1D = x.ndim == 1
    x2 = x.reshape(len(x), 1) if input1D else x
    if len(taps) > _FIR_MAX_TAPS:
        y = _conv_shift(x2, taps, (len(taps) - 1) // 2, x2.shape[0], on_dev)
    else:
        xp, _keep = _dev.arg(x2, np.complex128)
        y = _dev.empty(on_dev, x2.shape, np.complex128)
        _backend.fir(x2.shape[0], x2.shape[1], len(taps), taps.ctypes.data_as(C.c_void_p), xp, _dev.out_ptr(y))
    if on_dev:
        return y.reshape(-1) if input1D else y
    if prec is not None:
        y = y.astype(prec)
    elif _is_complex(x2):
        y = y.astype(x2.dtype, copy=False)
    else:                                          # y = x.copy(); y[:, n] = ... keeps x's dtype (core.py:115-119)
        y = y.real.astype(x2.dtype if x2.dtype.kind == "f" else np.float64)
    return y.flatten() if input1D else y


_FIR_MAX_TAPS = 4096           # ssf_fir_filter: one LDS block pair per 8192-point block, at least half of it output


def _conv_shift(x2, h, shift, outLen, on_dev):
    """out[n, m] = sum_t h[t] x2[n + shift - t, m], n in [0, outLen), x2 zero-extended on both sides: ssf_fir_long, any number of
    taps, host or device signal -- a DeviceArray never leaves the device.  blockwiseFFTConv's result (optic/dsp/core.py:1043-1046)
    is shift = (len(h) - 1) // 2, outLen = len(x2).  Cost: ceil(len(h) / 4096) overlap-save passes over the signal."""
    taps = np.ascontiguousarray(h, dtype=np.complex128)
    xp, _keep = _dev.arg(x2, np.complex128)
    out = _dev.empty(on_dev, (int(outLen), x2.shape[1]), np.complex128)
    _backend.fir_long(x2.shape[0], outLen, x2.shape[1], len(taps), taps.ctypes.data_as(C.c_void_p), shift, xp, _dev.out_ptr(out))
    return out


def delaySignal(sig, delay, Fs=1, NFFT=1024):
    """Fractional delay by FFT overlap-save filtering (optic/dsp/core.py:880-922).  NFFT = 1024 (the reference's default,
    a 512-tap delay filter) is one device call; any other NFFT -- an NFFT // 2-sample frequency response, or None = the next
    power of two above the padded length, i.e. about N / 2 taps -- builds the reference's impulse response on the host (NFFT // 2
    values) and convolves on the device segment by segment (ssf_fir_long: ceil(NFFT / 8192) passes over the signal; at 2^20
    samples and NFFT = None that is 64 passes, tens of milliseconds).  The reference's zero padding, np.roll(-1) and [:N] cut
    are index arithmetic of that call.  A DeviceArray stays on the device (complex128 in, complex128 out)."""
    on_dev = _dev.is_device(sig)
    if NFFT != 1024:
        s1 = sig.reshape(-1) if on_dev else np.asarray(sig).reshape(-1)
        N = s1.shape[0]
        padLen = int(np.ceil(np.abs(delay * Fs)))                                  # core.py:905-909
        if NFFT is None:
            NFFT = 2 ** int(np.ceil(np.log2(N + padLen)))
        if int(NFFT) < 2:
            raise ValueError("delaySignal: NFFT must be at least 2")
        freq = np.fft.fftfreq(int(NFFT) // 2, d=1 / Fs)
        H = np.exp(-1j * 2 * np.pi * freq * delay)                                 # core.py:916
        h = np.fft.fftshift(np.fft.ifft(H))                                        # core.py:1015-1016: centred impulse response
        D = (len(h) - 1) // 2
        # y = conv_same(sig zero-padded by padLen)[:N + padLen]; roll(y, -1)[:N] (core.py:920-922) = y[1 : N + 1] -- and with no
        # padding the last sample wraps around to y[0]
        x2 = s1.reshape(N, 1)
        y = _conv_shift(x2, h, D + 1, N, on_dev)
        if padLen == 0:
            y0 = _conv_shift(x2, h, D, 1, on_dev)
            if on_dev:
                lib = _lib.load()
                _lib.raise_for(lib, None, lib.ssf_device_memcpy(y.device, C.c_void_p(y.ptr.value + 16 * (N - 1)), y0.ptr, 16))
            else:
                y[N - 1] = y0[0]
        y = y.reshape(-1)
        if on_dev:
            return y
        return y if np.any(np.iscomplex(s1)) else y.real
    if not on_dev:
        sig = np.asarray(sig)
    s1 = sig.reshape(-1)
    xp, _keep = _dev.arg(s1, np.complex128)
    out = _dev.empty(on_dev, s1.shape, np.complex128)
    _backend.delay(s1.shape[0], delay, Fs, xp, _dev.out_ptr(out))
    if on_dev:
        return out
    # core.py:1043-1046 decides on the values, not the dtype: a complex array whose imaginary parts are all
    # zero comes back as float64
    return out if np.any(np.iscomplex(sig)) else out.real


def decimate(sigIn, param):
    """Maximum-variance sampling phase per column, then every ``SpSin / SpSout``-th sample
    (optic/dsp/core.py:435-491)."""
    on_dev = _dev.is_device(sigIn)
    if not on_dev:
        sigIn = np.asarray(sigIn)
    input1D = sigIn.ndim == 1
    x2 = sigIn.reshape(len(sigIn), 1) if input1D else sigIn
    decFactor = int(param.SpSin / param.SpSout)
    if x2.shape[0] % param.SpSin:
        raise ValueError(f"cannot reshape array of size {x2.shape[0]} into shape ({param.SpSin})")
    xp, _keep = _dev.arg(x2, np.complex128)
    out = _dev.empty(on_dev, ((x2.shape[0] + decFactor - 1) // decFactor, x2.shape[1]), np.complex128)
    _backend.decimate(x2.shape[0], x2.shape[1], int(param.SpSin), decFactor, xp, _dev.out_ptr(out))
    if on_dev:
        return out.reshape(-1) if input1D else out
    if not _is_complex(sigIn):
        out = out.real
    out = out.astype(sigIn.dtype, copy=False)
    return out.flatten() if input1D else out


# ------------------------------------------------------------------------- passive optics
def pbs(E, θ=0):
    """Polarisation beam splitter (optic/models/devices.py:223-260): the input field rotated by θ and split, one element-wise
    device pass (``ssf_pbs``; inside pdmCoherentReceiver the same rotation rides in the first filter's loads).  ``E``: (N, 2), or
    (N,) taken as the x polarisation; numpy or complex128 DeviceArray.  Returns ``(Ex, Ey)``, (N,) each, of the caller's kind."""
    on_dev = _dev.is_device(E)
    if not on_dev:
        E = np.asarray(E)
    if E.ndim == 2 and E.shape[1] > 2:
        logg.error("E need to be a (N,2) or a (N,) np.array")
    if E.ndim not in (1, 2) or (E.ndim == 2 and E.shape[1] != 2):
        raise ValueError("pbs: E must have shape (N, 2) or (N,)")
    N = E.shape[0]
    ptr, _keep = _dev.arg(E, np.complex128)
    Ex, Ey = _dev.empty(on_dev, (N,), np.complex128), _dev.empty(on_dev, (N,), np.complex128)
    _backend.pbs(N, 1 if E.ndim == 1 else 2, θ, ptr, _dev.out_ptr(Ex), _dev.out_ptr(Ey))
    return Ex, Ey


def opticalHybrid2x4(Es, Elo):
    """2x4 90-degree optical hybrid (optic/models/devices.py:462-500): the four outputs ``T @ [Es, 0, 0, Elo]`` as a (4, N) array, one
    element-wise device pass (``ssf_optical_hybrid_2x4``); numpy or complex128 DeviceArrays (both of one kind)."""
    assert tuple(Es.shape) == (len(Es),), "Es need to have a (N,) shape"
    assert tuple(Elo.shape) == (len(Elo),), "Elo need to have a (N,) shape"
    assert tuple(Es.shape) == tuple(Elo.shape), "Es and Elo need to have the same (N,) shape"
    on_dev = _dev.is_device(Es) or _dev.is_device(Elo)
    ps, _k1 = _dev.arg(Es, np.complex128)
    pl, _k2 = _dev.arg(Elo, np.complex128)
    N = len(Es)
    Eo = _dev.empty(on_dev, (4, N), np.complex128)
    _backend.hybrid(N, ps, pl, _dev.out_ptr(Eo))
    return Eo


# ------------------------------------------------------------------------------ detection
def _pd_fields(p, paramPD, seed_offset=0):
    """photodiode defaults and checks (devices.py:331-353)."""
    g = lambda k, d: getattr(paramPD, k, d) if paramPD is not None else d   # noqa: E731
    p.R, p.Tc, p.Id, p.RL, p.B, p.IpdSat = g("R", 1), g("Tc", 25), g("Id", 5e-9), g("RL", 50), g("B", 30e9), g("IpdSat", 5e-3)
    N = g("N", 255)
    if N % 2 == 0:
        logg.warning("Number of filter taps (N) was even, incrementing by one to make it odd.")
    p.N = int(N)
    fType = g("fType", "rect")
    if fType not in ("rect", "gauss"):
        raise ValueError("fType must be 'rect' or 'gauss'")
    p.fType = 0 if fType == "rect" else 1
    p.ideal = int(bool(g("ideal", False)))
    p.shotNoise, p.thermalNoise = int(bool(g("shotNoise", True))), int(bool(g("thermalNoise", True)))
    p.currentSaturation = int(bool(g("currentSaturation", False)))
    p.bandwidthLimitation = int(bool(g("bandwidthLimitation", True)))
    assert p.R > 0, "PD responsivity should be a positive scalar"
    seed = g("seed", None)
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0] >> 1)
    p.rng_seed = int(seed) + seed_offset
    if not p.ideal:
        p.Fs = _fs(paramPD)
        assert p.Fs >= 2 * p.B, "Sampling frequency Fs needs to be at least twice of B."
    return p


def _rx(mode, N, nmodes, p, in0, lo, un, out_shape, out_dtype, npd):
    """Marshal one ssf_rx_run call: device arrays stay on the device, numpy arrays are converted."""
    on_dev = _dev.is_device(in0)
    ip, _k0 = _dev.arg(in0, np.complex128)
    lp, _k1 = (None, None) if lo is None else _dev.arg(lo, np.complex128)
    up, _k2 = (None, None)
    if un is not None:
        if not _dev.is_device(un):
            un = np.ascontiguousarray(un, dtype=np.float64)
        assert tuple(un.shape) == (npd, 2, N), f"_unit_normals must have shape ({npd}, 2, {N})"
        up, _k2 = _dev.arg(un, np.float64)
    out = _dev.empty(on_dev, out_shape, out_dtype)
    _backend.rx(mode, N, nmodes, p, ip, lp, up, _dev.out_ptr(out))
    return out


def photodiode(E, param=None, _unit_normals=None):
    """Pin photodiode (optic/models/devices.py:289-399): R |E|^2 (summed over modes), optional
    saturation, shot / thermal noise and the low-pass frequency response."""
    if not _dev.is_device(E):
        E = np.asarray(E)
    E2 = E.reshape(len(E), 1) if E.ndim == 1 else E
    p = _pd_fields(_lib.RxParams(), param)
    return _rx(_MODE["photodiode"], E2.shape[0], E2.shape[1], p, E2, None, _unit_normals, (E2.shape[0],), np.float64, 1)


def balancedPD(E1, E2, param=None, _unit_normals=None):
    """Balanced photodiode pair (optic/models/devices.py:402-459): i(E1) - i(E2)."""
    assert E1.shape == E2.shape, "E1 and E2 need to have the same shape"
    if len(E1.shape) != 1 or _dev.is_device(E1) or _dev.is_device(E2):
        # (N, M) fields: each photodiode sums |E|^2 over its modes (devices.py:355-357), the second one with seed + 1
        # (devices.py:447-456): two launches of the photodiode pipeline, the difference on the host / device.  Device arrays of any
        # shape go this way too (the reference's own formulation, i1 - i2): stacking them for the one-launch path below would
        # take them through the host
        param2 = param
        if param is not None and getattr(param, "seed", None) is not None:
            param2 = copy.copy(param)
            param2.seed = param.seed + 1
        un = _unit_normals
        i1 = photodiode(E1, param, None if un is None else un[0:1])
        i2 = photodiode(E2, param2, None if un is None else un[1:2])
        if _dev.is_device(i1) or _dev.is_device(i2):
            if not _dev.is_device(i1):
                i1 = _dev.to_device(i1)
            xp, _keep = _dev.arg(i2, np.float64)
            lib = _lib.load()
            _lib.raise_for(lib, None, lib.ssf_device_axpy(i1.device, i1.size, -1.0, xp, i1.ptr))    # i1 -= i2, in HBM
            return i1
        return i1 - i2
    p = _pd_fields(_lib.RxParams(), param)
    return _rx(_MODE["balancedPD"], len(E1), 2, p, np.stack([E1, E2], axis=1), None, _unit_normals, (len(E1),), np.float64, 2)


def _iq_fields(p, k, par):
    p.ampImb[k] = getattr(par, "ampImb", 0)
    p.phaseImb[k] = getattr(par, "phaseImb", 0)
    p.timeSkew[k] = getattr(par, "timeSkew", 0)


def iqMixing(sig, param):
    """IQ amplitude / phase imbalance and skew (optic/dsp/core.py:925-970)."""
    p = _lib.RxParams()
    Fs = getattr(param, "Fs", None)
    if Fs is None:
        logg.error("Sampling frequency not provided.")
        raise AttributeError("Sampling frequency not provided: set param.Fs")
    p.Fs = Fs
    _iq_fields(p, 0, param)
    if not _dev.is_device(sig):
        sig = np.asarray(sig)
    return _rx(_MODE["iqMixing"], len(sig), 1, p, sig.reshape(-1), None, None, (len(sig),), np.complex128, 0)


def _two_rates(p, fe_fs):
    """The reference takes paramPD.Fs for the photodiode model (noise scale, low-pass design: devices.py:331-353) and
    paramFE.Fs for the polarisation delay, IQ mixing and skew (devices.py:562-571, 648-651): ssf_rx_params carries both."""
    p.Fs_pd = p.Fs if p.Fs and abs(p.Fs - fe_fs) > 1e-12 * abs(fe_fs) else 0.0
    p.Fs = fe_fs


def coherentReceiver(Es, Elo, paramFE=None, paramPD=None, _unit_normals=None):
    """Single-polarisation coherent front-end (optic/models/devices.py:503-571): 2x4 hybrid, two
    balanced photodiode pairs, IQ impairments."""
    assert Es.shape == (len(Es),), "Es need to have a (N,) shape"
    assert Elo.shape == (len(Elo),), "Elo need to have a (N,) shape"
    assert Es.shape == Elo.shape, "Es and Elo need to have the same (N,) shape"
    Fs = _fs(paramFE)
    if paramPD is None:
        paramPD = parameters()
        paramPD.Fs = Fs
    p = _pd_fields(_lib.RxParams(), paramPD)
    _two_rates(p, Fs)
    _iq_fields(p, 0, paramFE)
    return _rx(_MODE["coherentReceiver"], len(Es), 1, p, Es, Elo, _unit_normals, (len(Es),), np.complex128, 4)


def pdmCoherentReceiver(Es, Elo, paramFE, paramPD=None, _unit_normals=None):
    """Polarisation-multiplexed coherent front-end (optic/models/devices.py:574-668).  paramFE: Fs,
    polRotation, pdl, polDelay, phaseImbX/Y, ampImbX/Y, timeSkewX/Y; paramPD: see photodiode.
    Returns the (N, 2) down-converted signal (a DeviceArray when Es is one)."""
    assert len(Es) == len(Elo), "Es and Elo need to have the same length"
    if not _dev.is_device(Es):
        Es = np.asarray(Es)
    if Es.ndim != 2 or Es.shape[1] != 2:
        raise ValueError("Es must be a (N, 2) polarisation-multiplexed field")
    p = _pdm_fields(paramFE, paramPD)
    lo = Elo if _dev.is_device(Elo) else np.asarray(Elo).reshape(-1)
    return _rx(_MODE["pdmCoherentReceiver"], len(Es), 2, p, Es, lo.reshape(-1), _unit_normals, (len(Es), 2), np.complex128, 8)


def _pdm_fields(paramFE, paramPD):
    """ssf_rx_params of a pdmCoherentReceiver call (devices.py:617-651)."""
    Fs = _fs(paramFE)
    if paramPD is None:
        paramPD = parameters()
        paramPD.Fs = Fs
    p = _pd_fields(_lib.RxParams(), paramPD)
    _two_rates(p, Fs)
    p.polRotation = getattr(paramFE, "polRotation", 0)
    p.pdl = getattr(paramFE, "pdl", 0)
    p.polDelay = getattr(paramFE, "polDelay", 0)
    for k, s in enumerate("XY"):
        p.ampImb[k] = getattr(paramFE, "ampImb" + s, 0)
        p.phaseImb[k] = getattr(paramFE, "phaseImb" + s, 0)
        p.timeSkew[k] = getattr(paramFE, "timeSkew" + s, 0)
    return p


def pdmCoherentReceiverChain(Es, Elo, paramFE, paramPD, h, paramDec, paramEDC):
    """``edc(decimate(firFilter(h, pdmCoherentReceiver(Es, Elo, paramFE, paramPD)), paramDec), paramEDC)`` -- the receiver side of the
    coherent notebooks (examples/test_WDM_transmission.ipynb cells 17 - 23) -- as ONE call into the library (``ssf_rx_chain``): the
    same result as the four calls, one host wait instead of four, decimate's variance search in the matched filter's stores and its
    gather in the compensating filter's loads.  No reference equivalent (the reference has the four functions); every argument is
    the corresponding call's.  ``Es``: (N, 2) numpy or complex128 DeviceArray (then a DeviceArray comes back)."""
    from . import models as _m
    assert len(Es) == len(Elo), "Es and Elo need to have the same length"
    on_dev = _dev.is_device(Es)
    if not on_dev:
        Es = np.asarray(Es)
    if Es.ndim != 2 or Es.shape[1] != 2:
        raise ValueError("Es must be a (N, 2) polarisation-multiplexed field")
    N = len(Es)
    p = _pdm_fields(paramFE, paramPD)
    taps = np.ascontiguousarray(h, dtype=np.complex128)
    if len(taps) > _FIR_MAX_TAPS:
        raise ValueError("pdmCoherentReceiverChain: a matched filter of at most %d taps (use the separate calls)" % _FIR_MAX_TAPS)
    decFactor = int(paramDec.SpSin / paramDec.SpSout)
    if N % paramDec.SpSin:
        raise ValueError(f"cannot reshape array of size {N} into shape ({paramDec.SpSin})")
    K, Nfft, Hf = _m._edc_filter(paramEDC, _m._require_fs(paramEDC))
    if Nfft < K:
        raise ValueError("FFT size is smaller than filter length")
    if K > _m._OLS_MAX_TAPS:
        raise ValueError("pdmCoherentReceiverChain: a compensating filter of at most %d taps (use the separate calls)" % _m._OLS_MAX_TAPS)
    nfft = _m._ols_block(K)
    H = _m._edc_block_response(K, nfft, Hf)
    ip, _k0 = _dev.arg(Es, np.complex128)
    lo = Elo if _dev.is_device(Elo) else np.asarray(Elo).reshape(-1)
    lp, _k1 = _dev.arg(lo.reshape(-1), np.complex128)
    out = _dev.empty(on_dev, ((N + decFactor - 1) // decFactor, 2), np.complex128)
    _backend.rx_chain(N, p, ip, lp, taps, int(paramDec.SpSin), decFactor, H, K, nfft, _dev.out_ptr(out))
    return out
