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
device draws from counter-based Philox streams instead (one per photodiode, keyed by
``param.seed``), so noisy runs agree with the reference statistically, not sample by sample
(same policy as the EDFA).  ``_unit_normals`` feeds host-supplied standard
normals through the same arithmetic for exact checks."""
import ctypes as C
import logging as logg

import numpy as np

from . import _lib
from .utils import parameters

_MODE = {"photodiode": 0, "balancedPD": 1, "coherentReceiver": 2, "pdmCoherentReceiver": 3, "iqMixing": 4}


class _HipBackend:
    """Calls into libssf_hip.so (tests swap in the CPU emulator of the same kernels)."""

    def _dev(self):
        from .models import _state
        return _state["device"]

    def fir(self, x, taps):
        lib = _lib.load()
        out = np.empty_like(x)
        rc = lib.ssf_fir_filter(self._dev(), x.shape[0], x.shape[1], len(taps), taps.ctypes.data_as(C.c_void_p),
                                x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        _lib.raise_for(lib, None, rc)
        return out

    def delay(self, x, delay, Fs):
        lib = _lib.load()
        out = np.empty_like(x)
        rc = lib.ssf_delay_signal(self._dev(), x.shape[0], float(delay), float(Fs), x.ctypes.data_as(C.c_void_p),
                                  out.ctypes.data_as(C.c_void_p))
        _lib.raise_for(lib, None, rc)
        return out

    def decimate(self, x, SpSin, dec):
        lib = _lib.load()
        out = np.empty(((x.shape[0] + dec - 1) // dec, x.shape[1]), dtype=np.complex128)
        sd = (C.c_int32 * x.shape[1])()
        rc = lib.ssf_decimate(self._dev(), x.shape[0], x.shape[1], int(SpSin), int(dec), x.ctypes.data_as(C.c_void_p),
                              out.ctypes.data_as(C.c_void_p), sd)
        _lib.raise_for(lib, None, rc)
        return out, list(sd)

    def rx(self, mode, N, nmodes, p, in0, lo, un, out):
        lib = _lib.load()
        rc = lib.ssf_rx_run(self._dev(), mode, N, nmodes, C.byref(p), in0.ctypes.data_as(C.c_void_p),
                            lo.ctypes.data_as(C.c_void_p) if lo is not None else None,
                            un.ctypes.data_as(C.POINTER(C.c_double)) if un is not None else None,
                            out.ctypes.data_as(C.c_void_p))
        _lib.raise_for(lib, None, rc)


_backend = _HipBackend()


def _c128(x):
    return np.ascontiguousarray(x, dtype=np.complex128)


def _fs(param, default=None):
    try:
        return param.Fs
    except AttributeError:
        if default is not None:
            return default
        logg.error("Simulation sampling frequency (Fs) not provided.")
        raise AttributeError("Simulation sampling frequency (Fs) not provided: set param.Fs") from None


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
    overlap-save launch for all columns; at most 4096 taps."""
    x = np.asarray(x)
    h = np.asarray(h)
    input1D = x.ndim == 1
    x2 = x.reshape(len(x), 1) if input1D else x
    y = _backend.fir(_c128(x2), _c128(h))
    if prec is not None:
        y = y.astype(prec)
    elif np.iscomplexobj(x2):
        y = y.astype(x2.dtype, copy=False)
    else:                                          # y = x.copy(); y[:, n] = ... keeps x's dtype (core.py:115-119)
        y = y.real.astype(x2.dtype if x2.dtype.kind == "f" else np.float64)
    return y.flatten() if input1D else y


def delaySignal(sig, delay, Fs=1, NFFT=1024):
    """Fractional delay by FFT overlap-save filtering (optic/dsp/core.py:880-922); NFFT is fixed to the
    reference's default 1024."""
    if NFFT != 1024:
        raise ValueError("delaySignal on the GPU uses NFFT = 1024 (the reference's default)")
    sig = np.asarray(sig)
    out = _backend.delay(_c128(sig.reshape(-1, 1)), delay, Fs).reshape(-1)
    return out if np.iscomplexobj(sig) else out.real      # core.py:1043-1046


def decimate(sigIn, param):
    """Maximum-variance sampling phase per column, then every ``SpSin / SpSout``-th sample
    (optic/dsp/core.py:435-491)."""
    sigIn = np.asarray(sigIn)
    input1D = sigIn.ndim == 1
    x2 = sigIn.reshape(len(sigIn), 1) if input1D else sigIn
    decFactor = int(param.SpSin / param.SpSout)
    if x2.shape[0] % param.SpSin:
        raise ValueError(f"cannot reshape array of size {x2.shape[0]} into shape ({param.SpSin})")
    out, _ = _backend.decimate(_c128(x2), int(param.SpSin), decFactor)
    if not np.iscomplexobj(sigIn):
        out = out.real
    out = out.astype(sigIn.dtype, copy=False)
    return out.flatten() if input1D else out


# ------------------------------------------------------------------------- passive optics (host glue)
def pbs(E, θ=0):
    """Polarisation beam splitter (optic/models/devices.py:223-260): 2x2 rotation, host glue.  Inside
    pdmCoherentReceiver the same rotation runs on the device."""
    E = np.asarray(E)
    if E.ndim == 1:
        E = np.repeat(E, 2).reshape(-1, 2)
        E[:, 1] = 0
    elif E.shape[1] > 2:
        logg.error("E need to be a (N,2) or a (N,) np.array")
    rot = np.array([[np.cos(θ), -np.sin(θ)], [np.sin(θ), np.cos(θ)]]) + 1j * 0
    E = E @ rot
    return E[:, 0], E[:, 1]


def opticalHybrid2x4(Es, Elo):
    """2x4 90-degree optical hybrid (optic/models/devices.py:462-500): constant 4x4 matrix, host glue."""
    assert Es.shape == (len(Es),), "Es need to have a (N,) shape"
    assert Elo.shape == (len(Elo),), "Elo need to have a (N,) shape"
    assert Es.shape == Elo.shape, "Es and Elo need to have the same (N,) shape"
    T = np.array([[1 / 2, 1j / 2, 1j / 2, -1 / 2], [1j / 2, -1 / 2, 1 / 2, 1j / 2],
                  [1j / 2, 1 / 2, -1j / 2, -1 / 2], [-1 / 2, 1j / 2, -1 / 2, 1j / 2]])
    return T @ np.array([Es, np.zeros((Es.size,)), np.zeros((Es.size,)), Elo])


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


def photodiode(E, param=None, _unit_normals=None):
    """Pin photodiode (optic/models/devices.py:289-399): R |E|^2 (summed over modes), optional
    saturation, shot / thermal noise and the low-pass frequency response."""
    E = np.asarray(E)
    E2 = E.reshape(len(E), 1) if E.ndim == 1 else E
    p = _pd_fields(_lib.RxParams(), param)
    out = np.empty(E2.shape[0], dtype=np.float64)
    _backend.rx(_MODE["photodiode"], E2.shape[0], E2.shape[1], p, _c128(E2), None, _normals(_unit_normals, 1, len(out)), out)
    return out


def balancedPD(E1, E2, param=None, _unit_normals=None):
    """Balanced photodiode pair (optic/models/devices.py:402-459): i(E1) - i(E2)."""
    assert E1.shape == E2.shape, "E1 and E2 need to have the same shape"
    if np.asarray(E1).ndim != 1:
        raise ValueError("balancedPD on the GPU takes (N,) fields")
    p = _pd_fields(_lib.RxParams(), param)
    out = np.empty(len(E1), dtype=np.float64)
    _backend.rx(_MODE["balancedPD"], len(E1), 2, p, _c128(np.stack([E1, E2], axis=1)), None,
                _normals(_unit_normals, 2, len(out)), out)
    return out


def _normals(un, npd, N):
    if un is None:
        return None
    un = np.ascontiguousarray(un, dtype=np.float64)
    assert un.shape == (npd, 2, N), f"_unit_normals must have shape ({npd}, 2, {N})"
    return un


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
    sig = np.asarray(sig)
    out = np.empty(len(sig), dtype=np.complex128)
    _backend.rx(_MODE["iqMixing"], len(sig), 1, p, _c128(sig), None, None, out)
    return out


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
    p.Fs = Fs
    _iq_fields(p, 0, paramFE)
    out = np.empty(len(Es), dtype=np.complex128)
    _backend.rx(_MODE["coherentReceiver"], len(Es), 1, p, _c128(Es), _c128(Elo), _normals(_unit_normals, 4, len(Es)), out)
    return out


def pdmCoherentReceiver(Es, Elo, paramFE, paramPD=None, _unit_normals=None):
    """Polarisation-multiplexed coherent front-end (optic/models/devices.py:574-668).  paramFE: Fs,
    polRotation, pdl, polDelay, phaseImbX/Y, ampImbX/Y, timeSkewX/Y; paramPD: see photodiode.
    Returns the (N, 2) down-converted signal."""
    assert len(Es) == len(Elo), "Es and Elo need to have the same length"
    Es = np.asarray(Es)
    if Es.ndim != 2 or Es.shape[1] != 2:
        raise ValueError("Es must be a (N, 2) polarisation-multiplexed field")
    Fs = _fs(paramFE)
    if paramPD is None:
        paramPD = parameters()
        paramPD.Fs = Fs
    p = _pd_fields(_lib.RxParams(), paramPD)
    p.Fs = Fs
    p.polRotation = getattr(paramFE, "polRotation", 0)
    p.pdl = getattr(paramFE, "pdl", 0)
    p.polDelay = getattr(paramFE, "polDelay", 0)
    for k, s in enumerate("XY"):
        p.ampImb[k] = getattr(paramFE, "ampImb" + s, 0)
        p.phaseImb[k] = getattr(paramFE, "phaseImb" + s, 0)
        p.timeSkew[k] = getattr(paramFE, "timeSkew" + s, 0)
    out = np.empty((len(Es), 2), dtype=np.complex128)
    _backend.rx(_MODE["pdmCoherentReceiver"], len(Es), 2, p, _c128(Es), _c128(np.asarray(Elo).reshape(-1)),
                _normals(_unit_normals, 8, len(Es)), out)
    return out
