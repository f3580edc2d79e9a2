"""WDM transmitter on the GPU: ``simpleWDMTx`` with the reference's parameters
and return values (optic/models/tx.py:42-228), signal path on the device through ``ssf_wdm_tx``.

Split of the work:
  host (numpy, a few thousand values, and it has to be numpy: these are the reference's seeded
  ``np.random`` draws) -- constellation and its pmf, the symbol sequences of every channel and
  polarisation, the pulse-shaping taps, the LO phase-noise random walk, the WDM grid;
  device -- everything that touches the N = nSymbols * SpS samples: zero-stuffing + pulse-shaping FIR,
  peak normalisation, IQ modulator, power normalisation, frequency shift, accumulation of the
  channels.  ``device_output=True`` returns the field as a DeviceArray, ready for ``manakovSSF``.

With a seed the result equals the reference's to rounding (same draws, same arithmetic); without one
both draw fresh entropy."""
import ctypes as C
import logging as logg

import numpy as np

from . import _lib
from . import device as _dev
from .utils import parameters

_DEFAULTS = (("M", 16), ("constType", "qam"), ("Rs", 32e9), ("SpS", 16), ("probDist", "uniform"), ("shapingFactor", 0),
             ("seed", None), ("nBits", 60000), ("pulseType", "rrc"), ("nFilterTaps", 1024), ("pulseRollOff", 0.01),
             ("mzmScale", 0.5), ("powerPerChannel", -3), ("nChannels", 5), ("Fc", 193.1e12), ("laserLinewidth", 0),
             ("wdmGridSpacing", 50e9), ("nPolModes", 1), ("prgsBar", True))


class _HipBackend:
    def wdm_tx(self, p, symbols, taps, phi, amp, deltaF, out_ptr, power):
        from .models import _state
        lib = _lib.load()
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None   # noqa: E731
        _lib.raise_for(lib, None, lib.ssf_wdm_tx(_state["device"], C.byref(p), symbols.ctypes.data_as(C.c_void_p), dp(taps),
                                                 dp(phi), dp(amp), dp(deltaF), out_ptr, dp(power)))


_backend = _HipBackend()


# ------------------------------------------------------------------ constellations (optic/comm/modulation.py:35-198)
def _constellation(M, constType):
    """Constellation in the reference's own point order (qamConst / pskConst / pamConst)."""
    if constType == "qam":
        side = int(np.sqrt(M))
        if side * side != M:
            raise ValueError("square QAM needs M = 4, 16, 64, ...")
        levels = np.arange(-(side - 1), side, 2)
        grid = levels[None, :] + 1j * levels[::-1, None]            # rows: imaginary part from +L down to -L
        grid[1::2] = grid[1::2, ::-1]                               # boustrophedon rows
        return grid
    if constType == "psk":
        return np.exp(1j * np.arange(0, 2 * np.pi, 2 * np.pi / M))
    if constType == "pam":
        return np.arange(-(M - 1), M, 2)
    raise ValueError("constType must be 'qam', 'psk' or 'pam'")


def grayMapping(M, constType):
    """Constellation sorted by the integer value of each point's Gray label (modulation.py:64-118)."""
    const = np.asarray(_constellation(M, constType)).reshape(M)
    labels = np.arange(M) ^ (np.arange(M) >> 1)                    # grayCode(log2 M), as integers
    dtype = np.float32 if constType == "pam" else np.complex64     # the reference stores the table in single precision
    return const.astype(dtype)[np.argsort(labels, kind="stable")]


def _symbol_source(nSymbols, M, constType, dist, shapingFactor, seed):
    """optic/comm/sources.py:137-212: unit-energy constellation, np.random.choice under np.random.seed."""
    if seed is not None:
        np.random.seed(seed)
    const = np.asarray(_constellation(M, constType))
    if dist == "uniform":
        px = np.ones(M) / M
    else:
        px = np.exp(-shapingFactor * np.abs(const) ** 2)
        px = (px / np.sum(px)).flatten()
    const = const / np.sqrt(np.sum(px * np.abs(const.flatten()) ** 2))
    if dist == "uniform" and M & (M - 1) == 0:
        # np.random.choice(a, n, p) is a.take(cdf.searchsorted(random_sample(n), 'right')) with cdf = cumsum(p) / cdf[-1]; for p = 1 / M
        # with M a power of two the cdf is k / M exactly, so the index is floor(u M): the same draws from the same stream without the
        # per-call validation and the binary search (2.4 -> 0.3 ms per 65 536 symbols: 22 such calls per 11-channel transmitter)
        return const.flatten()[(np.random.random_sample(nSymbols) * M).astype(np.intp)]
    return np.random.choice(const.flatten(), nSymbols, p=px)


# ------------------------------------------------------------------ pulse shapes (optic/dsp/core.py:129-270)
def _rrc(t, alpha):
    h = np.empty(len(t))
    for i, ti in enumerate(t):                                     # a few hundred taps: scalar code keeps the
        if ti == 0:                                                # reference's branch structure and rounding
            h[i] = 1 + alpha * (4 / np.pi - 1)
        elif abs(ti) == 1 / (4 * alpha):
            h[i] = (alpha / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha)) + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
        else:
            t1, t2 = np.pi * ti / 1, 4 * alpha * ti / 1
            h[i] = (1 / 1) * (np.sin(t1 * (1 - alpha)) + 4 * alpha * ti / 1 * np.cos(t1 * (1 + alpha))) / (np.pi * ti * (1 - t2**2))
    return h


def _rc(t, alpha):
    h = np.empty(len(t))
    for i, ti in enumerate(t):
        if abs(ti) == 1 / (2 * alpha):
            h[i] = np.pi / 4 * np.sinc(1 / (2 * alpha))
        else:
            h[i] = np.sinc(ti) * np.cos(np.pi * alpha * ti) / (1 - 4 * alpha**2 * ti**2)
    return h


_PULSES = {}


def pulseShape(param):
    """Pulse-shaping filter taps, normalised to unit sum (optic/dsp/core.py:217-270): 'rect', 'nrz', 'rrc', 'rc'."""
    kind = getattr(param, "pulseType", "rrc")
    SpS = getattr(param, "SpS", 2)
    n = getattr(param, "nFilterTaps", 256)
    ro = getattr(param, "rollOff", 0.1)
    if kind == "rect":
        h = np.concatenate((np.zeros(int(SpS / 2)), np.ones(SpS), np.zeros(int(SpS / 2))))
    elif kind == "nrz":
        t = np.linspace(-2, 2, SpS)
        h = np.convolve(np.ones(SpS), 2 / np.sqrt(np.pi) * np.exp(-(t**2)), mode="full")
    elif kind in ("rrc", "rc"):
        key = (kind, SpS, n, ro)
        if key not in _PULSES:                                     # (a thousand scalar evaluations: kept between calls)
            t = np.linspace(-n // 2, n // 2, n) * (1 / SpS)
            if len(_PULSES) >= 8:
                _PULSES.pop(next(iter(_PULSES)))
            _PULSES[key] = _rrc(t, ro) if kind == "rrc" else _rc(t, ro)
        h = _PULSES[key]
    else:
        raise ValueError("pulseType must be 'rect', 'nrz', 'rrc' or 'rc'")
    return h / np.sum(h)


def phaseNoise(lw, Nsamples, Ts, seed=None):
    """Laser phase-noise random walk (optic/dsp/core.py:792-826): the same draws, accumulated in order."""
    if seed is not None:
        np.random.seed(seed)
    steps = np.random.normal(0, np.sqrt(2 * np.pi * lw * Ts), max(Nsamples - 1, 0))
    return np.concatenate(([0.0], np.cumsum(steps)))[:Nsamples]


def basicLaserModel(param=None):
    """Laser with a Maxwellian random-walk phase and relative intensity noise (optic/models/devices.py:729-790): the receiver's
    local oscillator in the coherent notebooks.  Parameters (defaults): P [10 dBm], lw [1e3 Hz], RIN_var [1e-20], Fs, Ns
    [1000], seed [None], freqShift [0 Hz].  The seeded draws are the reference's ``np.random`` draws (phase walk with ``seed``,
    intensity noise with ``seed + 73``), so with a seed the field equals the reference's; a host array, which
    ``pdmCoherentReceiver`` uploads with the signal."""
    from .models import gaussianComplexNoise
    try:
        Fs = param.Fs
    except AttributeError:
        raise AttributeError("basicLaserModel: simulation sampling frequency (param.Fs) not provided") from None
    P = getattr(param, "P", 10)
    lw = getattr(param, "lw", 1e3)
    RIN_var = getattr(param, "RIN_var", 1e-20)
    Ns = int(getattr(param, "Ns", 1000))
    seed = getattr(param, "seed", None)
    fshift = getattr(param, "freqShift", 0)
    pn = phaseNoise(lw, Ns, 1 / Fs, None if seed is None else seed)
    deltaP = gaussianComplexNoise(pn.shape, RIN_var, None if seed is None else seed + 73)
    fo = 2 * np.pi * fshift * np.arange(Ns) / Fs if fshift != 0 else 0
    return np.sqrt(1e-3 * 10 ** (P / 10) + deltaP) * np.exp(1j * (fo + pn))


# ------------------------------------------------------------------ simpleWDMTx
def simpleWDMTx(param, device_output=False):
    """Simple WDM transmitter (optic/models/tx.py:42-228).  Parameters (defaults): M [16], constType ['qam'], Rs
    [32e9], SpS [16], probDist ['uniform'], shapingFactor [0], seed [None], nBits [60000], pulseType ['rrc'],
    nFilterTaps [1024], pulseRollOff [0.01], mzmScale [0.5], powerPerChannel [-3 dBm, scalar or list], nChannels [5],
    Fc [193.1e12], laserLinewidth [0], wdmGridSpacing [50e9], nPolModes [1], prgsBar [True].

    Returns (sigTxWDM (N, nPolModes), symbTxWDM (nSymbols, nPolModes, nChannels), param) with param.pmf and
    param.wdmFreqGrid set; ``device_output=True`` leaves sigTxWDM in HBM (DeviceArray)."""
    for k, d in _DEFAULTS:
        setattr(param, k, getattr(param, k, d))
    Fs = 1 / ((1 / param.Rs) / param.SpS)
    bits = int(np.log2(param.M))
    nSymbols = int(param.nBits / np.log2(param.M))
    constSymb = grayMapping(param.M, param.constType)
    if param.probDist == "uniform":
        px = np.ones(param.M) / param.M
    elif param.probDist == "maxwell-boltzmann":
        px = np.exp(-param.shapingFactor * np.abs(constSymb) ** 2)
        px = px / np.sum(px)
    else:
        raise ValueError("Invalid probability distribution.")
    param.pmf = px
    pp = parameters()
    pp.pulseType, pp.nFilterTaps, pp.rollOff, pp.SpS = param.pulseType, param.nFilterTaps, param.pulseRollOff, param.SpS
    pulse = np.ascontiguousarray(pulseShape(pp), dtype=np.float64)
    nCh, nPol = int(param.nChannels), int(param.nPolModes)
    freqGrid = np.arange(-np.floor(nCh / 2), np.floor(nCh / 2) + 1, 1) * param.wdmGridSpacing
    if nCh % 2 == 0:
        freqGrid += param.wdmGridSpacing / 2
    if type(param.powerPerChannel) == list:
        assert len(param.powerPerChannel) == nCh, "list length of power per channel does not match number of channels."
        Pch = 10 ** (np.array(param.powerPerChannel) / 10) * 1e-3
    else:
        Pch = 10 ** (param.powerPerChannel / 10) * 1e-3 * np.ones(nCh)
    N = int(nSymbols * param.SpS)
    if nSymbols != param.nBits // bits:
        raise ValueError("nBits must give the same symbol count for the source and the time axis")   # tx.py:112, 125

    symbols = np.empty((nCh, nPol, nSymbols), dtype=np.complex128)
    # Laser phase noise.  With a seed the random walk is the reference's own np.random draws (phaseNoise below: host, N values
    # per channel, uploaded).  WITHOUT param.seed the walk is generated on the device (Philox, two launches per channel:
    # ssf_tx_params.pn_seed) -- no N-sample draw on the host, no N-sample upload.  A deliberate, documented deviation (statistical
    # parity): the reference draws its N - 1 normals per channel from numpy's GLOBAL stream even then (tx.py:199, also for a zero
    # linewidth), so a caller who seeds np.random globally and leaves param.seed = None gets a reproducible reference run whose
    # symbols after channel 0 / mode 0 this call does not reproduce draw for draw; pass param.seed for that.
    device_pn = bool(param.laserLinewidth) and param.seed is None
    # WITH a seed the reference reseeds np.random with the SAME param.seed before every channel's walk (tx.py:199): every channel
    # gets the same N - 1 draws.  They are drawn once and one row crosses the bus (ssf_tx_params.phi_rows = 1); np.random's global
    # state is left as the reference leaves it -- the state after the walk's draws when that is the last thing the loop draws
    # (one polarisation), the last symbol source's otherwise.  (Round 5 drew them per channel: 11 x 19 ms at 2^20 samples.)
    phi = None
    seed = param.seed
    walk_state = None
    for ch in range(nCh):                                          # the reference's draw order (tx.py:184-210)
        logg.info("channel %d\t fc : %3.4f THz" % (ch, (param.Fc + freqGrid[ch]) / 1e12))
        for mode in range(nPol):
            logg.info("  mode #%d\t power: %.2f dBm" % (mode, 10 * np.log10((Pch[ch] / nPol) / 1e-3)))
            symbols[ch, mode] = _symbol_source(nSymbols, param.M, param.constType, param.probDist, param.shapingFactor, seed)
            if param.seed is not None:
                seed += 1
            if mode == 0 and param.seed is not None:
                if ch == 0 and param.laserLinewidth:
                    phi = np.ascontiguousarray(phaseNoise(param.laserLinewidth, N, 1 / Fs, seed=param.seed).reshape(1, N))
                    walk_state = np.random.get_state()
                elif nPol == 1 and ch == nCh - 1:                  # the loop's last draw: leave the stream where the reference does
                    if walk_state is not None:
                        np.random.set_state(walk_state)
                    else:
                        phaseNoise(param.laserLinewidth, N, 1 / Fs, seed=param.seed)

    p = _lib.TxParams(Fs=Fs, mzmScale=float(param.mzmScale), nSymbols=nSymbols, SpS=int(param.SpS), nChannels=nCh,
                      nPolModes=nPol, ntaps=len(pulse))
    if phi is not None:
        p.phi_rows = 1
    if device_pn:
        from .models import _device_seed
        p.pn_sigma = float(np.sqrt(2 * np.pi * param.laserLinewidth / Fs))
        p.pn_seed = _device_seed(None)
    amp = np.sqrt(Pch / nPol).astype(np.float64)
    power = np.zeros(nCh * nPol)
    sig = _dev.empty(device_output, (N, nPol), np.complex128)
    _backend.wdm_tx(p, symbols, pulse, phi, amp, np.ascontiguousarray(freqGrid, dtype=np.float64), _dev.out_ptr(sig), power)
    for ch in range(nCh):
        logg.info("channel %d\t power: %.2f dBm\n" % (ch, 10 * np.log10(np.sum(power[ch * nPol:(ch + 1) * nPol]) / 1e-3)))
    logg.info("total WDM signal power: %.2f dBm" % (10 * np.log10(np.sum(power) / 1e-3)))
    param.wdmFreqGrid = freqGrid
    symbTxWDM = np.ascontiguousarray(np.transpose(symbols, (2, 1, 0)))           # (nSymbols, nPolModes, nChannels)
    return sig, symbTxWDM, param
