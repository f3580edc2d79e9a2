"""Host side of the MI355X split-step Fourier path: the functions a notebook
imports instead of ``optic.models.modelsGPU`` (reference optic/models/modelsGPU.py).

    from opticommpy_amd.modelsGPU import manakovSSF, manakovDBP, ssfm, edfa

Same ``f(Ei, param)`` parameters-object API (defaults written back onto ``param``,
``returnParameters``, ``saveSpanN``, ``prgsBar``), numpy in / numpy out.  All
propagation runs in hand-written HIP kernels behind the C ABI of include/ssf.h;
this module only validates arguments, converts the (N, ncols) field to the
struct-of-arrays layout the ABI takes, and drives the per-span progress bar.
"""
import atexit
import ctypes as C
import logging as logg
import os
import threading
import time
from collections import OrderedDict

import numpy as np

from . import _lib
from . import device as _dev
from .utils import parameters

_H_PLANCK = 6.62607015e-34          # scipy.constants.h (devices.py:721)

NONCONV_WARNING = "Warning: target SSFM error tolerance was not achieved in {} iterations"

_state = {"device": int(os.environ.get("SSF_DEVICE", os.environ.get("LOCAL_RANK", "0")) or 0),
          "engine": {"auto": 0, "rocfft": 1, "fused": 2}[os.environ.get("SSF_ENGINE", "auto").lower()]}

#: stats (and trace, when requested) of the most recent propagation call
last_run = {}


def set_device(index):
    """Select the GPU used by subsequent calls (default: $SSF_DEVICE / $LOCAL_RANK / 0)."""
    _state["device"] = int(index)


def set_engine(name):
    """'auto' (default) | 'rocfft' (any N) | 'fused' (N = 2^m, roofline pipeline)."""
    _state["engine"] = {"auto": 0, "rocfft": 1, "fused": 2}[name]


def checkGPU():
    """True when a HIP device is usable (reference optic/dsp/coreGPU.py:11-24)."""
    try:
        return _lib.load().ssf_device_count() > 0
    except (RuntimeError, OSError, AttributeError):
        return False


# ----------------------------------------------------------------------------
# plan cache: one handle per (device, N, nrows, precision, engine)
# ----------------------------------------------------------------------------
class UnitsUnsupported(RuntimeError):
    """ssf_plan_set_units refused: this plan's pipeline (rocFFT / Bluestein / one-launch rows) cannot carry independent units;
    the caller falls back to one call per unit (include/ssf.h)."""


class _Plan:
    def __init__(self, device, N, nrows, prec_code, engine, units=1):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.ssf_plan_create(device, N, nrows, prec_code, engine, C.byref(h))
        _lib.raise_for(self.lib, None, rc)
        if units > 1:                                         # rows form `units` independent fields (include/ssf.h)
            rc = self.lib.ssf_plan_set_units(h, units)
            if rc:
                try:
                    if rc == -6:
                        raise UnitsUnsupported(_lib.error_message(self.lib, h, rc))
                    _lib.raise_for(self.lib, h, rc)
                finally:
                    self.lib.ssf_plan_destroy(h)
        self.h, self.N, self.nrows, self.prec_code, self.units = h, N, nrows, prec_code, units
        self.dtype = np.complex128 if prec_code == _lib.SSF_C128 else np.complex64

    def check(self, rc):
        _lib.raise_for(self.lib, self.h, rc)

    def close(self):
        if self.h:
            self.lib.ssf_plan_destroy(self.h)
            self.h = None


# Plans are cached per host thread: a plan (device buffers + stream + control block) is not thread-safe, and
# mgpu.run_sharded drives two lanes per GPU from two threads.
_tls = threading.local()
_all_plan_caches = []
_caches_lock = threading.Lock()
_MAX_PLANS = 4


def _plans():
    d = getattr(_tls, "plans", None)
    if d is None:
        d = _tls.plans = OrderedDict()
        with _caches_lock:
            _all_plan_caches.append(d)
    return d


def _get_plan(N, nrows, prec_code, engine=None, units=1):
    """engine: None = the process-wide choice (set_engine); an explicit code otherwise (never through the global: other
    threads -- run_sharded's lanes -- create plans at the same time)."""
    key = (_state["device"], int(N), int(nrows), prec_code, _state["engine"] if engine is None else engine, int(units))
    plans = _plans()
    pl = plans.pop(key, None)
    if pl is None:
        while len(plans) >= _MAX_PLANS:
            plans.popitem(last=False)[1].close()
        pl = _Plan(*key)
    plans[key] = pl
    pl.lib.ssf_plan_set_lanes(pl.h, int(getattr(_tls, "lanes", 1)))       # (mgpu.run_sharded: this thread is one of several lanes)
    return pl


def _set_lane_hint(n):
    """Called by the lanes of mgpu.run_sharded: the plans this thread uses share the GPU with n - 1 other lanes."""
    _tls.lanes = max(1, int(n))


def engine_supported(name, N, nrows=2, prec=np.complex128):
    """True when engine `name` can build a plan for this shape on the current device."""
    eng = {"auto": 0, "rocfft": 1, "fused": 2}[name]
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.ssf_plan_create(_state["device"], int(N), int(nrows), _prec_code(prec), eng, C.byref(h))
    if rc == 0:
        lib.ssf_plan_destroy(h)
    return rc == 0


def release_plans():
    """Free every cached plan (device memory, FFT plans, streams) of every thread; call it when no propagation is
    in flight."""
    with _caches_lock:
        caches = list(_all_plan_caches)
    for d in caches:
        while d:
            d.popitem()[1].close()


atexit.register(release_plans)


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def _require_fs(param):
    try:
        return param.Fs
    except AttributeError:
        logg.error("Simulation sampling frequency (Fs) not provided.")
        raise AttributeError("Simulation sampling frequency (Fs) not provided: set param.Fs") from None


def _prec_code(prec):
    return _lib.SSF_C64 if np.dtype(prec) == np.dtype(np.complex64) else _lib.SSF_C128


def _amp_code(amp):
    return {"edfa": _lib.AMP_EDFA, "ideal": _lib.AMP_IDEAL}.get(amp, _lib.AMP_NONE) if isinstance(amp, str) else _lib.AMP_NONE


def _edfa_noise_power(G, NF, Fc, Fs):
    """devices.py:712-722 -> (G_lin, p_noise)."""
    NF_lin = 10 ** (NF / 10)
    G_lin = 10 ** (G / 10)
    nsp = (G_lin * NF_lin - 1) / (2 * (G_lin - 1))
    return G_lin, (G_lin - 1) * nsp * _H_PLANCK * Fc * Fs


def gaussianComplexNoise(shapeOut, σ2=1.0, seed=None):
    """Complex circular Gaussian noise from numpy's global RNG (optic/dsp/core.py:739-763)."""
    if seed is not None:
        np.random.seed(seed)
    return np.random.normal(0, np.sqrt(σ2 / 2), shapeOut) + 1j * np.random.normal(0, np.sqrt(σ2 / 2), shapeOut)


def _span_noise(nrows, N, p_noise, seed, pairs, dtype):
    """ASE noise of one span in SoA layout.  For the Manakov model the reference
    calls edfa(Ech_x) then edfa(Ech_y), each on a (K, N) block and each re-seeding
    with the same seed (channels.py:444-445) -- reproduced here draw for draw."""
    if pairs:
        K = nrows // 2
        out = np.empty((nrows, N), dtype=dtype)
        out[0::2] = gaussianComplexNoise((K, N), p_noise, seed)
        out[1::2] = gaussianComplexNoise((K, N), p_noise, seed)
        return out
    return np.ascontiguousarray(gaussianComplexNoise((nrows, N), p_noise, seed).astype(dtype))


def _device_seed(seed):
    """Key of the on-device ASE generator (Philox4x32-10): derived from param.seed when given,
    fresh entropy otherwise; never 0 (0 means "no device noise" in the ABI)."""
    if seed is None:
        v = int.from_bytes(os.urandom(8), "little") >> 1
    else:
        v = (int(seed) * 0x9E3779B97F4A7C15 + 0x1234567) & 0x7FFFFFFFFFFFFFFF
    return v or 1


def _captured_spans(save_list, Nspans):
    """Spans that `spanN in saveSpanN` (channels.py:453) would hit, in encounter order."""
    return [s for s in range(1, Nspans + 1) if s in save_list]


def _tqdm(iterable, disable):
    if disable:
        return iterable
    try:
        from tqdm.auto import tqdm
        return tqdm(iterable)
    except Exception:           # tqdm is optional
        return iterable


def _execute(pl, cp, Nspans, save, prgs, noise_fn, want_trace, max_steps_hint):
    """Upload done by the caller; runs all spans, span by span when a progress
    bar or per-span noise is needed, otherwise in one ABI call."""
    lib = pl.lib
    st = _lib.Stats()
    traces = []

    def one(s0, s1, noise):
        tr = None
        if want_trace:
            cap = max_steps_hint * (s1 - s0 + 1)
            hz = np.full(cap, np.nan)
            it = np.zeros(cap, dtype=np.int32)
            lm = np.full(cap * max(cp.maxIter, 1), np.nan)
            tr = _lib.Trace(cap, 0, hz.ctypes.data_as(C.POINTER(C.c_double)),
                            it.ctypes.data_as(C.POINTER(C.c_int32)), lm.ctypes.data_as(C.POINTER(C.c_double)))
        nptr = noise.ctypes.data_as(C.c_void_p) if noise is not None else None
        pl.check(lib.ssf_execute(pl.h, C.byref(cp), s0, s1, nptr, C.byref(st), C.byref(tr) if tr else None))
        if tr:
            n = min(tr.count, tr.capacity)
            traces.append((hz[:n], it[:n], lm[: n * max(cp.maxIter, 1)].reshape(n, max(cp.maxIter, 1)), tr.count))

    if Nspans >= 1:
        if prgs or noise_fn is not None:
            for span in _tqdm(range(1, Nspans + 1), disable=not prgs):
                one(span, span, noise_fn(span) if noise_fn else None)
        else:
            one(1, Nspans, None)
    info = st.as_dict()
    if want_trace:
        info["hz"] = np.concatenate([t[0] for t in traces]) if traces else np.zeros(0)
        info["iters"] = np.concatenate([t[1] for t in traces]) if traces else np.zeros(0, np.int32)
        info["lims"] = [row[~np.isnan(row)] for t in traces for row in t[2]]
        info["trace_truncated"] = any(t[3] > len(t[0]) for t in traces)
    info["pipeline"] = _lib.PIPELINE_NAMES.get(lib.ssf_plan_pipeline(pl.h), "?")
    last_run.clear()
    last_run.update(info)
    return st


def _open_sink(pl, on_dev, N, ncols, nblk, n_captured):
    """Result array of a run with saveSpanN: (N, ncols * nblk), the captured spans streamed into their columns while
    the following span propagates (ssf_set_snapshot_sink); spans that are never reached stay zero (channels.py:375-377)."""
    shape = (N, ncols * nblk)
    if on_dev:
        out = _dev.empty(True, shape, pl.dtype)
        if n_captured < nblk:
            out.set(np.zeros(shape, dtype=pl.dtype))
    else:
        out = np.zeros(shape, dtype=pl.dtype) if n_captured < nblk else np.empty(shape, dtype=pl.dtype)
    if n_captured:
        pl.check(pl.lib.ssf_set_snapshot_sink(pl.h, _dev.out_ptr(out), ncols * nblk, 0))
    return out


def _close_sink(pl, n_captured):
    if n_captured:
        try:
            pl.check(pl.lib.ssf_sync_snapshots(pl.h))
        finally:
            pl.lib.ssf_set_snapshot_sink(pl.h, None, 0, 0)


def _fill_params(model, direction, param, Fs, Nspans, save_arr):
    cp = _lib.Params()
    cp.model, cp.direction = model, direction
    cp.Fs, cp.Fc, cp.alpha, cp.D, cp.gamma = float(Fs), float(param.Fc), float(param.alpha), float(param.D), float(param.gamma)
    cp.Lspan, cp.Nspans, cp.hz = float(param.Lspan), int(Nspans), float(param.hz)
    cp.maxIter = int(getattr(param, "maxIter", 1))
    cp.tol = float(getattr(param, "tol", 0.0))
    cp.nlprMethod = int(bool(getattr(param, "nlprMethod", False)))
    cp.maxNlinPhaseRot = float(getattr(param, "maxNlinPhaseRot", 0.0))
    cp.amp = _amp_code(param.amp)
    cp.NF = float(getattr(param, "NF", 4.5))
    cp.n_save = len(save_arr)
    # device ASE noise: which rows of the seed's noise stream this call's rows are.  A stand-alone call starts at 0; the
    # sharded / coupled drivers (mgpu.py) give every unit / rank its own block so that a shared param.seed does not
    # repeat the same noise in every Monte-Carlo unit or correlate the pairs of a coupled batch across ranks
    cp.rng_row_offset = int(getattr(param, "_rng_row_offset", 0))
    cp.save_spans = save_arr.ctypes.data_as(C.POINTER(C.c_int32)) if len(save_arr) else None
    return cp


# ----------------------------------------------------------------------------
# ssfm
# ----------------------------------------------------------------------------
def ssfm(Ei, param=None, _trace=False, _cpu_seed_policy=False):
    """Split-step Fourier method (symmetric, single-pol.) on the GPU.

    Same parameters as the reference (optic/models/modelsGPU.py:117-278 /
    optic/models/channels.py:112-249): Ltotal [400], Lspan [80], hz [0.5], alpha
    [0.2], D [16], gamma [1.3], Fc [193.1e12], Fs (mandatory), prec [complex128],
    amp ['edfa'], NF [4.5], seed [None], prgsBar [True], saveSpanN
    [[Ltotal // Lspan]], returnParameters [False].

    Returns the field after the last saved span, shape (N,) (or (N, len(saveSpanN))),
    dtype ``param.prec``; ``(Ech, param)`` when ``param.returnParameters``.
    """
    Fs = _require_fs(param)
    for name, dflt in (("Ltotal", 400), ("Lspan", 80), ("hz", 0.5), ("alpha", 0.2), ("D", 16),
                       ("gamma", 1.3), ("Fc", 193.1e12), ("prec", np.complex128), ("amp", "edfa"),
                       ("NF", 4.5), ("seed", None), ("prgsBar", True)):
        setattr(param, name, getattr(param, name, dflt))
    param.saveSpanN = getattr(param, "saveSpanN", [param.Ltotal // param.Lspan])
    param.returnParameters = getattr(param, "returnParameters", False)

    on_dev = _dev.is_device(Ei)
    if not on_dev:
        Ei = np.asarray(Ei)
    N = len(Ei)
    E = Ei.reshape(N)                                   # raises like the reference for ncols > 1
    Nspans = int(np.floor(param.Ltotal / param.Lspan))
    prec = _prec_code(param.prec)
    pl = _get_plan(N, 1, prec)
    in_ptr, _keep = _dev.arg(E, pl.dtype)               # a single row: SoA and the reference's layout coincide

    save_list = list(param.saveSpanN) if param.saveSpanN is not None else []
    captured = _captured_spans(save_list, Nspans)
    save_arr = np.array(captured, dtype=np.int32)
    cp = _fill_params(_lib.MODEL_NLSE, +1, param, Fs, Nspans, save_arr)

    noise_fn = None
    if cp.amp == _lib.AMP_EDFA:
        G = param.alpha * param.Lspan
        assert G > 0, "EDFA gain should be a positive scalar"
        assert param.NF >= 3, "The minimal EDFA noise figure is 3 dB"
        _, p_noise = _edfa_noise_power(G, param.NF, param.Fc, Fs)
        seed = param.seed
        if _cpu_seed_policy:      # draw-for-draw the CPU reference's numpy stream (golden-vector tests)
            def noise_fn(span):
                return _span_noise(1, N, p_noise, seed, False, pl.dtype)
        else:                     # product path: ASE generated on the device, one stream per span
            cp.rng_seed = _device_seed(seed)

    pl.check(pl.lib.ssf_upload(pl.h, in_ptr))
    nsteps = int(np.floor(param.Lspan / param.hz))
    sink = _open_sink(pl, on_dev, N, 1, len(save_list), len(captured)) if save_list else None
    try:
        st = _execute(pl, cp, Nspans, save_arr, param.prgsBar, noise_fn, _trace, nsteps + 1)
    finally:
        if sink is not None:
            _close_sink(pl, len(captured))

    if save_list:                                       # device in -> device out, numpy in -> numpy out
        out = sink.reshape(N) if len(save_list) == 1 else sink
    elif on_dev:
        out = _dev.empty(True, (N,), pl.dtype)
        pl.check(pl.lib.ssf_download(pl.h, out.ptr))
    else:
        res = np.empty((1, N), dtype=pl.dtype)
        pl.check(pl.lib.ssf_download(pl.h, res.ctypes.data_as(C.c_void_p)))
        out = res.reshape(N)
    return (out, param) if param.returnParameters else out


# ----------------------------------------------------------------------------
# manakovSSF / manakovDBP
# ----------------------------------------------------------------------------
def _manakov(Ei, param, direction, _trace, _cpu_seed_policy, _noise, _coupling=None, _units=1):
    t_in = time.perf_counter()
    Fs = _require_fs(param)
    defaults = [("Ltotal", 400), ("Lspan", 80), ("hz", 0.5), ("alpha", 0.2), ("D", 16), ("gamma", 1.3),
                ("Fc", 193.1e12), ("prec", np.complex128), ("amp", "edfa")]
    if direction > 0:
        defaults += [("NF", 4.5)]
    defaults += [("maxIter", 10), ("tol", 1e-5), ("nlprMethod", True), ("maxNlinPhaseRot", 2e-2)]
    if direction > 0:
        defaults += [("seed", None)]
    defaults += [("prgsBar", True)]
    for name, dflt in defaults:
        setattr(param, name, getattr(param, name, dflt))
    param.saveSpanN = getattr(param, "saveSpanN", [param.Ltotal // param.Lspan])
    param.returnParameters = getattr(param, "returnParameters", False)

    on_dev = _dev.is_device(Ei)
    if not on_dev:
        Ei = np.asarray(Ei)
    if Ei.ndim != 2 or Ei.shape[1] % 2 or Ei.shape[1] == 0:
        raise IndexError("manakov models need a 2-D field of shape (N, 2K): columns [x0, y0, x1, y1, ...]")
    N, ncols = Ei.shape
    K = ncols // 2
    save_list = list(param.saveSpanN) if param.saveSpanN is not None else []
    if _units > 1 and (_trace or _coupling is not None or ncols % (2 * _units)):
        raise ValueError("independent units: (N, 2 K units) columns, no trace, no cross-process coupling")
    if save_list and K > 1 and (_units == 1 or K // _units > 1):
        # the reference's snapshot write only broadcasts for K = 1 (channels.py:454-455)
        raise ValueError(f"could not broadcast input array from shape ({N},{K}) into shape ({N},1): "
                         "with more than one polarisation pair set param.saveSpanN = []")
    Nspans = int(np.floor(param.Ltotal / param.Lspan))
    prec = _prec_code(param.prec)
    dev_coupling = False
    if _coupling is not None:                 # rows of ONE reference call spread over several processes (mgpu.run_coupled):
        # an RCCL communicator + a natively split length: the device-resident pipeline all-gathers its partial sums on the plan's
        # stream (ssf_set_coupling_comm, no host in the loop); anything else: the host-driven engine with a reducer callback
        # The ranks must agree on the mode -- one that fell back to the host reducer while the others enqueue ncclAllGather
        # on their streams would leave the job waiting for the RCCL timeout -- so every rank attaches (once) and the outcome
        # is reduced over the communicator before anything runs.
        ch = getattr(_coupling, "h", None)
        attached = False
        bad_comm = None                       # (an error of THIS rank is raised after the collective below: the other ranks are in it)
        if ch is not None and _state["engine"] != _lib.ENGINE_ROCFFT:
            pl = _get_plan(N, ncols, prec)
            if pl.lib.ssf_plan_pipeline(pl.h) == 0:
                rc = pl.lib.ssf_set_coupling_comm(pl.h, ch)
                if rc == -1:                  # SSF_ERR_BAD_ARG: the communicator lives on another device than the plan
                    bad_comm = "run_coupled: " + (pl.lib.ssf_last_error(pl.h) or b"bad argument").decode()
                attached = rc == 0
        try:                                  # 0: attached, 1: this rank falls back to the host reducer, 2: this rank cannot run at all
            worst = float(_coupling.allreduce(np.array([2.0 if bad_comm else 0.0 if attached else 1.0]), "max")[0])
        except Exception:
            if attached:
                pl.lib.ssf_set_coupling_comm(pl.h, None)
            raise
        if worst >= 2.0:                      # every rank leaves here, together
            if attached:
                pl.lib.ssf_set_coupling_comm(pl.h, None)
            raise ValueError(bad_comm or "run_coupled: another rank's communicator does not live on its plan's device")
        any_failed = worst > 0.0
        dev_coupling = attached and not any_failed
        if attached and not dev_coupling:
            pl.lib.ssf_set_coupling_comm(pl.h, None)
        if not dev_coupling:
            pl = _get_plan(N, ncols, prec, engine=_lib.ENGINE_ROCFFT)              # host-driven control flow
    else:
        pl = _get_plan(N, ncols, prec, units=_units)
    try:                                      # (an argument error below must not leave the communicator on the cached plan)
        # the reference's own layout goes over the bus; the (N, 2K) -> (2K, N) conversion runs on the GPU
        in_ptr, _keep = _dev.arg(Ei, pl.dtype)

        captured = _captured_spans(save_list, Nspans)
        save_arr = np.array(captured, dtype=np.int32)
        cp = _fill_params(_lib.MODEL_MANAKOV, direction, param, Fs, Nspans, save_arr)

        noise_fn = None
        if direction > 0 and cp.amp == _lib.AMP_EDFA:
            G = param.alpha * param.Lspan
            assert G > 0, "EDFA gain should be a positive scalar"
            assert param.NF >= 3, "The minimal EDFA noise figure is 3 dB"
            _, p_noise = _edfa_noise_power(G, param.NF, param.Fc, Fs)
            seed = param.seed

            if _noise is not None:        # test hook: caller-supplied noise, (Nspans, 2K, N)
                def noise_fn(span):
                    return np.ascontiguousarray(_noise[span - 1], dtype=pl.dtype)
            elif _cpu_seed_policy:        # draw-for-draw the CPU reference's numpy stream (golden-vector tests)
                def noise_fn(span):
                    return _span_noise(ncols, N, p_noise, seed, True, pl.dtype)
            else:                         # product path: ASE generated on the device (Philox, per-span streams;
                cp.rng_seed = _device_seed(seed)   # x and y rows get INDEPENDENT noise -- the CPU reference re-seeds between its
                #                                    two edfa calls and so adds the same draw to x and y, channels.py:444-445; the cupy
                #                                    twin does not, modelsGPU.py:486-490: statistical parity either way)

    except BaseException:
        if dev_coupling:
            pl.lib.ssf_set_coupling_comm(pl.h, None)
        raise
    logg.info("Running Manakov SSF model on GPU (HIP, %s)..." % ("forward" if direction > 0 else "DBP"))
    if param.nlprMethod:
        hint = 1 << 16
    else:
        hint = int(np.ceil(param.Lspan / param.hz)) + 1
    sink, reducer = None, None
    try:                                      # (whatever fails below, the cached plan keeps no sink, no reducer, no communicator)
        t_up0 = time.perf_counter()
        pl.check(pl.lib.ssf_upload_aos(pl.h, in_ptr))
        t_up = time.perf_counter() - t_up0
        if save_list:
            sink = _open_sink(pl, on_dev, N, ncols, len(save_list), len(captured))
        if _coupling is not None and not dev_coupling:
            def _reduce(_ctx, vals, n, op):   # called by the engine with the partial sums / maxima it is about to use
                try:
                    a = np.ctypeslib.as_array(vals, shape=(n,))
                    a[:] = _coupling.allreduce(a.copy(), "max" if op else "sum")
                    return 0
                except Exception:             # (never let an exception cross the C boundary)
                    return 1
            reducer = _lib.REDUCE_FN(_reduce)
            pl.check(pl.lib.ssf_set_coupling(pl.h, C.cast(reducer, C.c_void_p), None))
        t_ex0 = time.perf_counter()
        st = _execute(pl, cp, Nspans, save_arr, param.prgsBar, noise_fn, _trace, hint)
        t_ex1 = time.perf_counter()
    finally:
        if reducer is not None:
            pl.lib.ssf_set_coupling(pl.h, None, None)
        if dev_coupling:
            pl.lib.ssf_set_coupling_comm(pl.h, None)
        if save_list:
            _close_sink(pl, len(captured))
    for _ in range(int(st.nonconverged_steps)):
        logg.warning(NONCONV_WARNING.format(param.maxIter))

    t_dn = time.perf_counter()
    if save_list:                                       # (N, 2 len(saveSpanN)), on the device when the input was
        out = sink
    elif on_dev:
        out = _dev.empty(True, (N, ncols), pl.dtype)
        pl.check(pl.lib.ssf_download_aos(pl.h, -1, out.ptr))
    else:
        res = np.empty((N, ncols), dtype=pl.dtype)
        pl.check(pl.lib.ssf_download_aos(pl.h, -1, res.ctypes.data_as(C.c_void_p)))
        out = res if (Ei.dtype == pl.dtype) else res.astype(Ei.dtype)     # `Ech = Ei.copy(); Ech[:, 0::2] = ...`
    t_out = time.perf_counter()
    last_run["upload_ms"] = t_up * 1e3                  # wall time of the call's two transfers (host <-> device, or device to device
    last_run["download_ms"] = (t_out - t_dn) * 1e3      # for a DeviceArray), beside device_ms of the spans
    # ... and of the call's other host-side segments: before the upload (parameters, plan), between upload and execute (snapshot sink),
    # the execute call, after it (sink closed, result)
    last_run["host_ms"] = {"before_upload": (t_up0 - t_in) * 1e3, "upload": t_up * 1e3, "sink_open": (t_ex0 - t_up0 - t_up) * 1e3,
                           "execute": (t_ex1 - t_ex0) * 1e3, "sink_close": (t_dn - t_ex1) * 1e3, "download": (t_out - t_dn) * 1e3}
    return (out, param) if param.returnParameters else out


def manakovSSF(Ei, param, _trace=False, _cpu_seed_policy=False, _noise=None, _coupling=None, _units=1):
    """Manakov split-step Fourier model (symmetric, dual-pol.) on the GPU.

    Reference: optic/models/modelsGPU.py:281-511 == optic/models/channels.py:252-468.
    ``Ei``: (N, 2K) complex, columns [x0, y0, x1, y1, ...].  Parameters (defaults):
    Ltotal [400], Lspan [80], hz [0.5], alpha [0.2], D [16], gamma [1.3], Fc
    [193.1e12], Fs (mandatory), prec [complex128], amp ['edfa'], NF [4.5], maxIter
    [10], tol [1e-5], nlprMethod [True], maxNlinPhaseRot [2e-2], prgsBar [True],
    saveSpanN [[Ltotal // Lspan]], seed [None], returnParameters [False].
    """
    return _manakov(Ei, param, +1, _trace, _cpu_seed_policy, _noise, _coupling, _units)


def manakovDBP(Ei, param, _trace=False):
    """Manakov SSF digital back-propagation on the GPU.

    Reference: optic/models/modelsGPU.py:564-772 == optic/dsp/equalization.py:976-1173.
    """
    return _manakov(Ei, param, -1, _trace, False, None)


# ----------------------------------------------------------------------------
# edfa, linearFiberChannel, setPowerforParSSFM
# ----------------------------------------------------------------------------
def edfa(Ei, param=None):
    """Simple EDFA model: gain + ASE noise (optic/models/modelsGPU.py:56-114 == optic/models/devices.py:671-726), one
    element-wise device pass (``ssf_edfa``).  Parameters: G [20 dB], NF [4.5 dB], Fc [193.1e12], Fs (mandatory), seed [None].

    numpy in, numpy out: the noise is the reference's own draw -- ``gaussianComplexNoise(Ei.shape, p_noise, seed)`` from numpy's
    global generator, seeded or not -- added on the device.  DeviceArray (complex128) in, DeviceArray out: the field never leaves
    HBM and the noise is generated there (Philox4x32-10 keyed by ``param.seed``, fresh entropy without one): statistical
    parity, as inside ``ssfm`` / ``manakovSSF``.  The result is complex128 like the reference's (its noise is)."""
    Fs = _require_fs(param)
    G = getattr(param, "G", 20)
    NF = getattr(param, "NF", 4.5)
    Fc = getattr(param, "Fc", 193.1e12)
    seed = getattr(param, "seed", None)
    assert G > 0, "EDFA gain should be a positive scalar"
    assert NF >= 3, "The minimal EDFA noise figure is 3 dB"
    G_lin, p_noise = _edfa_noise_power(G, NF, Fc, Fs)
    lib = _lib.load()
    on_dev = _dev.is_device(Ei)
    shape = tuple(Ei.shape)
    n = int(np.prod(shape))
    ncols = int(shape[1]) if len(shape) > 1 else 1
    in_ptr, _keep = _dev.arg(Ei, np.complex128)
    out = _dev.empty(on_dev, shape, np.complex128)
    if n == 0:
        return out
    if on_dev:
        noise_ptr, dev_seed = None, _device_seed(seed)
    else:
        noise = np.ascontiguousarray(gaussianComplexNoise(shape, p_noise, seed), dtype=np.complex128)
        noise_ptr, dev_seed = noise.ctypes.data_as(C.c_void_p), 0
    _lib.raise_for(lib, None, lib.ssf_edfa(_state["device"], n, ncols, float(G_lin), float(p_noise), dev_seed,
                                           int(getattr(param, "_rng_row_offset", 0)), in_ptr, noise_ptr, _dev.out_ptr(out)))
    return out


def linearFiberChannel(Ei, param):
    """Linear fiber channel (optic/models/channels.py:30-109) as one fused FFT . H . IFFT on the GPU.  Parameters: L [50],
    alpha [0.2], D [17], Fc [193.1e12], Fs (mandatory), returnParameters [False].

    The field crosses the ABI in the reference's own (N, modes) layout (the transposition runs on the device); a complex128
    DeviceArray stays in HBM and a DeviceArray comes back.  Like the reference's, the result is complex128 whatever the input's
    precision (its operator is: ``fft(Ei) * exp(...)``, channels.py:97); a complex64 input is cast up first, so the forward
    transform -- which numpy >= 2 runs in single precision for such an input -- is evaluated in double here."""
    Fs = _require_fs(param)
    param.L = getattr(param, "L", 50)
    param.alpha = getattr(param, "alpha", 0.2)
    param.D = getattr(param, "D", 17)
    param.Fc = getattr(param, "Fc", 193.1e12)
    param.returnParameters = getattr(param, "returnParameters", False)
    on_dev = _dev.is_device(Ei)
    if not on_dev:
        Ei = np.asarray(Ei)
    N = Ei.shape[0]
    one_d = Ei.ndim == 1
    E2 = Ei.reshape(N, -1)
    nm = E2.shape[1]
    pl = _get_plan(N, nm, _lib.SSF_C128)
    in_ptr, _keep = _dev.arg(E2, np.complex128)
    pl.check(pl.lib.ssf_upload_aos(pl.h, in_ptr))
    pl.check(pl.lib.ssf_linear_channel(pl.h, float(Fs), float(param.Fc), float(param.alpha), float(param.D),
                                       float(param.L), None, None))
    Eo = _dev.empty(on_dev, (N, nm), np.complex128)
    pl.check(pl.lib.ssf_download_aos(pl.h, -1, _dev.out_ptr(Eo)))
    if nm == 1 or one_d:
        Eo = Eo.reshape(N)
    return (Eo, param) if param.returnParameters else Eo


def _ols_block(K):
    """Transform size of the device's overlap-save kernel for a K-tap filter (rx_pipeline.h: fir_nfft)."""
    nfft = 256
    while nfft < 8 * K and nfft < 2048:
        nfft *= 2
    while nfft < 3 * K and nfft < 4096:
        nfft *= 2
    while nfft < 2 * K and nfft < 8192:
        nfft *= 2
    while nfft < K:
        nfft *= 2
    return nfft


_EDC_F = {}


def _edc_filter(param, Fs):
    """(NfilterCoeffs, Nfft, H) as optic/dsp/equalization.py:85-110 derives them.  A receiver loop asks for the same link call after
    call: the last few designs are kept by their parameters (a dozen small numpy calls are 20 us in front of a 0.1 ms call)."""
    L = getattr(param, "L", 50)
    D = getattr(param, "D", 16)
    Fc = getattr(param, "Fc", 193.1e12)
    Rs = getattr(param, "Rs", 32e9)
    NfilterCoeffs = getattr(param, "NfilterCoeffs", None)
    Nfft = getattr(param, "Nfft", None)
    try:
        key = (float(L), float(D), float(Fc), float(Rs), float(Fs), NfilterCoeffs, Nfft)
        hit = _EDC_F.get(key)
    except (TypeError, ValueError):
        key = hit = None
    if hit is not None:
        return hit
    res = _edc_design(L, D, Fc, Rs, Fs, NfilterCoeffs, Nfft)
    if key is not None:
        if len(_EDC_F) >= 8:
            _EDC_F.pop(next(iter(_EDC_F)))
        _EDC_F[key] = res
    return res


def _edc_design(L, D, Fc, Rs, Fs, NfilterCoeffs, Nfft):
    c_kms = 299792458.0 / 1e3
    lam = c_kms / Fc
    b2 = -(D * lam**2) / (2 * np.pi * c_kms)
    if NfilterCoeffs is None:
        NfilterCoeffs = int(2 * np.ceil(6.67 * np.abs(b2) * L * Rs**2 * (Fs / Rs)))
    if Nfft is None:
        Nfft = 2 ** int(np.ceil(np.log2(NfilterCoeffs)))
    w = 2 * np.pi * Fs * np.fft.fftfreq(NfilterCoeffs)
    return NfilterCoeffs, Nfft, np.exp(-1j * (b2 / 2) * (w**2) * L)


_EDC_H = {}


def _edc_block_response(K, Nfft, Hf):
    """core.py:1015-1020: centred impulse response, zero-padded to the block size, back to frequency.  A receiver loop designs the
    same filter call after call: the last few responses are kept (keyed by the K frequency samples themselves)."""
    key = (K, Nfft, Hf.tobytes())
    H = _EDC_H.get(key)
    if H is None:
        h = np.pad(np.fft.fftshift(np.fft.ifft(Hf)), (0, Nfft - K), mode="constant")
        H = np.ascontiguousarray(np.fft.fft(h), dtype=np.complex128)
        if len(_EDC_H) >= 8:
            _EDC_H.pop(next(iter(_EDC_H)))
        _EDC_H[key] = H
    return H


_OLS_MAX_TAPS = 4096          # longer impulse responses are convolved segment by segment on the device (ssf_fir_long)


def blockwiseFFTConv(x, h, NFFT=None, freqDomainFilter=False):
    """Blockwise convolution by the overlap-and-save FFT method (optic/dsp/core.py:973-1046; cupy twin optic/dsp/coreGPU.py:81-170):
    the 'same'-mode linear convolution of the 1-D signal ``x`` with the filter ``h`` -- an impulse response, or with
    ``freqDomainFilter=True`` a frequency response centred at DC (turned into its centred impulse response as core.py:1015-1016
    does) -- delay ``(len(h) - 1) // 2`` compensated, ``len(x)`` samples.  ``NFFT`` must not be smaller than the filter (the
    reference's error); otherwise the block size of an overlap-save evaluation does not change the convolution it computes, and
    the device kernel uses its own (LDS transforms of up to 8192 points; filters of more than 4096 taps segment by segment,
    ceil(len(h) / 4096) passes over the signal).  numpy in, numpy out (real when ``x`` has no imaginary part, core.py:1043-1046);
    a complex128 DeviceArray in, a DeviceArray out, never through the host."""
    from . import rx as _rx
    on_dev = _dev.is_device(x)
    xs = x if on_dev else np.asarray(x)
    if xs.ndim != 1:
        raise ValueError("blockwiseFFTConv: x must be one-dimensional")
    h = np.asarray(h)
    sigLen, K = xs.shape[0], len(h)
    if NFFT is None:
        NFFT = 2 ** int(np.ceil(np.log2(np.max([sigLen, K]))))
    if NFFT < K:
        raise ValueError("FFT size is smaller than filter length")
    ht = np.fft.fftshift(np.fft.ifft(h)) if freqDomainFilter else h
    y = _rx._conv_shift(xs.reshape(sigLen, 1), ht, (K - 1) // 2, sigLen, on_dev).reshape(-1)
    if on_dev:
        return y
    return y if np.any(np.iscomplex(xs)) else y.real


def _edc_long(sigIn, sig2, one_d, on_dev, K, Hf):
    """edc with an impulse response of more than _OLS_MAX_TAPS taps: blockwiseFFTConv(x, Hf, freqDomainFilter=True) of the
    reference (optic/dsp/core.py:973-1046) for every column -- the centred impulse response (delay (K - 1) // 2), convolved on the
    device segment by segment (ssf_fir_long); a DeviceArray stays on the device."""
    from . import rx as _rx
    h = np.fft.fftshift(np.fft.ifft(Hf))                          # core.py:1015-1016
    acc = _rx._conv_shift(sig2, h, (K - 1) // 2, sig2.shape[0], on_dev)
    if on_dev:
        return acc.reshape(-1) if one_d else acc
    ncols = acc.shape[1]
    res = acc if np.iscomplexobj(sigIn) else acc.real
    if np.iscomplexobj(sigIn):
        for m_ in range(ncols):
            if not np.any(np.iscomplex(sig2[:, m_])):
                res[:, m_] = res[:, m_].real
    res = res.astype(sigIn.dtype, copy=False)
    return res.flatten() if one_d else res


def edc(sigIn, param):
    """Electronic chromatic dispersion compensation on the GPU (optic/dsp/equalization.py:36-122).

    Same overlap-and-save FFT filter as the reference (optic/dsp/core.py:973-1046): the
    ``NfilterCoeffs``-tap frequency response ``exp(-j beta2/2 w^2 L)`` is turned into a zero-padded
    impulse response on the host (a few hundred taps), every mode's blocks are transformed,
    filtered and stitched in one HIP launch.  Parameters: L [50], D [16], Fc [193.1e12], Fs
    (mandatory), Rs [32e9], NfilterCoeffs [None], Nfft [None] (must be a power of two >= 16)."""
    Fs = _require_fs(param)
    on_dev = _dev.is_device(sigIn)
    if not on_dev:
        sigIn = np.asarray(sigIn)
    one_d = sigIn.ndim == 1
    sig2 = sigIn.reshape(sigIn.size, 1) if one_d else sigIn
    K, Nfft, Hf = _edc_filter(param, Fs)
    if Nfft < K:
        raise ValueError("FFT size is smaller than filter length")
    logg.info("Running CD compensation...")
    logg.info(f"CD filter length: {K} taps, FFT size: {Nfft}")
    # The block size of an overlap-save evaluation does not change the linear convolution it computes, only how much of every
    # transform is overlap: the reference's default Nfft (the next power of two above the filter length) leaves as little as a
    # fifth of each block as output.  The device kernel takes powers of two in [16, 8192]; the block is chosen like firFilter's
    # (rx_pipeline.h: fir_nfft): eight times the taps up to 2048 points, 4096 above 682 taps, 8192 above 2048.  Filters that leave less than
    # half of an 8192-point block as output (> 4096 taps) are split into segments of the impulse response, on the device
    # (_edc_long): the reference takes any length (core.py:973-1046).
    if K > _OLS_MAX_TAPS:
        return _edc_long(sigIn, sig2, one_d, on_dev, K, Hf)
    Nfft = _ols_block(K)
    H = _edc_block_response(K, Nfft, Hf)
    in_ptr, _keep = _dev.arg(sig2, np.complex128)
    out = _dev.empty(on_dev, sig2.shape, np.complex128)
    lib = _lib.load()
    rc = lib.ssf_overlap_save(_state["device"], sig2.shape[0], sig2.shape[1], _lib.SSF_C128, int(Nfft), int(K),
                              H.ctypes.data_as(C.c_void_p), in_ptr, _dev.out_ptr(out))
    _lib.raise_for(lib, None, rc)
    if on_dev:
        return out.reshape(-1) if one_d else out
    res = out if np.iscomplexobj(sigIn) else out.real        # core.py:1043-1046
    if np.iscomplexobj(sigIn):
        for m in range(sig2.shape[1]):                       # ... which looks at the values: a column without any
            if not np.any(np.iscomplex(sig2[:, m])):         # imaginary part is filtered as a real signal
                res[:, m] = res[:, m].real
    res = res.astype(sigIn.dtype, copy=False)                # sigOut = np.zeros(sigIn.shape, dtype=sigIn.dtype)
    return res.flatten() if one_d else res


def nlinPhaseRot(Ex, Ey, Pch, γ):
    """Nonlinear phase shift per unit length of the Manakov step (optic/models/channels.py:471-493, cupy twin
    optic/models/modelsGPU.py:514-535): ``((8/9) γ (Pch + Ex conj(Ex) + Ey conj(Ey)) / 2).real`` on the GPU.
    ``manakovSSF`` evaluates this inside its column kernel; this is the function by itself.  Arrays of any common
    shape (numpy or DeviceArray, complex128; complex64 input is evaluated in double and returned as float32)."""
    dev = _dev
    lib = _lib.load()
    on_dev = dev.is_device(Ex)
    single = (Ex.dtype == np.complex64)
    shape = Ex.shape
    if tuple(Ey.shape) != tuple(shape) or tuple(Pch.shape) != tuple(shape):
        raise ValueError("nlinPhaseRot: Ex, Ey and Pch must have the same shape")
    if dev.is_device(Pch):
        if Pch.dtype != np.float64:
            raise TypeError("nlinPhaseRot: a device-resident Pch must be float64 (the real part)")
        pp, kp = Pch.ptr, Pch
    else:
        kp = np.ascontiguousarray(np.real(Pch), dtype=np.float64)
        pp = kp.ctypes.data_as(C.c_void_p)
    px, kx = dev.arg(Ex, np.complex128)
    py, ky = dev.arg(Ey, np.complex128)
    out = dev.empty(on_dev, shape, np.float64)
    n = int(np.prod(shape))
    rc = lib.ssf_nlin_phase_rot(_state["device"], n, float(γ), px, py, pp, dev.out_ptr(out))
    _lib.raise_for(lib, None, rc)
    del kx, ky, kp
    return out.astype(np.float32) if (single and not on_dev) else out


def convergenceCondition(Ex_fd, Ey_fd, Ex_conv, Ey_conv):
    """Convergence measure of the trapezoidal iteration (optic/models/channels.py:496-519, cupy twin
    optic/models/modelsGPU.py:538-561): ``sqrt(|Ex_fd - Ex_conv|² + |Ey_fd - Ey_conv|²) / sqrt(|Ex_conv|² + |Ey_conv|²)``
    (Frobenius norms over the whole arrays), reduced on the GPU; returns a Python float."""
    dev = _dev
    lib = _lib.load()
    shape = tuple(Ex_fd.shape)
    for a in (Ey_fd, Ex_conv, Ey_conv):
        if tuple(a.shape) != shape:
            raise ValueError("convergenceCondition: all four arrays must have the same shape")
    ptrs, keep = [], []
    for a in (Ex_fd, Ey_fd, Ex_conv, Ey_conv):
        p_, k_ = dev.arg(a, np.complex128)
        ptrs.append(p_)
        keep.append(k_)
    lim = C.c_double()
    rc = lib.ssf_convergence_condition(_state["device"], int(np.prod(shape)), *ptrs, C.byref(lim))
    _lib.raise_for(lib, None, rc)
    return float(lim.value)


def signalPower(x):
    """Total power of x (optic/dsp/core.py:69-84)."""
    return np.sum(np.mean(x * np.conj(x), axis=0).real)


def setPowerforParSSFM(sig, powers):
    """Per-pair launch-power normalisation for K > 1 batches (modelsGPU.py:775-788)."""
    powers_lin = (10 ** (np.asarray(powers) / 10) * 1e-3).repeat(2) / 2
    for i in np.arange(0, sig.shape[1], 2):
        for k in range(2):
            sig[:, i + k] = np.sqrt(powers_lin[i] / signalPower(sig[:, i + k])) * sig[:, i + k]
            print("power mode %d: %.2f dBm" % (i + k, 10 * np.log10(signalPower(sig[:, i + k]) / 1e-3)))
    return sig
