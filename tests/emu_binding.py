"""ctypes binding of tests/emu/libssf_emu.so (CPU emulator of the fused-engine kernels).
Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from opticommpy_amd import _lib

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
_emu = None


def load():
    global _emu
    if _emu is None:
        subprocess.check_call(["make", "-C", EMU_DIR, "-s"])
        _emu = C.CDLL(os.environ.get("SSF_EMU_LIB") or os.path.join(EMU_DIR, "libssf_emu.so"))      # (SSF_EMU_LIB: a variant build, A/B experiments)
        _emu.emu_run.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(_lib.Params), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.POINTER(_lib.Stats), C.POINTER(_lib.Trace),
                                 C.POINTER(C.c_long)]
        _emu.emu_linear_channel.argtypes = [C.c_int64, C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_void_p, C.c_void_p]
        _emu.emu_supported.argtypes = [C.c_int64, C.c_int]
        _emu.emu_overlap_save.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _emu.emu_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
        _emu.emu_gauss.argtypes = [C.c_int64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p]
        _emu.emu_rows_lin.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p] + [C.c_double] * 4
        _emu.emu_split.argtypes = [C.c_int64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return _emu


def _amp(amp):
    return {"edfa": 2, "ideal": 1}.get(amp, 0) if isinstance(amp, str) else 0


def run(func, Ei, cfg, noise=None, max_steps=4096, trace=True):
    """Run ssfm / manakovSSF / manakovDBP of a cfg dict on the emulator.  Returns (out, info)."""
    emu = load()
    dt = np.complex64 if cfg.get("prec") == "complex64" else np.complex128
    prec = 0 if dt == np.complex64 else 1
    Ei = np.asarray(Ei)
    N = Ei.shape[0]
    E2 = Ei.reshape(N, -1)
    ncols = E2.shape[1]
    soa = np.ascontiguousarray(E2.T, dtype=dt)
    Ltotal, Lspan = cfg.get("Ltotal", 400), cfg.get("Lspan", 80)
    Nspans = int(np.floor(Ltotal / Lspan))
    save = cfg.get("saveSpanN", [Ltotal // Lspan] if func != "ssfm_final" else [])
    captured = [s for s in range(1, Nspans + 1) if s in save]
    save_arr = np.array(captured, dtype=np.int32)
    p = _lib.Params()
    p.model = 0 if func.startswith("ssfm") else 1
    p.direction = -1 if func == "manakovDBP" else 1
    p.Fs, p.Fc = cfg["Fs"], cfg.get("Fc", 193.1e12)
    p.alpha, p.D, p.gamma = cfg.get("alpha", 0.2), cfg.get("D", 16), cfg.get("gamma", 1.3)
    p.Lspan, p.Nspans, p.hz = Lspan, Nspans, cfg.get("hz", 0.5)
    p.maxIter, p.tol = cfg.get("maxIter", 10), cfg.get("tol", 1e-5)
    p.nlprMethod = int(cfg.get("nlprMethod", True)) if p.model == 1 else 0
    p.maxNlinPhaseRot = cfg.get("maxNlinPhaseRot", 2e-2)
    p.amp, p.NF = _amp(cfg.get("amp", "edfa")), cfg.get("NF", 4.5)
    p.rng_seed = int(cfg.get("_rng_seed", 0))
    p.rng_row_offset = int(cfg.get("_rng_row_offset", 0))
    p.n_save = len(save_arr)
    p.save_spans = save_arr.ctypes.data_as(C.POINTER(C.c_int32)) if len(save_arr) else None
    out = np.empty_like(soa)
    snaps = np.zeros((max(len(captured), 1), ncols, N), dtype=dt)
    st = _lib.Stats()
    hz = np.full(max_steps, np.nan)
    it = np.zeros(max_steps, dtype=np.int32)
    lm = np.full(max_steps * p.maxIter, np.nan)
    tr = _lib.Trace(max_steps, 0, hz.ctypes.data_as(C.POINTER(C.c_double)), it.ctypes.data_as(C.POINTER(C.c_int32)),
                    lm.ctypes.data_as(C.POINTER(C.c_double)))
    launches = C.c_long(0)
    nz = None
    if noise is not None:
        nz = np.ascontiguousarray(noise, dtype=dt)
    rc = emu.emu_run(N, ncols, prec, C.byref(p), soa.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                     snaps.ctypes.data_as(C.c_void_p), nz.ctypes.data_as(C.c_void_p) if nz is not None else None,
                     C.byref(st), C.byref(tr) if trace else None, C.byref(launches))
    assert rc == 0, f"emu_run rc={rc}"
    n = int(tr.count)
    info = st.as_dict()
    info.update(hz=hz[:n], iters=it[:n], lims=[r[~np.isnan(r)] for r in lm[: n * p.maxIter].reshape(n, p.maxIter)],
                launches=launches.value, snaps=snaps[: st.n_snapshots])
    return out, info


def rows_lin(rows, hzh, lin_a, lin_b, w_scale, in_place=False):
    """(nrows, N) -> ifft(fft(row) * exp((lin_a + 1j * lin_b * w**2) * hzh)), w = w_scale * fftfreq(N): the one-launch
    linear step of the general-length engine (short 5-smooth lengths)."""
    emu = load()
    x = np.ascontiguousarray(rows)
    out = x.copy() if in_place else np.empty_like(x)
    src = out if in_place else x
    rc = emu.emu_rows_lin(x.shape[1], x.shape[0], 0 if x.dtype == np.complex64 else 1, src.ctypes.data_as(C.c_void_p),
                          out.ctypes.data_as(C.c_void_p), hzh, lin_a, lin_b, w_scale)
    assert rc == 0, f"emu_rows_lin rc={rc}"
    return out


def linear_channel(Ei, Fs, Fc, alpha, D, L, dtype=np.complex128):
    emu = load()
    Ei = np.asarray(Ei)
    N = Ei.shape[0]
    soa = np.ascontiguousarray(Ei.reshape(N, -1).T, dtype=dtype)
    out = np.empty_like(soa)
    rc = emu.emu_linear_channel(N, soa.shape[0], 0 if dtype == np.complex64 else 1, Fs, Fc, alpha, D, L,
                                soa.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0, f"emu_linear_channel rc={rc}"
    return out.T.reshape(Ei.shape)


def philox(ctr, key):
    """ctr: 4 x uint32, key: 2 x uint32 -> 4 x uint32 (Philox4x32-10)."""
    emu = load()
    out = (C.c_uint32 * 4)()
    emu.emu_philox(ctr[0] | (ctr[1] << 32), ctr[2] | (ctr[3] << 32), key[0] | (key[1] << 32), out)
    return [int(x) for x in out]


def gauss(n, row, span, seed, sigma):
    emu = load()
    re, im = np.empty(n), np.empty(n)
    emu.emu_gauss(n, row, span, seed, sigma, re.ctypes.data_as(C.c_void_p), im.ctypes.data_as(C.c_void_p))
    return re + 1j * im


def edc(sigIn, param):
    """The product's edc host logic (opticommpy_amd.models.edc) with the kernel run on the emulator."""
    from opticommpy_amd import models
    emu = load()
    sigIn = np.asarray(sigIn)
    one_d = sigIn.ndim == 1
    sig2 = sigIn.reshape(sigIn.size, 1) if one_d else sigIn
    K, Nfft, Hf = models._edc_filter(param, param.Fs)
    Nfft = models._ols_block(K)                      # (the product's block size: models.edc)
    h = np.pad(np.fft.fftshift(np.fft.ifft(Hf)), (0, Nfft - K), mode="constant")
    H = np.ascontiguousarray(np.fft.fft(h), dtype=np.complex128)
    x = np.ascontiguousarray(sig2, dtype=np.complex128)
    out = np.empty_like(x)
    lg = int(np.log2(Nfft))
    assert emu.emu_overlap_save(x.shape[0], x.shape[1], lg, K, H.ctypes.data_as(C.c_void_p),
                                x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    res = out if np.iscomplexobj(sigIn) else out.real
    res = res.astype(sigIn.dtype, copy=False)
    return res.flatten() if one_d else res


def edfa(Ei, G_lin, p_noise, seed=0, row0=0, noise=None):
    """ssf_edfa's kernel (rx_kernels.h: optics_body, OPT_EDFA) on the emulator: Ei sqrt(G_lin) + noise (supplied, or Philox when seed != 0)."""
    e = load()
    e.emu_optics.argtypes = EmuRxBackend._OPTICS
    x = np.ascontiguousarray(Ei, dtype=np.complex128)
    out = np.empty_like(x)
    nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.complex128)
    sigma = float(np.sqrt(p_noise / 2)) if (nz is None and seed) else 0.0
    rc = e.emu_optics(0, x.size, x.shape[1] if x.ndim > 1 else 1, float(np.sqrt(G_lin)), sigma, int(seed), int(row0),
                      x.ctypes.data_as(C.c_void_p), None if nz is None else nz.ctypes.data_as(C.c_void_p),
                      out.ctypes.data_as(C.c_void_p), None)
    assert rc == 0, f"emu_optics rc={rc}"
    return out


class EmuRxBackend:
    """Drop-in for opticommpy_amd.rx._backend: the receiver pipeline (rx_pipeline.h + rx_kernels.h) on the
    CPU emulator instead of the GPU.  Same marshalling code (opticommpy_amd/rx.py) in front of it."""

    def __init__(self):
        e = load()
        e.emu_fir.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        e.emu_delay.argtypes = [C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        e.emu_decimate.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        e.emu_rx_run.argtypes = [C.c_int, C.c_int64, C.c_int, C.POINTER(_lib.RxParams), C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_double), C.c_void_p]
        self.e = e

    @staticmethod
    def _check(rc):
        if rc == -1:
            raise ValueError("bad argument")
        if rc == -6:
            raise RuntimeError("unsupported")
        assert rc == 0, f"emulator rc={rc}"

    def fir(self, N, ncols, ntaps, taps, x, out):
        self._check(self.e.emu_fir(N, ncols, ntaps, taps, x, out))

    def fir_long(self, inLen, outLen, ncols, ntaps, taps, shift, x, out):
        self.e.emu_fir_long.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        self._check(self.e.emu_fir_long(inLen, outLen, ncols, ntaps, taps, shift, x, out))

    def delay(self, N, delay, Fs, x, out):
        self._check(self.e.emu_delay(N, float(delay), float(Fs), x, out))

    def decimate(self, N, ncols, SpSin, dec, x, out):
        sd = (C.c_int32 * ncols)()
        self._check(self.e.emu_decimate(N, ncols, int(SpSin), int(dec), x, out, sd))
        return list(sd)

    _OPTICS = [C.c_int, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

    def pbs(self, N, ncols, theta, E, Ex, Ey):
        self.e.emu_optics.argtypes = self._OPTICS
        self._check(self.e.emu_optics(1, N, ncols, float(np.cos(theta)), float(np.sin(theta)), 0, 0, E, None, Ex, Ey))

    def hybrid(self, N, Es, Elo, Eo):
        self.e.emu_optics.argtypes = self._OPTICS
        self._check(self.e.emu_optics(2, N, 1, 0.0, 0.0, 0, 0, Es, Elo, Eo, None))

    def rx_chain(self, N, p, Es, Elo, taps, SpSin, dec, H, K, nfft, out):
        self.e.emu_rx_chain.argtypes = [C.c_int64, C.POINTER(_lib.RxParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int32)]
        self._check(self.e.emu_rx_chain(N, C.byref(p), Es, Elo, taps.ctypes.data_as(C.c_void_p), len(taps), SpSin, dec,
                                        H.ctypes.data_as(C.c_void_p), int(K), int(nfft), out, None))

    def rx(self, mode, N, nmodes, p, in0, lo, un, out):
        self._check(self.e.emu_rx_run(mode, N, nmodes, C.byref(p), in0, lo,
                                      C.cast(un, C.POINTER(C.c_double)) if un is not None else None, out))


class EmuTxBackend:
    """Drop-in for opticommpy_amd.wdm_tx._backend (the transmitter's signal path on the CPU emulator)."""

    def __init__(self):
        e = load()
        e.emu_wdm_tx.argtypes = [C.POINTER(_lib.TxParams), C.c_void_p] + [C.POINTER(C.c_double)] * 4 + [C.c_void_p, C.POINTER(C.c_double)]
        self.e = e

    def wdm_tx(self, p, symbols, taps, phi, amp, deltaF, out_ptr, power):
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None   # noqa: E731
        rc = self.e.emu_wdm_tx(C.byref(p), symbols.ctypes.data_as(C.c_void_p), dp(taps), dp(phi), dp(amp), dp(deltaF), out_ptr, dp(power))
        assert rc == 0, f"emulator rc={rc}"
