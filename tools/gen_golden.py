#!/usr/bin/env python3
"""Generate golden vectors for the SSFM hot path by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); never on the GPU box.
The reference needs ``numba`` (absent here): a throw-away stub whose ``njit`` is
the identity decorator is registered first -- every ``@njit`` function on the
hot path is a plain numpy expression.

Output: tests/golden/<case>.npz, each holding
    Ei        input field                                  (N,) or (N, 2K)
    out       reference output
    cfg       JSON: function name + every parameter that was set
    hz, iters per-step step size and iteration count       (Manakov/DBP only)
    lims      all convergence values, flattened            (Manakov/DBP only)
    margin    min |lim - tol| / tol over the run           (Manakov/DBP only)
    extra_*   case-specific extras (e.g. linear-channel output, edfa noise)

    long_*    long runs (1001 steps of a lossy span and more): the input is NOT stored but regenerated from the
              seeded recipe in cfg["synth"]; stored are the per-step iteration list, every lim value, the
              output decimated by cfg["dec"], its per-column power and a seeded random projection of the full
              output (so an error anywhere in the array shows), see long_vectors()

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [rx|rx_edges|tx|bfc|mixed|rx_chain20|chain|long|long20|notebook|cfg3|long_c3 [128|64|merge]|units45 [log2n]]
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):     # the BLAS thread pool makes scipy.linalg.norm
    os.environ.setdefault(_v, "1")                                            # 40x slower on a busy box

_nb = types.ModuleType("numba")


def _identity_decorator(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


_nb.njit = _nb.jit = _identity_decorator
_nb.prange = range
_nb_typed = types.ModuleType("numba.typed")
_nb_typed.List = list
_nb.typed = _nb_typed
sys.modules["numba"] = _nb
sys.modules["numba.typed"] = _nb_typed
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

import optic.models.channels as ref_ch  # noqa: E402
from optic.dsp.equalization import manakovDBP as ref_dbp  # noqa: E402
from optic.utils import parameters  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def synth_field(N, ncols, seed, p_dbm, dtype=np.complex128):
    """SURVEY.md 8d recipe: band-limited complex Gaussian, each column P/2."""
    rng = np.random.default_rng(seed)
    E = (rng.normal(size=(N, ncols)) + 1j * rng.normal(size=(N, ncols))) / np.sqrt(2)
    F = np.fft.fft(E, axis=0)
    F[N // 4: 3 * N // 4, :] = 0
    E = np.fft.ifft(F, axis=0)
    p_lin = 10 ** (p_dbm / 10) * 1e-3
    E = E * np.sqrt((p_lin / 2) / np.mean(np.abs(E) ** 2, axis=0))
    return E.astype(dtype)


class Tracer:
    """Wraps reference.convergenceCondition / np.exp-free bookkeeping to record
    lim values and infer per-step iteration counts and step sizes."""

    def __init__(self, mod, name="convergenceCondition"):
        self.mod, self.name = mod, name
        self.lims = []

    def __enter__(self):
        self.orig = getattr(self.mod, self.name)

        def wrapped(*a):
            v = self.orig(*a)
            self.lims.append(float(v))
            return v

        setattr(self.mod, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.mod, self.name, self.orig)


def mk_param(**kw):
    p = parameters()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def cfg_json(func, kw):
    d = {"func": func}
    for k, v in kw.items():
        if k == "prec":
            v = np.dtype(v).name
        d[k] = v
    return json.dumps(d)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **arrs)
    return os.path.getsize(path)


def split_iters(lims, tol, maxIter):
    """lims (flat) -> per-step iteration counts, replaying the reference's
    break rule (channels.py:429-434)."""
    iters, n = [], 0
    for v in lims:
        n += 1
        if v < tol or n == maxIter:
            iters.append(n)
            n = 0
    assert n == 0
    return iters


def run_manakov(name, func, Ei, kw, extra=None):
    import optic.dsp.equalization as ref_eq
    mod = ref_ch if func == "manakovSSF" else ref_eq
    p = mk_param(**kw)
    with Tracer(mod) as tr:
        out = ref_ch.manakovSSF(Ei, p) if func == "manakovSSF" else ref_dbp(Ei, p)
    lims = np.array(tr.lims)
    iters = np.array(split_iters(tr.lims, p.tol, p.maxIter))
    margin = float(np.min(np.abs(lims - p.tol) / p.tol))
    arrs = dict(Ei=Ei, out=out, cfg=cfg_json(func, kw), iters=iters, lims=lims, margin=margin)
    if extra:
        arrs.update(extra)
    sz = save(name, **arrs)
    print(f"{name:34s} steps={len(iters):4d} iters={iters.sum():5d} "
          f"(min {iters.min()} max {iters.max()}) margin={margin:.2e} out={out.dtype}{out.shape} {sz/1024:.0f} KiB")
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    base = dict(Fc=193.1e12, prgsBar=False)

    # ---------------- ssfm: the reference's own three TestSSFM setups ----------------
    rng = np.random.default_rng(8)
    sig = 1e-3 * (rng.normal(size=4096) + 1j * rng.normal(size=4096))
    kw = dict(Ltotal=80, Lspan=80, hz=1, alpha=0.2, D=16, gamma=0, Fs=64e9, amp=None, **base)
    out = ref_ch.ssfm(sig, mk_param(**kw))
    lin = ref_ch.linearFiberChannel(sig, mk_param(L=80, alpha=0.2, D=16, Fc=193.1e12, Fs=64e9))
    np.testing.assert_allclose(out, lin, atol=1e-12)
    save("ssfm_ref_gamma0", Ei=sig, out=out, cfg=cfg_json("ssfm", kw), extra_linear=lin)

    rng = np.random.default_rng(9)
    sig = rng.normal(size=4096) + 1j * rng.normal(size=4096)
    kw = dict(Ltotal=80, Lspan=80, hz=1, alpha=0, D=0, gamma=1.3, Fs=64e9, amp=None, **base)
    out = ref_ch.ssfm(sig, mk_param(**kw))
    kw0 = dict(kw, gamma=0)
    out0 = ref_ch.ssfm(sig, mk_param(**kw0))
    assert not np.allclose(np.abs(np.fft.fft(out)), np.abs(np.fft.fft(out0)))
    save("ssfm_ref_spm", Ei=sig, out=out, cfg=cfg_json("ssfm", kw), extra_gamma0=out0)

    rng = np.random.default_rng(10)
    sig = rng.normal(size=4096) + 1j * rng.normal(size=4096)
    kw = dict(Ltotal=80, Lspan=80, hz=1, alpha=0, D=16, gamma=1.3, Fs=64e9, amp=None, **base)
    out = ref_ch.ssfm(sig, mk_param(**kw))
    save("ssfm_ref_power", Ei=sig, out=out, cfg=cfg_json("ssfm", kw))

    # ---------------- ssfm: BASELINE config-1 shape + spans/amp variants ----------------
    E = synth_field(4096, 1, 1, 0.0).reshape(-1)
    E = E * np.sqrt(2)  # single pol carries the whole 0 dBm
    kw = dict(Ltotal=50, Lspan=50, hz=0.5, alpha=0.2, D=16, gamma=1.3, Fs=512e9, amp=None, **base)
    save("ssfm_c1shape", Ei=E, out=ref_ch.ssfm(E, mk_param(**kw)), cfg=cfg_json("ssfm", kw))
    kw = dict(Ltotal=100, Lspan=50, hz=1.0, alpha=0.2, D=16, gamma=1.3, Fs=512e9, amp="ideal", **base)
    save("ssfm_2span_ideal", Ei=E, out=ref_ch.ssfm(E, mk_param(**kw)), cfg=cfg_json("ssfm", kw))
    kw = dict(Ltotal=90, Lspan=40, hz=0.7, alpha=0.25, D=17, gamma=1.5, Fs=256e9, amp=None, **base)
    save("ssfm_truncating", Ei=E.reshape(-1, 1), out=ref_ch.ssfm(E.reshape(-1, 1), mk_param(**kw)),
         cfg=cfg_json("ssfm", kw))
    # non power-of-two length (a typical SpS x Nsymb product)
    E3 = synth_field(3000, 1, 4, 3.0).reshape(-1) * np.sqrt(2)
    kw = dict(Ltotal=30, Lspan=30, hz=0.5, alpha=0.2, D=16, gamma=1.3, Fs=512e9, amp="ideal", **base)
    save("ssfm_n3000", Ei=E3, out=ref_ch.ssfm(E3, mk_param(**kw)), cfg=cfg_json("ssfm", kw))
    # edfa with a fixed seed: under the numba stub the noise is numpy's legacy global RNG
    kw = dict(Ltotal=40, Lspan=20, hz=1.0, alpha=0.2, D=16, gamma=1.3, Fs=512e9, amp="edfa", NF=5.0,
              seed=77, **base)
    save("ssfm_edfa_seed77", Ei=E, out=ref_ch.ssfm(E, mk_param(**kw)), cfg=cfg_json("ssfm", kw))

    # ---------------- manakovSSF ----------------
    N = 1024
    mk = dict(alpha=0.2, D=16, gamma=1.3, Fs=512e9, maxIter=10, tol=1e-5, **base)
    E0 = synth_field(N, 2, 20, 0.0)
    E8 = synth_field(N, 2, 21, 8.4)
    E13 = synth_field(N, 2, 22, 13.0)
    E13k2 = synth_field(N, 4, 23, 13.0)
    E8k2 = synth_field(N, 4, 24, 8.4)

    run_manakov("mk_fix_p0_none", "manakovSSF", E0,
                dict(mk, Ltotal=20, Lspan=20, hz=0.5, nlprMethod=False, amp=None))
    run_manakov("mk_fix_p8_ideal_2span", "manakovSSF", E8,
                dict(mk, Ltotal=40, Lspan=20, hz=0.5, nlprMethod=False, amp="ideal", saveSpanN=[1, 2]))
    run_manakov("mk_fix_p13_ideal_k2", "manakovSSF", E13k2,
                dict(mk, Ltotal=20, Lspan=10, hz=0.25, nlprMethod=False, amp="ideal", saveSpanN=[]))
    run_manakov("mk_fix_nonint_lastspan", "manakovSSF", E8,
                dict(mk, Ltotal=25, Lspan=10, hz=0.3, nlprMethod=False, amp=None, saveSpanN=[2]))
    run_manakov("mk_fix_p13_hz1", "manakovSSF", E13,
                dict(mk, Ltotal=20, Lspan=20, hz=1.0, nlprMethod=False, amp="ideal", saveSpanN=[]))
    run_manakov("mk_adp_p8_none", "manakovSSF", E8,
                dict(mk, Ltotal=20, Lspan=20, hz=0.5, nlprMethod=True, maxNlinPhaseRot=2e-2, amp=None))
    run_manakov("mk_adp_p13_ideal_2span", "manakovSSF", E13,
                dict(mk, Ltotal=20, Lspan=10, hz=0.5, nlprMethod=True, maxNlinPhaseRot=2e-2, amp="ideal",
                     saveSpanN=[1, 2]))
    run_manakov("mk_adp_p8_k2", "manakovSSF", E8k2,
                dict(mk, Ltotal=10, Lspan=10, hz=0.5, nlprMethod=True, maxNlinPhaseRot=2e-2, amp=None,
                     saveSpanN=[]))
    run_manakov("mk_maxiter_hit", "manakovSSF", E13,
                dict(mk, Ltotal=5, Lspan=5, hz=1.0, nlprMethod=False, amp=None, maxIter=3, tol=1e-9,
                     saveSpanN=[]))
    # defaults path: only Fs and the lengths given (amp='edfa' default would add noise -> amp None)
    run_manakov("mk_defaults", "manakovSSF", E0,
                dict(Fs=512e9, Ltotal=10, Lspan=5, amp=None, prgsBar=False))
    # complex64 field + prec=complex64
    E8c = E8.astype(np.complex64)
    run_manakov("mk_fix_p8_c64", "manakovSSF", E8c,
                dict(mk, Ltotal=20, Lspan=20, hz=0.5, nlprMethod=False, amp="ideal", prec=np.complex64,
                     saveSpanN=[]))
    run_manakov("mk_adp_p8_c64", "manakovSSF", E8c,
                dict(mk, Ltotal=10, Lspan=10, hz=0.5, nlprMethod=True, amp=None, prec=np.complex64))
    # larger N, few steps (exercises the two-pass FFT decomposition at a non-trivial size)
    E8big = synth_field(4096, 2, 25, 8.4)
    run_manakov("mk_fix_p8_n4096", "manakovSSF", E8big,
                dict(mk, Ltotal=2, Lspan=2, hz=0.25, nlprMethod=False, amp="ideal", saveSpanN=[]))
    # non power-of-two
    E8n = synth_field(1500, 2, 26, 8.4)
    run_manakov("mk_fix_p8_n1500", "manakovSSF", E8n,
                dict(mk, Ltotal=10, Lspan=10, hz=0.5, nlprMethod=False, amp="ideal", saveSpanN=[]))
    # edfa with seed (CPU reference: same seed every span, x and y share the noise)
    run_manakov("mk_edfa_seed5", "manakovSSF", E8,
                dict(mk, Ltotal=20, Lspan=10, hz=0.5, nlprMethod=False, amp="edfa", NF=4.5, seed=5,
                     saveSpanN=[1, 2]))

    # ---------------- manakovDBP ----------------
    run_manakov("dbp_fix_hz10", "manakovDBP", E8,
                dict(mk, Ltotal=80, Lspan=40, hz=10, nlprMethod=False, amp="edfa", saveSpanN=[]))
    run_manakov("dbp_fix_hz05_none", "manakovDBP", E13,
                dict(mk, Ltotal=10, Lspan=10, hz=0.5, nlprMethod=False, amp=None))
    run_manakov("dbp_adp_ideal", "manakovDBP", E8,
                dict(mk, Ltotal=20, Lspan=10, hz=0.5, nlprMethod=True, maxNlinPhaseRot=2e-2, amp="ideal",
                     saveSpanN=[1, 2]))
    # forward -> DBP round trip
    fkw = dict(mk, Ltotal=20, Lspan=10, hz=0.5, nlprMethod=False, amp="ideal", saveSpanN=[])
    fwd = ref_ch.manakovSSF(E8, mk_param(**fkw))
    back = run_manakov("dbp_roundtrip", "manakovDBP", fwd, dict(fkw), extra=dict(extra_orig=E8))
    rt = np.linalg.norm(back - E8) / np.linalg.norm(E8)
    print(f"round trip rel-L2 = {rt:.2e}")

    # ---------------- edc (overlap-save chromatic dispersion compensation) ----------------
    from optic.dsp.equalization import edc as ref_edc
    rng = np.random.default_rng(50)
    sig = (rng.normal(size=(4096, 2)) + 1j * rng.normal(size=(4096, 2))) / np.sqrt(2)
    for name, x, kw in (
            ("edc_2mode_default", sig, dict(L=50, D=16, Fc=193.1e12, Fs=64e9, Rs=32e9)),
            ("edc_1d_long_link", sig[:3000, 0].copy(), dict(L=800, D=17, Fc=193.1e12, Fs=64e9, Rs=32e9)),
            ("edc_given_nfft", sig, dict(L=100, D=16, Fc=193.1e12, Fs=64e9, Rs=32e9, NfilterCoeffs=45, Nfft=256)),
            ("edc_real_input", sig[:2048].real.copy(), dict(L=20, D=16, Fc=193.1e12, Fs=64e9, Rs=32e9))):
        out = ref_edc(x, mk_param(**kw))
        save(name, Ei=x, out=out, cfg=cfg_json("edc", kw))
        print(f"{name:24s} in {x.dtype}{x.shape} out {out.dtype}{out.shape}")
    rx_vectors()
    tx_vectors()


def rx_vectors():
    """receiver front-end, FIR filtering, decimation"""
    import optic.dsp.core as ref_core
    import optic.models.devices as ref_dev
    rng = np.random.default_rng(60)
    x2 = (rng.normal(size=(3000, 2)) + 1j * rng.normal(size=(3000, 2))) / np.sqrt(2)
    xr = rng.normal(size=2048)
    h255 = ref_core.lowPassFIR(30e9, 128e9, 255, "rect")
    hg = ref_core.lowPassFIR(20e9, 128e9, 64, "gauss")          # even tap count
    t = np.arange(-64, 65) / 16.0
    hrrc = ref_core.rrcFilterTaps(t, 0.1, 1.0)
    hcx = (rng.normal(size=33) + 1j * rng.normal(size=33)) / 8
    for name, h, x in (("rx_fir_lp255_2mode", h255, x2), ("rx_fir_gauss64_real1d", hg, xr),
                       ("rx_fir_rrc129_1d", hrrc, x2[:, 0].copy()), ("rx_fir_complex_taps", hcx, x2),
                       ("rx_fir_taps_longer_than_signal", h255, x2[:100, 0].copy())):
        out = ref_core.firFilter(h, x)
        save(name, h=h, Ei=x, out=out, cfg=cfg_json("firFilter", {}))
        print(f"{name:32s} in {x.dtype}{x.shape} taps {len(h)} out {out.dtype}{out.shape}")
    save("rx_lowpassfir", rect=h255, gauss=hg, cfg=cfg_json("lowPassFIR", dict(rect=[30e9, 128e9, 255], gauss=[20e9, 128e9, 64])))

    # decimate: 16 -> 2 samples per symbol, 2 modes, and a 1-D real signal 8 -> 1
    sps = 16
    sym = rng.choice([-3.0, -1.0, 1.0, 3.0], size=(256, 2)) + 1j * rng.choice([-3.0, -1.0, 1.0, 3.0], size=(256, 2))
    up = np.zeros((256 * sps, 2), dtype=complex)
    up[5::sps] = sym                                               # sampling phase 5
    pulse = ref_core.rrcFilterTaps(np.arange(-8 * sps, 8 * sps + 1) / sps, 0.2, 1.0)
    shaped = ref_core.firFilter(pulse / np.max(np.abs(pulse)), up)
    for name, x, kw in (("rx_decimate_16to2", shaped, dict(SpSin=16, SpSout=2)),
                        ("rx_decimate_8to1_real1d", shaped[::2, 0].real.copy(), dict(SpSin=8, SpSout=1))):
        out = ref_core.decimate(x, mk_param(**kw))
        save(name, Ei=x, out=out, cfg=cfg_json("decimate", kw))
        print(f"{name:32s} in {x.dtype}{x.shape} out {out.dtype}{out.shape}")

    # delaySignal / iqMixing
    xs = x2[:2500, 0].copy()
    for name, kw in (("rx_delay_frac", dict(delay=0.37 / 128e9, Fs=128e9)), ("rx_delay_neg3p2", dict(delay=-3.2 / 128e9, Fs=128e9)),
                     ("rx_delay_zero", dict(delay=0.0, Fs=128e9))):
        out = ref_core.delaySignal(xs, kw["delay"], kw["Fs"])
        save(name, Ei=xs, out=out, cfg=cfg_json("delaySignal", kw))
    for name, kw in (("rx_iqmix_imbalance", dict(ampImb=1.5, phaseImb=0.2, timeSkew=0.0, Fs=128e9)),
                     ("rx_iqmix_skew", dict(ampImb=0.0, phaseImb=0.0, timeSkew=2.5e-12, Fs=128e9))):
        out = ref_core.iqMixing(xs, mk_param(**kw))
        save(name, Ei=xs, out=out, cfg=cfg_json("iqMixing", kw))

    # photodiode / balancedPD / hybrid / coherent receivers: deterministic settings (no shot / thermal noise)
    Es = synth_field(4096, 2, 61, 0.0)
    Fs = 128e9
    tt = np.arange(4096) / Fs
    Elo = np.sqrt(10e-3) * np.exp(1j * (2 * np.pi * 150e6 * tt + 0.3))
    pd_quiet = dict(Fs=Fs, B=30e9, shotNoise=False, thermalNoise=False)
    for name, kw in (("rx_pd_ideal", dict(ideal=True)), ("rx_pd_bandlimited", dict(pd_quiet)),
                     ("rx_pd_saturating_gauss", dict(pd_quiet, currentSaturation=True, IpdSat=8e-4, fType="gauss", N=128, R=0.8))):
        out = ref_dev.photodiode(Es[:, 0].copy(), mk_param(**kw))
        save(name, Ei=Es[:, 0], out=out, cfg=cfg_json("photodiode", kw))
    out = ref_dev.photodiode(Es.copy(), mk_param(**pd_quiet))
    save("rx_pd_two_modes", Ei=Es, out=out, cfg=cfg_json("photodiode", pd_quiet))
    out = ref_dev.balancedPD(Es[:, 0].copy(), Es[:, 1].copy(), mk_param(**pd_quiet))
    save("rx_bpd_bandlimited", Ei=Es, out=out, cfg=cfg_json("balancedPD", pd_quiet))
    out = ref_dev.opticalHybrid2x4(Es[:, 0].copy(), Elo)
    save("rx_hybrid", Ei=Es[:, 0], Elo=Elo, out=out, cfg=cfg_json("opticalHybrid2x4", {}))
    fe1 = dict(Fs=Fs, ampImb=0.5, phaseImb=0.05, timeSkew=1e-12)
    out = ref_dev.coherentReceiver(Es[:, 0].copy(), Elo, mk_param(**fe1), mk_param(**pd_quiet))
    save("rx_coh_single_pol", Ei=Es[:, 0], Elo=Elo, out=out, cfg=cfg_json("coherentReceiver", dict(fe=fe1, pd=pd_quiet)))
    for name, fe, pd in (
            ("rx_pdm_default_pd_ideal", dict(Fs=Fs), dict(ideal=True)),
            ("rx_pdm_bandlimited", dict(Fs=Fs), pd_quiet),
            ("rx_pdm_impaired", dict(Fs=Fs, polRotation=0.4, pdl=1.2, polDelay=3e-12, ampImbX=0.8, phaseImbX=0.1,
                                     timeSkewX=2e-12, ampImbY=-0.5, phaseImbY=-0.07, timeSkewY=-1e-12),
             dict(pd_quiet, R=0.9, N=201))):
        out = ref_dev.pdmCoherentReceiver(Es.copy(), Elo, mk_param(**fe), mk_param(**pd))
        save(name, Ei=Es, Elo=Elo, out=out, cfg=cfg_json("pdmCoherentReceiver", dict(fe=fe, pd=pd)))
        print(f"{name:32s} out {out.dtype}{out.shape}")
    # seeded noise: the reference's own draw order (np.random.seed(seed); shot normals; thermal normals per photodiode)
    pdn = dict(Fs=Fs, B=30e9, seed=11)
    out = ref_dev.photodiode(Es[:, 0].copy(), mk_param(**pdn))
    np.random.seed(11)
    u_shot = np.random.normal(0, 1, 4096)
    u_th = np.random.normal(0, 1, 4096)
    save("rx_pd_noise_seed11", Ei=Es[:, 0], out=out, extra_shot=u_shot, extra_thermal=u_th, cfg=cfg_json("photodiode", pdn))



def rx_edge_vectors():
    """Receiver-side argument edges: delaySignal with NFFT != 1024 (even, not a power of two, None),
    balancedPD on (N, M) fields, pdmCoherentReceiver with paramPD.Fs != paramFE.Fs."""
    import optic.dsp.core as ref_core
    import optic.models.devices as ref_dev
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(62)
    xs = (rng.normal(size=10000) + 1j * rng.normal(size=10000)) / np.sqrt(2)
    for name, x, kw in (("rx_delay_nfft256", xs[:2500].copy(), dict(delay=0.37 / 128e9, Fs=128e9, NFFT=256)),
                        ("rx_delay_nfft2048_neg", xs[:2500].copy(), dict(delay=-5.3 / 128e9, Fs=128e9, NFFT=2048)),
                        ("rx_delay_nfft1000", xs[:3000].copy(), dict(delay=2.25 / 128e9, Fs=128e9, NFFT=1000)),
                        ("rx_delay_nfft_none", xs, dict(delay=1.6 / 128e9, Fs=128e9, NFFT=None)),
                        ("rx_delay_nfft_none_real", xs.real.copy(), dict(delay=-0.4 / 128e9, Fs=128e9, NFFT=None))):
        out = ref_core.delaySignal(x, kw["delay"], kw["Fs"], kw["NFFT"])
        save(name, Ei=x, out=out, cfg=cfg_json("delaySignal", kw))
        print(f"{name:32s} in {x.dtype}{x.shape} out {out.dtype}{out.shape}")
    Es4 = np.concatenate([synth_field(4096, 2, 63, 0.0), synth_field(4096, 2, 64, -3.0)], axis=1)
    Fs = 128e9
    pd_quiet = dict(Fs=Fs, B=30e9, shotNoise=False, thermalNoise=False)
    out = ref_dev.balancedPD(Es4[:, :2].copy(), Es4[:, 2:].copy(), mk_param(**pd_quiet))
    save("rx_bpd_two_modes_each", Ei=Es4, out=out, cfg=cfg_json("balancedPD2d", pd_quiet))
    print(f"rx_bpd_two_modes_each            out {out.dtype}{out.shape}")
    Es = synth_field(4096, 2, 61, 0.0)
    tt = np.arange(4096) / Fs
    Elo = np.sqrt(10e-3) * np.exp(1j * (2 * np.pi * 150e6 * tt + 0.3))
    fe = dict(Fs=Fs, polRotation=0.3, polDelay=2e-12, timeSkewX=1.5e-12, timeSkewY=-1e-12)
    pd = dict(pd_quiet, Fs=1.5 * Fs, N=129)                     # the photodiode model designed at another rate (devices.py:331-353)
    out = ref_dev.pdmCoherentReceiver(Es.copy(), Elo, mk_param(**fe), mk_param(**pd))
    save("rx_pdm_two_sampling_rates", Ei=Es, Elo=Elo, out=out, cfg=cfg_json("pdmCoherentReceiver", dict(fe=fe, pd=pd)))
    print(f"rx_pdm_two_sampling_rates        out {out.dtype}{out.shape}")
    fe1 = dict(Fs=Fs, ampImb=0.5, timeSkew=1e-12)
    out = ref_dev.coherentReceiver(Es[:, 0].copy(), Elo, mk_param(**fe1), mk_param(**pd))
    save("rx_coh_two_sampling_rates", Ei=Es[:, 0], Elo=Elo, out=out, cfg=cfg_json("coherentReceiver", dict(fe=fe1, pd=pd)))


def tx_vectors():
    """WDM transmitter"""
    import optic.comm.modulation as ref_mod
    import optic.dsp.core as ref_core
    import optic.models.devices as ref_dev
    import optic.models.tx as ref_tx
    save("tx_gray_maps", qam4=ref_mod.grayMapping(4, "qam"), qam16=ref_mod.grayMapping(16, "qam"),
         qam64=ref_mod.grayMapping(64, "qam"), psk8=ref_mod.grayMapping(8, "psk"), pam4=ref_mod.grayMapping(4, "pam"),
         cfg=cfg_json("grayMapping", {}))
    pulses = {}
    for name, kw in (("rrc", dict(pulseType="rrc", SpS=16, nFilterTaps=1024, rollOff=0.01)),
                     ("rrc_odd", dict(pulseType="rrc", SpS=8, nFilterTaps=513, rollOff=0.25)),
                     ("rc", dict(pulseType="rc", SpS=4, nFilterTaps=64, rollOff=0.5)),
                     ("nrz", dict(pulseType="nrz", SpS=16)), ("rect", dict(pulseType="rect", SpS=8))):
        pulses[name] = ref_core.pulseShape(mk_param(**kw))
    save("tx_pulses", cfg=cfg_json("pulseShape", {}), **pulses)
    rng = np.random.default_rng(70)
    u = (rng.normal(size=2000) + 1j * rng.normal(size=2000)) * 0.4
    lo = np.exp(1j * rng.normal(size=2000) * 0.1)
    save("tx_iqm", u=u, lo=lo, out=ref_dev.iqm(lo, u), out_scalar_lo=ref_dev.iqm(1.0, u), cfg=cfg_json("iqm", {}))
    save("tx_phase_noise", out=ref_core.phaseNoise(100e3, 3000, 1 / 512e9, seed=5), cfg=cfg_json("phaseNoise", dict(lw=100e3, N=3000, Ts=1 / 512e9, seed=5)))
    for name, kw in (
            ("tx_wdm_qam16_5ch_1pol", dict(M=16, nBits=2048, SpS=16, nChannels=5, nPolModes=1, seed=123, prgsBar=False)),
            ("tx_wdm_qam64_4ch_2pol_linewidth", dict(M=64, nBits=3072, SpS=8, nChannels=4, nPolModes=2, seed=7, laserLinewidth=100e3,
                                                     powerPerChannel=[-3, -1.5, 0, 1], pulseRollOff=0.1, nFilterTaps=513,
                                                     wdmGridSpacing=37.5e9, prgsBar=False)),
            ("tx_wdm_shaped_nrz_2pol", dict(M=16, nBits=1024, SpS=16, nChannels=1, nPolModes=2, seed=3, pulseType="nrz",
                                            probDist="maxwell-boltzmann", shapingFactor=0.05, mzmScale=0.25, prgsBar=False)),
            ("tx_wdm_psk_3ch", dict(M=8, constType="psk", nBits=1536, SpS=4, nChannels=3, nPolModes=1, seed=11, Rs=10e9,
                                    wdmGridSpacing=12.5e9, nFilterTaps=128, pulseRollOff=0.2, prgsBar=False))):
        sig, symb, par = ref_tx.simpleWDMTx(mk_param(**kw))
        save(name, out=sig, symb=symb, freqGrid=par.wdmFreqGrid, pmf=par.pmf, cfg=cfg_json("simpleWDMTx", kw))
        print(f"{name:36s} sig {sig.dtype}{sig.shape} symb {symb.shape}")


def projection(out, seed=4242):
    """(ncols,) complex: sum_n out[n, c] * r[n] with a seeded unit-variance complex vector r"""
    rng = np.random.default_rng(seed)
    r = (rng.normal(size=out.shape[0]) + 1j * rng.normal(size=out.shape[0])) / np.sqrt(2)
    return out.astype(np.complex128).T @ r


def run_long(name, func, synth, kw, dec=64):
    """One long reference run; the input comes from synth_field(*synth)."""
    import optic.dsp.equalization as ref_eq
    import time
    N, ncols, seed, p_dbm = synth
    Ei = synth_field(N, ncols, seed, p_dbm, np.dtype(kw.get("prec", np.complex128)).type)
    mod = ref_ch if func == "manakovSSF" else ref_eq
    p = mk_param(**kw)
    t0 = time.time()
    with Tracer(mod) as tr:
        out = ref_ch.manakovSSF(Ei, p) if func == "manakovSSF" else ref_dbp(Ei, p)
    lims = np.array(tr.lims)
    iters = np.array(split_iters(tr.lims, p.tol, p.maxIter), dtype=np.int8)
    margin = float(np.min(np.abs(lims - p.tol) / p.tol))
    cfg = json.loads(cfg_json(func, kw))
    cfg["synth"], cfg["dec"] = list(synth), dec
    o = out.astype(np.complex128)
    sz = save(name, cfg=json.dumps(cfg), iters=iters, lims=lims, margin=margin, out_dec=out[::dec].copy(),
              out_power=np.sum(np.abs(o) ** 2, axis=0), out_proj=projection(out))
    changes = int(np.count_nonzero(np.diff(iters.astype(int))))
    print(f"{name:26s} steps={len(iters):5d} iters={int(iters.sum()):6d} (min {iters.min()} max {iters.max()}, {changes} changes) "
          f"margin={margin:.2e} {sz/1024:.0f} KiB  {time.time()-t0:.0f} s", flush=True)
    return out


def long_vectors():
    """Long runs (identical per-step iteration counts across the 3 -> 2 crossover of a lossy span)."""
    os.makedirs(OUT, exist_ok=True)
    c2 = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
              Ltotal=80, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    run_long("long_c2_n14", "manakovSSF", (1 << 14, 2, 2, 8.4), dict(c2))
    run_long("long_c2_n16", "manakovSSF", (1 << 16, 2, 2, 8.4), dict(c2))
    run_long("long_adp_n14", "manakovSSF", (1 << 14, 2, 12, 8.4), dict(c2, nlprMethod=True, maxNlinPhaseRot=2e-3))
    run_long("long_k2_n14", "manakovSSF", (1 << 14, 4, 13, 11.4), dict(c2))
    run_long("long_dbp_n14", "manakovDBP", (1 << 14, 2, 14, 5.0), dict(c2, amp="edfa"))
    run_long("long_n48000", "manakovSSF", (48000, 2, 15, 8.4), dict(c2, Ltotal=40, Lspan=40))
    # the reference's own complex64 path against its complex128 result over BASELINE config 3's 10 spans
    # (10 010 steps): the yardstick for the single-precision gate (both runs: complex64 input samples)
    for lg in (14, 16):
        synth = (1 << lg, 2, 3, 8.4)
        kw = dict(c2, Ltotal=800, saveSpanN=[1, 2, 4, 10])
        Ei = synth_field(*synth, np.complex64)
        import time
        t0 = time.time()
        o128 = ref_ch.manakovSSF(Ei.astype(np.complex128), mk_param(**dict(kw, prec=np.complex128))).astype(np.complex128)
        o64 = ref_ch.manakovSSF(Ei, mk_param(**dict(kw, prec=np.complex64)))
        assert o64.dtype == np.complex64 and o128.shape == (1 << lg, 8)
        dev, pr = [], []
        for i in range(4):
            a, b = o64[:, 2 * i:2 * i + 2].astype(np.complex128), o128[:, 2 * i:2 * i + 2]
            dev.append(np.linalg.norm(a - b) / np.linalg.norm(b))
            pr.append(np.sum(np.abs(a) ** 2) / np.sum(np.abs(b) ** 2))
        cfg = json.loads(cfg_json("manakovSSF", kw))
        cfg["synth"], cfg["dec"] = list(synth), 64
        save(f"long_c64drift_n{lg}", cfg=json.dumps(cfg), spans=np.array([1, 2, 4, 10]), ref_c64_rel_l2=np.array(dev),
             ref_c64_power_ratio=np.array(pr), out128_dec=o128[::64].copy(), out128_proj=projection(o128),
             out128_power=np.sum(np.abs(o128) ** 2, axis=0))
        print(f"long_c64drift_n{lg}: reference complex64 vs complex128 after 1/2/4/10 spans: rel-L2 "
              + " ".join(f"{x:.2e}" for x in dev) + "  power ratio - 1 " + " ".join(f"{x-1:+.2e}" for x in pr)
              + f"  {time.time()-t0:.0f} s", flush=True)


def notebook_vectors():
    """The reference's published GPU benchmark (examples/benchmarck_GPU_processing.ipynb cells 8 - 10: Fs 128 GS/s, adaptive step,
    maxIter 5) at two of its signal lengths: 2e5 samples over one full 50 km span, and its top size 2e6 = 2^7 x 5^6 -- the length
    that needs the mixed-radix COLUMN stage on the device -- over the first 12 km; amp='ideal' (deterministic)."""
    os.makedirs(OUT, exist_ok=True)
    nb = dict(Fs=128e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, hz=0.5, maxIter=5, tol=1e-5, nlprMethod=True, maxNlinPhaseRot=2e-2,
              prgsBar=False, amp="ideal", saveSpanN=[])
    run_long("long_nb_n200000", "manakovSSF", (200000, 2, 71, 1.0), dict(nb, Ltotal=50, Lspan=50), dec=100)
    run_long("long_nb_n2000000", "manakovSSF", (2000000, 2, 71, 1.0), dict(nb, Ltotal=12, Lspan=12), dec=1000)


def long20_vector():
    """BASELINE config 2 at its full size: 2^20 samples, one 80 km span, 1001 steps (about 15 minutes of reference time)."""
    os.makedirs(OUT, exist_ok=True)
    c2 = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
              Ltotal=80, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    run_long("long_c2_n20", "manakovSSF", (1 << 20, 2, 2, 8.4), dict(c2), dec=512)


def cfg3_vector(steps=6):
    """BASELINE config 3's own field (2^22 samples, seed 3, 8.4 dBm, complex64 samples) through the reference for `steps`
    passes of the step loop, in complex128 (the samples cast up) and in the reference's complex64 mode (complex64 input +
    prec=complex64): the base the full-length single-precision tests stand on.  About 1 minute."""
    import time
    os.makedirs(OUT, exist_ok=True)
    N, dec = 1 << 22, 2048
    synth = (N, 2, 3, 8.4)
    kw = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
              Ltotal=(steps - 0.5) * 0.08, Lspan=(steps - 0.5) * 0.08, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    E64 = synth_field(*synth, np.complex64)
    arrs = {}
    for tag, Ei, prec in (("128", E64.astype(np.complex128), np.complex128), ("64", E64, np.complex64)):
        t0 = time.time()
        p = mk_param(**dict(kw, prec=prec))
        with Tracer(ref_ch) as tr:
            out = ref_ch.manakovSSF(Ei, p)
        assert out.dtype == np.dtype(prec) and out.shape == (N, 2)
        iters = np.array(split_iters(tr.lims, p.tol, p.maxIter), dtype=np.int8)
        assert len(iters) == steps
        o = out.astype(np.complex128)
        arrs.update({"iters" + tag: iters, "lims" + tag: np.array(tr.lims), "out%s_dec" % tag: out[::dec].copy(),
                     "out%s_power" % tag: np.sum(np.abs(o) ** 2, axis=0), "out%s_proj" % tag: projection(out)})
        print(f"cfg3 complex{tag}: iters {iters.tolist()}  {time.time()-t0:.0f} s", flush=True)
        if tag == "128":
            o128 = o
        else:
            arrs["ref_c64_rel_l2"] = float(np.linalg.norm(o - o128) / np.linalg.norm(o128))
    cfg = json.loads(cfg_json("manakovSSF", kw))
    cfg["synth"], cfg["dec"], cfg["steps"] = list(synth), dec, steps
    sz = save("wl_cfg3_n22", cfg=json.dumps(cfg), **arrs)
    print(f"wl_cfg3_n22: reference complex64 vs complex128 after {steps} steps: {arrs['ref_c64_rel_l2']:.2e}  {sz/1024:.0f} KiB", flush=True)


def long_c3_vector(which="both"):
    """BASELINE config 3's own field (2^22 samples, seed 3, 8.4 dBm, complex64 samples) through the reference for ONE FULL
    SPAN (80 km, hz 0.08: 1001 passes of the step loop, through the 3 -> 2 iteration crossover), in complex128 (samples cast
    up) and in the reference's complex64 mode.  About an hour per precision: `which` = "128" /
    "64" writes long_c3_n22_c128.part.npz / _c64.part.npz so that both can run side by side; "merge" joins them into
    tests/golden/long_c3_n22.npz."""
    import time
    os.makedirs(OUT, exist_ok=True)
    N, dec = 1 << 22, 2048
    synth = (N, 2, 3, 8.4)
    kw = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
              Ltotal=80, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    part = lambda tag: os.path.join(OUT, "..", "..", "gpurun_out", "long_c3_n22_c%s.part.npz" % tag)
    if which == "merge":
        arrs = {}
        for tag in ("128", "64"):
            with np.load(part(tag)) as d:
                arrs.update({k: d[k] for k in d.files})
        # the reference's complex64 deviation from its complex128 result on what is stored (decimated output, projection)
        a, b = arrs["out64_dec"].astype(np.complex128), arrs["out128_dec"]
        arrs["ref_c64_rel_l2_dec"] = float(np.linalg.norm(a - b) / np.linalg.norm(b))
        cfg = json.loads(cfg_json("manakovSSF", kw))
        cfg["synth"], cfg["dec"], cfg["steps"] = list(synth), dec, int(len(arrs["iters128"]))
        sz = save("long_c3_n22", cfg=json.dumps(cfg), **arrs)
        print(f"long_c3_n22: steps {cfg['steps']}  iters128 {int(arrs['iters128'].sum())} iters64 {int(arrs['iters64'].sum())}  "
              f"reference complex64 vs complex128 (decimated) {arrs['ref_c64_rel_l2_dec']:.2e}  {sz/1024:.0f} KiB", flush=True)
        return
    E64 = synth_field(*synth, np.complex64)
    for tag, prec in (("128", np.complex128), ("64", np.complex64)):
        if which not in ("both", tag):
            continue
        Ei = E64.astype(prec)
        t0 = time.time()
        p = mk_param(**dict(kw, prec=prec))
        with Tracer(ref_ch) as tr:
            out = ref_ch.manakovSSF(Ei, p)
        assert out.dtype == np.dtype(prec) and out.shape == (N, 2)
        iters = np.array(split_iters(tr.lims, p.tol, p.maxIter), dtype=np.int8)
        o = out.astype(np.complex128)
        np.savez(part(tag), **{"iters" + tag: iters, "lims" + tag: np.array(tr.lims), "out%s_dec" % tag: out[::dec].copy(),
                               "out%s_power" % tag: np.sum(np.abs(o) ** 2, axis=0), "out%s_proj" % tag: projection(out)})
        print(f"long_c3 complex{tag}: steps {len(iters)} iters {int(iters.sum())} "
              f"({int(np.count_nonzero(np.diff(iters.astype(int))))} changes)  {time.time()-t0:.0f} s", flush=True)


def chain_vector():
    """The notebook chain end to end, in the order of examples/test_WDM_transmission.ipynb (cells 10, 14,
    18, 20, 22, 23): simpleWDMTx(seed) -> manakovSSF -> basicLaserModel (LO) -> pdmCoherentReceiver -> firFilter (matched filter)
    -> decimate -> edc, every stage the REFERENCE's own function fed with the previous stage's output.  simpleWDMTx's defaults
    give 240 000 samples (60 000 bits, 16-QAM, 16 samples per symbol); the channel is two 50 km spans with amp='ideal' (the
    deterministic amplifier) and the adaptive step of the notebook.  Stored: a projection of the symbols, every stage's decimated output and a
    seeded projection of all of it, the channel's iteration list, the final output in full (30 000 x 2)."""
    import time
    import optic.dsp.core as ref_core
    import optic.models.devices as ref_dev
    import optic.models.tx as ref_tx
    from optic.dsp.equalization import edc as ref_edc
    os.makedirs(OUT, exist_ok=True)
    t0 = time.time()
    tx = dict(M=16, Rs=32e9, SpS=16, pulseType="rrc", nFilterTaps=1024, pulseRollOff=0.01, powerPerChannel=-2, nChannels=5,
              Fc=193.1e12, laserLinewidth=100e3, wdmGridSpacing=37.5e9, nPolModes=2, nBits=60000, seed=123, prgsBar=False)
    paramTx = mk_param(**tx)
    sigTx, symbTx, paramTx = ref_tx.simpleWDMTx(paramTx)
    Fs = paramTx.Rs * paramTx.SpS
    ch = dict(Ltotal=100, Lspan=50, alpha=0.2, D=16, gamma=1.3, Fc=paramTx.Fc, hz=0.5, maxIter=5, tol=1e-5, nlprMethod=True,
              maxNlinPhaseRot=2e-2, prgsBar=False, Fs=Fs, amp="ideal", saveSpanN=[])
    pch = mk_param(**ch)
    with Tracer(ref_ch) as tr:
        sigCh = ref_ch.manakovSSF(sigTx, pch)
    iters = np.array(split_iters(tr.lims, pch.tol, pch.maxIter), dtype=np.int8)
    chIndex = int(np.floor(paramTx.nChannels / 2))
    lo = dict(P=10, lw=100e3, RIN_var=0, Ns=len(sigCh), Fs=Fs, seed=789, freqShift=float(paramTx.wdmFreqGrid[chIndex]) - 128e6)
    sigLO = ref_dev.basicLaserModel(mk_param(**lo))
    fe = dict(Fs=Fs, polRotation=np.pi / 3, pdl=0, polDelay=3 / paramTx.Rs, phaseImbX=0.0, phaseImbY=0.0, ampImbX=0, ampImbY=0)
    pd = dict(B=paramTx.Rs, Fs=Fs, ideal=True, seed=1011)
    sigRx = ref_dev.pdmCoherentReceiver(sigCh, sigLO, mk_param(**fe), mk_param(**pd))
    ps = dict(SpS=paramTx.SpS, nFilterTaps=paramTx.nFilterTaps, rollOff=paramTx.pulseRollOff, pulseType=paramTx.pulseType)
    pulse = ref_core.pulseShape(mk_param(**ps))
    sigMF = ref_core.firFilter(pulse, sigRx)
    dec = dict(SpSin=paramTx.SpS, SpSout=2)
    sigDec = ref_core.decimate(sigMF, mk_param(**dec))
    ed = dict(L=pch.Ltotal, D=pch.D, Fc=pch.Fc, Rs=paramTx.Rs, Fs=2 * paramTx.Rs)
    sigEDC = ref_edc(sigDec, mk_param(**ed))
    d = 256
    cfg = dict(func="chain", tx=tx, ch=ch, lo=lo, fe=fe, pd=pd, ps=ps, dec=dec, edc=ed, chIndex=chIndex, d=d)
    sz = save("chain_wdm_240k", cfg=json.dumps(cfg), symb_head=symbTx[:64].copy(), symb_proj=projection(symbTx.reshape(len(symbTx), -1)),
              freqGrid=paramTx.wdmFreqGrid, iters=iters, lims=np.array(tr.lims),
              tx_dec=sigTx[::d].copy(), tx_proj=projection(sigTx), ch_dec=sigCh[::d].copy(), ch_proj=projection(sigCh),
              lo_dec=sigLO[::d].copy(), rx_dec=sigRx[::d].copy(), rx_proj=projection(sigRx), mf_dec=sigMF[::d].copy(),
              mf_proj=projection(sigMF), dec_proj=projection(sigDec), out=sigEDC, out_proj=projection(sigEDC))
    print(f"chain_wdm_240k: N {sigTx.shape} steps {len(iters)} iters {int(iters.sum())}  out {sigEDC.dtype}{sigEDC.shape}  {sz/1024:.0f} KiB  {time.time()-t0:.0f} s")


def bfc_vectors():
    """blockwiseFFTConv itself (optic/dsp/core.py:973-1046), the overlap-save convolution under edc / delaySignal, as a callable
    (missing #3): impulse responses and frequency responses, odd and even lengths, a filter longer than the
    signal, a real signal, a 6001-tap filter (more than one 4096-tap segment on the device), NFFT given and None."""
    import optic.dsp.core as ref_core
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(90)
    cx = lambda n: (rng.normal(size=n) + 1j * rng.normal(size=n)) / np.sqrt(2)      # noqa: E731
    f512 = np.fft.fftfreq(512, d=1 / 64e9)
    cases = (
        ("bfc_taps33_complex", cx(3000), cx(33) / 8, None, False),
        ("bfc_taps64_real_nfft256", rng.normal(size=2048), rng.normal(size=64) / 8, 256, False),
        ("bfc_freq512_nfft1024", cx(5000), np.exp(-1j * 2 * np.pi * f512 * 37.3e-12), 1024, True),
        ("bfc_taps6001_nfft8192", cx(20000), cx(6001) / 64, 8192, False),
        ("bfc_freq10000_nfft16384", cx(12000), np.exp(-1j * 0.5 * (2 * np.pi * np.fft.fftfreq(10000)) ** 2 * 9000.0), 16384, True),
        ("bfc_taps255_longer_than_signal", cx(100), cx(255) / 16, None, False),
    )
    for name, x, h, nfft, fd in cases:
        out = ref_core.blockwiseFFTConv(x, h, NFFT=nfft, freqDomainFilter=fd)
        save(name, Ei=x, h=h, out=out, cfg=cfg_json("blockwiseFFTConv", dict(NFFT=nfft, freqDomainFilter=fd)))
        print(f"{name:34s} x {x.dtype}{x.shape} h {h.dtype}{h.shape} NFFT {nfft} freq {fd} out {out.dtype}{out.shape}")


def mixed_dtype_vectors():
    """The GPU twin's mixed-dtype calls (missing #4): optic/models/modelsGPU.py:214-226, 402-404 casts the input
    to `prec` (`Ei_ = cp.asarray(Ei).astype(prec)`), computes everything in `prec`, and returns either the snapshots in `prec`
    (:376-377) or `Ech = Ei.copy(); Ech[:, 0::2] = ...` (:505-507), i.e. the result cast back to the INPUT's dtype.  The twin itself
    cannot be imported here (no cupy); what it computes is the CPU reference on the cast input ("CPU semantics + the cast",
    SURVEY.md 8c), which is what is run: complex64 samples with the default prec (complex128 arithmetic, complex64 out) and
    complex128 samples with prec = complex64 (the reference's complex64 mode, complex128 out), each with saveSpanN = [] and [1]."""
    os.makedirs(OUT, exist_ok=True)
    base = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=2, Lspan=1, hz=0.25,
                nlprMethod=False, amp="ideal")
    for name, in_dt, prec in (("mix_c64in_prec128", np.complex64, np.complex128), ("mix_c128in_prec64", np.complex128, np.complex64)):
        Ei = synth_field(1 << 10, 2, 77, 8.4, in_dt)
        for tag, save in (("final", []), ("span1", [1])):
            kw = dict(base, saveSpanN=save, prec=prec)
            p = mk_param(**kw)
            with Tracer(ref_ch) as tr:
                out = ref_ch.manakovSSF(Ei.astype(prec), p)
            out = out.astype(in_dt) if not save else out.astype(prec)
            iters = np.array(split_iters(tr.lims, p.tol, p.maxIter))
            save_name = f"{name}_{tag}"
            globals()["save"](save_name, Ei=Ei, out=out, iters=iters, cfg=cfg_json("manakovSSF", kw))
            print(f"{save_name:30s} in {Ei.dtype} prec {np.dtype(prec).name} out {out.dtype}{out.shape} iters {int(iters.sum())}")


def rx_chain20_vector():
    """The receiver side of the notebook chain at 2^20 samples for bench.py's "also" leg: the REFERENCE's
    basicLaserModel (LO) -> pdmCoherentReceiver -> firFilter (1024-tap RRC matched filter) -> decimate (16 -> 2) -> edc (800 km) on
    a seeded 2 x 2^20 field (the channel benchmarks' recipe: band-limited complex Gaussian, 0 dBm).  Stored: the decimated final
    output, per-column power and a seeded projection of all of it (bench.py regenerates the input from the recipe)."""
    import time
    import optic.dsp.core as ref_core
    import optic.models.devices as ref_dev
    from optic.dsp.equalization import edc as ref_edc
    os.makedirs(OUT, exist_ok=True)
    t0 = time.time()
    N, Fs, Rs = 1 << 20, 512e9, 32e9
    synth = (N, 2, 9, 0.0)
    E = synth_field(*synth)
    lo = dict(P=10, lw=100e3, RIN_var=0, Ns=N, Fs=Fs, seed=789, freqShift=-128e6)
    sigLO = ref_dev.basicLaserModel(mk_param(**lo))
    fe = dict(Fs=Fs, polRotation=np.pi / 3, pdl=0, polDelay=3 / Rs, phaseImbX=0.0, phaseImbY=0.0, ampImbX=0, ampImbY=0)
    pd = dict(B=Rs, Fs=Fs, ideal=True, seed=1011)
    sigRx = ref_dev.pdmCoherentReceiver(E, sigLO, mk_param(**fe), mk_param(**pd))
    ps = dict(SpS=16, nFilterTaps=1024, rollOff=0.01, pulseType="rrc")
    sigMF = ref_core.firFilter(ref_core.pulseShape(mk_param(**ps)), sigRx)
    dec = dict(SpSin=16, SpSout=2)
    sigDec = ref_core.decimate(sigMF, mk_param(**dec))
    ed = dict(L=800, D=16, Fc=193.1e12, Rs=Rs, Fs=2 * Rs)
    out = ref_edc(sigDec, mk_param(**ed))
    d = 64
    cfg = dict(func="rx_chain", synth=list(synth), lo=lo, fe=fe, pd=pd, ps=ps, dec=dec, edc=ed, d=d)
    sz = save("wl_rx_chain_n20", cfg=json.dumps(cfg), out_dec=out[::d].copy(), out_power=np.sum(np.abs(out) ** 2, axis=0), out_proj=projection(out),
              rx_proj=projection(sigRx))
    print(f"wl_rx_chain_n20: out {out.dtype}{out.shape}  {sz/1024:.0f} KiB  {time.time()-t0:.0f} s")


def unit_checksum(out_cols, seed=4242):
    """bench.py's per-unit checksum of an (N, ncols) reference output: sum |E|^2 and <q, E> over the (ncols, N) SoA block
    with the seeded unit-variance complex vector q (bench.py: unit_checksum)."""
    o = np.ascontiguousarray(out_cols.T).astype(np.complex128)
    rng = np.random.default_rng(seed)
    q = (rng.normal(size=o.shape) + 1j * rng.normal(size=o.shape)) / np.sqrt(2)
    return float(np.sum(np.abs(o) ** 2)), complex(np.vdot(q, o))


def units45_vector(steps=8, log2n=20):
    """Every unit of BASELINE configs 4 and 5 through the reference for `steps` passes at the workload's own size
    (bench.py: config 4 = seeds 100..115, launch powers 0.4..7.9 dBm; config 5 = seeds 200..207 at 8.4 dBm, forward leg then
    manakovDBP over the same span with hz 0.08): the per-unit (sum |E|^2, <q, E>) that bench.py reports as unit_checksums
    (VERDICT round 3, item 1b).  About 4 minutes at 2^20."""
    import time
    os.makedirs(OUT, exist_ok=True)
    N = 1 << log2n
    L = (steps - 0.5) * 0.08
    kw = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
              Ltotal=L, Lspan=L, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    import optic.dsp.equalization as ref_eq
    t0 = time.time()
    c4, it4 = [], []
    for u in range(16):
        E = synth_field(N, 2, 100 + u, 8.4 - 8.0 + 0.5 * u)
        with Tracer(ref_ch) as tr:
            out = ref_ch.manakovSSF(E, mk_param(**kw))
        pw, pr = unit_checksum(out)
        c4.append([pw, pr.real, pr.imag])
        it4.append(len(tr.lims))
        print(f"config 4 unit {u:2d}: power {pw:.9e} iterations {len(tr.lims)}  {time.time()-t0:.0f} s", flush=True)
    c5, c5f, it5 = [], [], []
    for u in range(8):
        E = synth_field(N, 2, 200 + u, 8.4)
        with Tracer(ref_ch) as tr:
            fwd = ref_ch.manakovSSF(E, mk_param(**kw))
        with Tracer(ref_eq) as trb:
            back = ref_dbp(fwd, mk_param(**kw))
        pw, pr = unit_checksum(back)
        pwf, prf = unit_checksum(fwd)
        c5.append([pw, pr.real, pr.imag])
        c5f.append([pwf, prf.real, prf.imag])
        it5.append([len(tr.lims), len(trb.lims)])
        print(f"config 5 unit {u:2d}: power {pw:.9e} iterations {it5[-1]}  {time.time()-t0:.0f} s", flush=True)
    cfg = json.loads(cfg_json("manakovSSF+manakovDBP", kw))
    cfg.update(steps=steps, log2n=log2n, checksum_seed=4242)
    sz = save("wl_units45_n%d" % log2n, cfg=json.dumps(cfg), c4=np.array(c4), c4_iterations=np.array(it4), c5=np.array(c5),
              c5_forward=np.array(c5f), c5_iterations=np.array(it5))
    print(f"wl_units45_n{log2n}: {sz} B  {time.time()-t0:.0f} s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "notebook":   # the reference benchmark's own lengths (2e5, 2e6 samples), adaptive step
        notebook_vectors()
    elif len(sys.argv) > 1 and sys.argv[1] == "long20":  # only the full-size config-2 vector
        long20_vector()
    elif len(sys.argv) > 1 and sys.argv[1] == "cfg3":    # config 3's own field, a few steps, complex128 and complex64
        cfg3_vector()
    elif len(sys.argv) > 1 and sys.argv[1] == "mixed":   # the GPU twin's mixed-dtype calls
        mixed_dtype_vectors()
    elif len(sys.argv) > 1 and sys.argv[1] == "rx_chain20":   # the receiver chain at 2^20 (bench.py's "also" leg)
        rx_chain20_vector()
    elif len(sys.argv) > 1 and sys.argv[1] == "bfc":     # blockwiseFFTConv as a callable
        bfc_vectors()
    elif len(sys.argv) > 1 and sys.argv[1] == "chain":   # the notebook chain end to end (transmitter -> channel -> receiver -> edc)
        chain_vector()
    elif len(sys.argv) > 1 and sys.argv[1] == "long_c3":  # config 3's own field over one full span (about an hour per precision)
        long_c3_vector(sys.argv[2] if len(sys.argv) > 2 else "both")
    elif len(sys.argv) > 1 and sys.argv[1] == "units45":  # every unit of configs 4 / 5 at workload size
        units45_vector(log2n=int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    elif len(sys.argv) > 1 and sys.argv[1] == "long":    # only the long-run vectors
        long_vectors()
    elif len(sys.argv) > 1 and sys.argv[1] == "tx":      # only the transmitter vectors
        os.makedirs(OUT, exist_ok=True)
        tx_vectors()
    elif len(sys.argv) > 1 and sys.argv[1] == "rx_edges":   # only the receiver-side argument edges (round 4)
        rx_edge_vectors()
    elif len(sys.argv) > 1 and sys.argv[1] == "rx":    # only the receiver-side vectors
        os.makedirs(OUT, exist_ok=True)
        rx_vectors()
    else:
        main()
