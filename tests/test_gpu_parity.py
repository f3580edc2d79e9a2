"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the
C ABI via the host wrapper, against
  * the committed golden vectors produced by the reference (tests/golden),
  * the CPU oracle on seeded inputs at sizes it finishes in seconds,
  * size-independent physics properties at BASELINE.json's full sizes.

Tolerances: complex128 rel-L2 <= 1e-10 and identical per-step
iteration counts; complex64 rel-L2 <= 5e-4 (iteration totals reported, a flip at
a threshold crossing is allowed and bounded by tol)."""
import numpy as np
import pytest

import opticommpy_amd as oa
from helpers import golden_names, load_golden, make_param, parity_gate, rel_l2, synth_field
from opticommpy_amd import models
from oracle import ssf_oracle as orc

pytestmark = pytest.mark.gpu

TOL_C128 = 1e-10
TOL_C64 = 5e-4
ENGINES = ["rocfft", "fused"]
FUNCS = {"ssfm": oa.ssfm, "manakovSSF": oa.manakovSSF, "manakovDBP": oa.manakovDBP}


@pytest.mark.parametrize("name", golden_names("edc_"))
def test_edc_golden_vectors(name):
    d, cfg = load_golden(name)
    out = oa.edc(d["Ei"], make_param(oa.parameters, cfg))
    assert out.dtype == d["out"].dtype and out.shape == d["out"].shape
    assert rel_l2(out, d["out"]) <= 1e-12


def test_edc_vs_oracle_large():
    E = synth_field(1 << 20, 2, 78, 0.0)
    p = oa.parameters()
    p.Fs, p.L, p.D, p.Fc, p.Rs = 64e9, 800.0, 17, 193.1e12, 32e9
    ref = orc.edc(E[: 1 << 16], make_param(orc.parameters, dict(Fs=64e9, L=800.0, D=17, Fc=193.1e12, Rs=32e9)))
    out = oa.edc(E, p)
    # the first 2^16 - (filter length) samples do not depend on what follows
    assert rel_l2(out[: 60000], ref[: 60000]) <= 1e-12
    assert out.shape == E.shape
ORC = {"ssfm": orc.ssfm, "manakovSSF": orc.manakovSSF, "manakovDBP": orc.manakovDBP}


def _is_pow2(n):
    return n & (n - 1) == 0


@pytest.fixture(autouse=True)
def _reset_engine():
    yield
    oa.set_engine("auto")


def _select(engine, N):
    if engine == "fused" and not models.engine_supported("fused", N):
        pytest.skip("fused engine does not support this length (yet)")
    oa.set_engine(engine)


def _run_hip(cfg, Ei, **kw):
    p = make_param(oa.parameters, cfg)
    out = FUNCS[cfg["func"]](Ei, p, _trace=True, **kw)
    return out, p, dict(models.last_run)


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("name", [n for n in golden_names() if not n.startswith(("edc_", "rx_", "tx_", "long_", "wl_", "chain_", "bfc_", "mix_"))])
def test_golden_vectors(name, engine):
    d, cfg = load_golden(name)
    _select(engine, d["Ei"].shape[0])
    kw = {}
    if cfg.get("amp") == "edfa" and cfg["func"] != "manakovDBP":
        kw["_cpu_seed_policy"] = True          # goldens come from the CPU reference (one seed for all spans)
    out, p, run = _run_hip(cfg, d["Ei"], **kw)
    ref = d["out"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert run["engine"] == engine
    c64 = cfg.get("prec") == "complex64"
    gate = parity_gate(cfg["func"], d["Ei"], cfg, TOL_C64 if c64 else TOL_C128)
    if gate is None:        # chaotic reference set-up: the reference asserts properties only, so do we
        assert orc.signalPower(out) == pytest.approx(orc.signalPower(d["Ei"]), rel=1e-9)
    else:
        assert rel_l2(out, ref) <= gate
    if "iters" in d:
        assert run["steps"] == len(d["iters"])
        if not c64:
            assert list(run["iters"]) == list(d["iters"])
            flat = np.concatenate(run["lims"])
            np.testing.assert_allclose(flat, d["lims"], rtol=1e-6)
        else:
            assert abs(int(run["iterations"]) - int(d["iters"].sum())) <= 2
        assert run["transforms"] == d["Ei"].shape[1] * (2 * run["steps"] + 2 * run["iterations"])


@pytest.mark.parametrize("engine", ENGINES)
def test_reference_property_gamma0_equals_linear_channel(engine):
    d, cfg = load_golden("ssfm_ref_gamma0")
    _select(engine, 4096)
    out, _, _ = _run_hip(cfg, d["Ei"])
    lp = oa.parameters()
    lp.L, lp.alpha, lp.D, lp.Fc, lp.Fs = 80, 0.2, 16, 193.1e12, 64e9
    lin = oa.linearFiberChannel(d["Ei"], lp)
    np.testing.assert_allclose(out, lin, atol=1e-12)
    np.testing.assert_allclose(lin, d["extra_linear"], atol=1e-12)


@pytest.mark.parametrize("engine", ENGINES)
def test_reference_property_spm_and_power(engine):
    d, cfg = load_golden("ssfm_ref_spm")
    _select(engine, 4096)
    out, _, _ = _run_hip(cfg, d["Ei"])
    out0, _, _ = _run_hip(dict(cfg, gamma=0), d["Ei"])
    assert not np.allclose(np.abs(np.fft.fft(out)), np.abs(np.fft.fft(out0)))
    d, cfg = load_golden("ssfm_ref_power")
    out, _, _ = _run_hip(cfg, d["Ei"])
    assert orc.signalPower(out) == pytest.approx(orc.signalPower(d["Ei"]), rel=1e-9)


def _mk_cfg(**kw):
    base = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5,
                prgsBar=False)
    base.update(kw)
    return base


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("N,adaptive,prec", [(1 << 14, False, "complex128"), (1 << 16, False, "complex128"),
                                             (1 << 14, True, "complex128"), (1 << 16, False, "complex64"),
                                             (12000, False, "complex128")])
def test_manakov_vs_oracle_seeded(engine, N, adaptive, prec):
    _select(engine, N)
    E = synth_field(N, 2, 31, 8.4, np.dtype(prec).type)
    cfg = _mk_cfg(Ltotal=4, Lspan=2, hz=0.08, nlprMethod=adaptive, maxNlinPhaseRot=2e-2, amp="ideal",
                  saveSpanN=[], prec=prec)
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
    out, _, run = _run_hip(cfg, E)
    c64 = prec == "complex64"
    assert rel_l2(out, ref) <= (TOL_C64 if c64 else TOL_C128)
    assert run["steps"] == tr["steps"]
    if not c64:
        assert list(run["iters"]) == tr["iters"]
        np.testing.assert_allclose(run["hz"], tr["hz"], rtol=1e-9)


@pytest.mark.parametrize("engine", ENGINES)
def test_manakov_batched_pairs_vs_oracle(engine):
    N = 1 << 13
    _select(engine, N)
    E = synth_field(N, 6, 32, 10.0)
    cfg = _mk_cfg(Ltotal=2, Lspan=1, hz=0.1, nlprMethod=True, amp=None, saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
    out, _, run = _run_hip(cfg, E)
    assert rel_l2(out, ref) <= TOL_C128
    assert list(run["iters"]) == tr["iters"]


@pytest.mark.parametrize("engine", ENGINES)
def test_dbp_and_ssfm_vs_oracle_seeded(engine):
    N = 1 << 15
    _select(engine, N)
    E = synth_field(N, 2, 33, 6.0)
    cfg = _mk_cfg(func="manakovDBP", Ltotal=40, Lspan=20, hz=10, nlprMethod=False, amp="edfa", saveSpanN=[])
    tr = {}
    ref = orc.manakovDBP(E, make_param(orc.parameters, cfg), trace=tr)
    out, _, run = _run_hip(cfg, E)
    assert rel_l2(out, ref) <= TOL_C128
    assert list(run["iters"]) == tr["iters"]
    e1 = E[:, 0].copy()
    cfg = dict(func="ssfm", Ltotal=10, Lspan=5, hz=0.25, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9,
               amp="ideal", prgsBar=False, saveSpanN=[1, 2])
    ref = orc.ssfm(e1, make_param(orc.parameters, cfg))
    out, _, _ = _run_hip(cfg, e1)
    assert out.shape == ref.shape == (N, 2)
    assert rel_l2(out, ref) <= TOL_C128


@pytest.mark.parametrize("engine", ENGINES)
def test_edfa_gain_exact_with_host_noise_and_noise_variance(engine):
    N = 1 << 12
    _select(engine, N)
    E = synth_field(N, 2, 34, 3.0)
    cfg = _mk_cfg(Ltotal=20, Lspan=10, hz=0.5, nlprMethod=False, amp="edfa", NF=5.0, saveSpanN=[])
    rng = np.random.default_rng(1)
    noise = 1e-4 * (rng.normal(size=(2, 2, N)) + 1j * rng.normal(size=(2, 2, N)))
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), noise=noise)
    p = make_param(oa.parameters, cfg)
    out = oa.manakovSSF(E, p, _noise=noise)
    assert rel_l2(out, ref) <= TOL_C128
    # statistical check of the library's own draw (GPU-twin seed policy: seed + span)
    p = make_param(oa.parameters, dict(cfg, Ltotal=10, seed=3))
    o1 = oa.manakovSSF(E, p)
    pz = make_param(oa.parameters, dict(cfg, Ltotal=10))
    oz = oa.manakovSSF(E, pz, _noise=np.zeros((1, 2, N), complex))
    _, p_noise = orc.edfa_noise_power(0.2 * 10, 5.0, 193.1e12, 512e9)
    assert np.mean(np.abs(o1 - oz) ** 2) == pytest.approx(p_noise, rel=0.1)


@pytest.mark.parametrize("engine", ENGINES)
def test_api_contract_on_device(engine):
    N = 2048
    _select(engine, N)
    E = synth_field(N, 2, 35, 0.0)
    E_before = E.copy()
    p = oa.parameters()
    p.Fs, p.Ltotal, p.Lspan, p.amp, p.prgsBar, p.returnParameters = 512e9, 10, 5, "ideal", True, True
    out, p2 = oa.manakovSSF(E, p)
    assert p2 is p and np.array_equal(E, E_before)              # input never modified
    assert out.shape == (N, 2) and out.dtype == np.complex128   # default saveSpanN = [2] -> one snapshot
    assert p.saveSpanN == [2] and p.hz == 0.5 and p.nlprMethod is True
    q = oa.parameters()
    q.Fs, q.Ltotal, q.Lspan, q.amp, q.prgsBar, q.saveSpanN = 512e9, 10, 5, "ideal", False, [1, 2, 7]
    out = oa.manakovSSF(E, q)
    assert out.shape == (N, 6) and np.all(out[:, 4:] == 0) and np.any(out[:, :4] != 0)


# ---------------------------------------------------------------------------
# full-size (BASELINE.json configs) property tests -- no oracle needed
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("engine", ENGINES)
def test_full_size_power_conservation_and_roundtrip(engine):
    N = 1 << 20
    _select(engine, N)
    E = synth_field(N, 2, 2, 8.4)
    cfg = _mk_cfg(Ltotal=1.6, Lspan=1.6, hz=0.08, nlprMethod=False, alpha=0.0, amp=None, saveSpanN=[])
    out, _, run = _run_hip(cfg, E)
    assert orc.signalPower(out) == pytest.approx(orc.signalPower(E), rel=1e-9)      # lossless fiber
    assert run["steps"] == 20 and run["iterations"] >= 40
    assert rel_l2(out, E) > 1e-3                                                     # it did propagate
    back, _, _ = _run_hip(dict(cfg, func="manakovDBP"), out)
    assert rel_l2(back, E) < 1e-6                                                    # DBP o SSF ~ identity


@pytest.mark.parametrize("engine", ENGINES)
def test_full_size_gamma0_equals_linear_channel_and_linearity(engine):
    N = 1 << 20
    _select(engine, N)
    E = synth_field(N, 2, 5, 0.0)
    cfg = _mk_cfg(Ltotal=0.8, Lspan=0.8, hz=0.08, nlprMethod=False, gamma=0.0, amp=None, saveSpanN=[])
    out, _, _ = _run_hip(cfg, E)
    lp = oa.parameters()
    lp.L, lp.alpha, lp.D, lp.Fc, lp.Fs = 0.8, 0.2, 16, 193.1e12, 512e9
    lin = oa.linearFiberChannel(E, lp)
    assert rel_l2(out, lin) <= 1e-11
    out2, _, _ = _run_hip(cfg, (2 - 1j) * E)
    assert rel_l2(out2, (2 - 1j) * out) <= 1e-12                                     # gamma = 0 => linear


@pytest.mark.parametrize("engine", ENGINES)
def test_full_size_config2_short_vs_oracle(engine):
    """BASELINE config 2 inputs (N = 2^20, seed 2, 8.4 dBm, hz 0.08) for 3 steps."""
    N = 1 << 20
    _select(engine, N)
    E = synth_field(N, 2, 2, 8.4)
    cfg = _mk_cfg(Ltotal=0.24, Lspan=0.24, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
    out, _, run = _run_hip(cfg, E)
    assert rel_l2(out, ref) <= TOL_C128
    assert list(run["iters"]) == tr["iters"]


@pytest.mark.parametrize("engine", ENGINES)
def test_full_size_config3_c64(engine):
    """BASELINE config 3 shape (N = 2^22, complex64): power bookkeeping over one short span."""
    N = 1 << 22
    _select(engine, N)
    E = synth_field(N, 2, 3, 8.4, np.complex64)
    cfg = _mk_cfg(Ltotal=0.8, Lspan=0.8, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[], prec="complex64")
    out, _, run = _run_hip(cfg, E)
    assert out.dtype == np.complex64 and run["steps"] == 11     # 0.08 accumulates to just under 0.8: the
    # reference also takes an 11th, rounding-sized step (channels.py:387, 398-400)
    assert orc.signalPower(out) == pytest.approx(orc.signalPower(E), rel=1e-4)       # ideal amp restores the power


@pytest.mark.parametrize("engine", ENGINES)
def test_c64_long_run_does_not_drift_from_c128(engine):
    """2002 fixed steps (2 x 80 km, hz = 0.08): single precision must stay within the tolerance of
    the double-precision run, which is the one pinned against the oracle.  Guards the rounding of
    the butterfly constants and twiddles: a coherent 1e-7 amplitude bias per transform is
    invisible in a 10-step test and is a 0.3 % power loss here."""
    N = 1 << 16
    _select(engine, N)
    E = synth_field(N, 2, 7, 0.0)
    outs = {}
    for prec in ("complex128", "complex64"):
        cfg = _mk_cfg(Ltotal=160, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[], prec=prec)
        outs[prec], _, run = _run_hip(cfg, E.astype(prec))
        assert run["steps"] == 2002
    a, b = outs["complex64"].astype(np.complex128), outs["complex128"]
    # the fused kernels (the product path) hold the SURVEY 8c single-precision gate over twice its 1000 steps.  The
    # rocFFT engine is only the on-GPU cross-check of the transforms (never selected by AUTO inside the fused range): it
    # inherits the library's single-precision transforms, whose rounded twiddles drift like the reference's own complex64
    # path (measured 5.3e-4 here; the reference itself: 7.3e-4 after 10 spans, long_c64drift_*) -- its bound is that
    # yardstick, not the product gate
    tol, ptol = (TOL_C64, 4e-4) if engine == "fused" else (1.5e-3, 2e-3)
    assert rel_l2(a, b) <= tol
    assert np.sum(np.abs(a) ** 2) / np.sum(np.abs(b) ** 2) == pytest.approx(1.0, abs=ptol)


def test_untraced_runs_bound_lim0_and_give_the_traced_result():
    """Product runs record no trace: lim_0 is then only bounded from below (one sample in sixteen), which must
    not change anything; when the bound cannot exclude convergence at iterate 0 the iterate is rebuilt and
    lim_0 measured on all samples (fused engine)."""
    _select("fused", 1024)
    for name in ("mk_fix_p8_ideal_2span", "mk_adp_p13_ideal_2span", "mk_fix_p8_n4096"):
        d, cfg = load_golden(name)
        a = oa.manakovSSF(d["Ei"], make_param(oa.parameters, cfg), _trace=True)
        ra = dict(models.last_run)
        b = oa.manakovSSF(d["Ei"], make_param(oa.parameters, cfg))
        rb = dict(models.last_run)
        assert np.array_equal(a, b) and (ra["steps"], ra["iterations"]) == (rb["steps"], rb["iterations"])
        assert rb["rebuilt_iterates"] == 0
    # tol between the bound and lim_0: every step's iterate 0 is rebuilt once, then the iteration goes on
    d, cfg = load_golden("mk_fix_p0_none")
    oa.manakovSSF(d["Ei"], make_param(oa.parameters, cfg), _trace=True)
    lim0 = min(l[0] for l in models.last_run["lims"])
    cfg2 = dict(cfg, tol=float(lim0) * 0.6)
    tr = {}
    ref = orc.manakovSSF(d["Ei"], make_param(orc.parameters, cfg2), trace=tr)
    a = oa.manakovSSF(d["Ei"], make_param(oa.parameters, cfg2), _trace=True)
    assert list(models.last_run["iters"]) == tr["iters"] and rel_l2(a, ref) <= TOL_C128
    b = oa.manakovSSF(d["Ei"], make_param(oa.parameters, cfg2))
    assert np.array_equal(a, b) and 0 < models.last_run["rebuilt_iterates"] <= models.last_run["steps"]
    # tol around the bound (~ lim_0 / 4): some steps need the exact lim_0 after steps that did not, whose final stage stored
    # the field at one sample in sixteen only -- the whole field is recovered from E_hd first (fused_kernels.h: ST_RECOVER_A)
    oa.manakovSSF(d["Ei"], make_param(oa.parameters, cfg), _trace=True)
    mid = float(np.median([l[0] for l in models.last_run["lims"]]))
    hit = 0
    for f in (0.27, 0.25, 0.22):
        cfg3 = dict(cfg, tol=mid * f)
        a = oa.manakovSSF(d["Ei"], make_param(oa.parameters, cfg3), _trace=True)
        ra = dict(models.last_run)
        b = oa.manakovSSF(d["Ei"], make_param(oa.parameters, cfg3))
        rb = dict(models.last_run)
        assert np.array_equal(a, b) and (ra["steps"], ra["iterations"]) == (rb["steps"], rb["iterations"])
        assert ra["recovered_fields"] == 0
        hit += rb["recovered_fields"]
    assert hit > 0
    oa.set_engine("auto")


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("N,adaptive,prec", [(48000, False, "complex128"), (240000, True, "complex128"),
                                             (240000, False, "complex64"), (960000, False, "complex128"),
                                             (1440000, False, "complex128"), (1440000, True, "complex64")])
def test_notebook_lengths_vs_oracle(engine, N, adaptive, prec):
    """N = SpS x Nsymbols = 2^a * 3^b * 5^c (240 000 = 16 x 15 000 is the reference notebooks' default): the fused
    engine takes these with power-of-two column transforms and mixed-radix row transforms (mixed_fft.h)."""
    _select(engine, N)
    assert models.engine_supported("fused", N)
    E = synth_field(N, 2, 9, 8.4, np.dtype(prec).type)
    cfg = _mk_cfg(Ltotal=0.8, Lspan=0.4, hz=0.08, nlprMethod=adaptive, amp="ideal", saveSpanN=[], prec=prec)
    tr = {}
    ref = orc.manakovSSF(E.astype(np.complex128), make_param(orc.parameters, dict(cfg, prec="complex128")), trace=tr)
    out, _, run = _run_hip(cfg, E)
    assert run["engine"] == engine
    if prec == "complex128":
        assert rel_l2(out, ref) <= TOL_C128 and list(run["iters"]) == tr["iters"]
        out2 = oa.manakovSSF(E, make_param(oa.parameters, cfg))                    # untraced (lim_0 bound)
        assert np.array_equal(out, out2) if engine == "fused" else rel_l2(out2, ref) <= TOL_C128
    else:
        assert rel_l2(out.astype(np.complex128), ref) <= TOL_C64


def test_every_small_mixed_row_length_agrees_with_the_rocfft_engine():
    """All odd 5-smooth row lengths 64..640 (N = 128 x row length: 2^7 columns, so each one is the engine's row length) and
    a few even multiples (2^8 columns, row length m / 2): four Manakov steps on the fused engine against the rocFFT engine
    (every radix, plan shape and ragged last tile)."""
    rows = sorted({2 ** a * 3 ** b * 5 ** c for a in range(8) for b in range(6) for c in range(5)
                   if 64 <= 2 ** a * 3 ** b * 5 ** c <= 640 and (2 ** a * 3 ** b * 5 ** c) % 2 == 1})
    rows += [150, 250, 270, 486, 500, 540, 600, 1250, 1620]
    cfg = _mk_cfg(Ltotal=0.4, Lspan=0.4, hz=0.1, nlprMethod=False, amp="ideal", saveSpanN=[])
    for m in rows:
        N = 128 * m
        if not models.engine_supported("fused", N):
            continue
        E = synth_field(N, 2, m, 8.4)
        outs = {}
        for eng in ("fused", "rocfft"):
            oa.set_engine(eng)
            outs[eng] = oa.manakovSSF(E, make_param(oa.parameters, cfg), _trace=True)
            outs[eng + "_it"] = list(models.last_run["iters"])
            assert models.last_run["engine"] == eng
        assert rel_l2(outs["fused"], outs["rocfft"]) <= 1e-11 and outs["fused_it"] == outs["rocfft_it"], m
    oa.set_engine("auto")


def test_notebook_length_other_entry_points_on_the_fused_engine():
    N = 48000
    _select("fused", N)
    E = synth_field(N, 2, 5, 6.0)
    s = E[:, 0].copy()
    cfg = dict(func="ssfm", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, prgsBar=False, Ltotal=2.0, Lspan=1.0, hz=0.25,
               amp="ideal", saveSpanN=[])
    assert rel_l2(oa.ssfm(s, make_param(oa.parameters, cfg)), orc.ssfm(s, make_param(orc.parameters, cfg))) <= TOL_C128
    assert models.last_run["engine"] == "fused"
    mk = _mk_cfg(func="manakovDBP", Ltotal=2.0, Lspan=1.0, hz=0.5, nlprMethod=False, amp="ideal", saveSpanN=[])
    assert rel_l2(oa.manakovDBP(E, make_param(oa.parameters, mk)), orc.manakovDBP(E, make_param(orc.parameters, mk))) <= TOL_C128
    lp = make_param(oa.parameters, dict(Fs=512e9, L=3.0, alpha=0.2, D=16, Fc=193.1e12))
    lo = make_param(orc.parameters, dict(Fs=512e9, L=3.0, alpha=0.2, D=16, Fc=193.1e12))
    assert rel_l2(oa.linearFiberChannel(E, lp), orc.linearFiberChannel(E, lo)) <= 1e-12
    oa.set_engine("auto")


def test_single_process_multi_device_entry_point():
    """ssf_mgpu_run (one host thread per device): 3 independent units on device 0 must equal
    three separate reference calls."""
    import ctypes as C
    from opticommpy_amd import _lib, mgpu
    N = 1 << 12
    fields = np.stack([synth_field(N, 2, 60 + u, 6.0 + u).T for u in range(3)])       # (U, rows, N)
    cfg = _mk_cfg(Ltotal=2, Lspan=1, hz=0.1, nlprMethod=False, amp="ideal", saveSpanN=[])
    cp = models._fill_params(_lib.MODEL_MANAKOV, +1, make_param(oa.parameters, dict(cfg, NF=4.5)), 512e9, 2,
                             np.zeros(0, np.int32))
    outs, stats = mgpu.run_threads(fields, cp, devices=[0])
    for u in range(3):
        tr = {}
        ref = orc.manakovSSF(fields[u].T.copy(), make_param(orc.parameters, cfg), trace=tr)
        assert rel_l2(outs[u].T, ref) <= TOL_C128
        assert stats[u]["steps"] == tr["steps"] and stats[u]["iterations"] == tr["iterations"]


def test_run_sharded_drives_two_lanes_per_gpu(monkeypatch):
    """mgpu.run_sharded without a process group: the local units go through two host threads (own plan and stream
    each); outputs are those of one-at-a-time calls, bit for bit, for the default and for a chained compute."""
    from opticommpy_amd import mgpu
    N = 1 << 13
    fields = [synth_field(N, 2, 80 + u, 4.0 + u) for u in range(5)]
    cfg = _mk_cfg(Ltotal=2, Lspan=1, hz=0.1, nlprMethod=True, amp="ideal", saveSpanN=[])
    seq = [oa.manakovSSF(f, make_param(oa.parameters, cfg)) for f in fields]
    outs = mgpu.run_sharded(fields, make_param(oa.parameters, cfg))
    assert all(np.array_equal(a, b) for a, b in zip(outs, seq))

    def chain(E, p):                                   # forward channel, then back-propagation (config 5's shape)
        return oa.manakovDBP(oa.manakovSSF(E, p), p)
    seq2 = [chain(f, make_param(oa.parameters, cfg)) for f in fields]
    outs2 = mgpu.run_sharded(fields, make_param(oa.parameters, cfg), compute=chain)
    assert all(np.array_equal(a, b) for a, b in zip(outs2, seq2))
    monkeypatch.setenv("SSF_MGPU_LANES", "1")
    outs1 = mgpu.run_sharded(fields, make_param(oa.parameters, cfg))
    assert all(np.array_equal(a, b) for a, b in zip(outs1, seq))


def test_measured_copy_ceiling_probe():
    """ssf_device_copy_bandwidth: plausible (between 1 and 12 TB/s for 32 MiB in + out), bad sizes refused."""
    import ctypes as C
    from opticommpy_amd import _lib
    lib = _lib.load()
    g = C.c_double(0.0)
    assert lib.ssf_device_copy_bandwidth(0, 32 << 20, 20, C.byref(g)) == 0
    assert 1000.0 < g.value < 12000.0
    assert lib.ssf_device_copy_bandwidth(0, 1000, 20, C.byref(g)) == -1
    assert lib.ssf_device_copy_bandwidth(99, 32 << 20, 20, C.byref(g)) == -1


def test_kernel_profiling_api():
    import ctypes as C
    from opticommpy_amd import _lib
    N = 1 << 14
    oa.set_engine("fused")
    E = synth_field(N, 2, 70, 8.0)
    p = make_param(oa.parameters, _mk_cfg(Ltotal=1, Lspan=1, hz=0.1, nlprMethod=False, amp=None, saveSpanN=[]))
    oa.manakovSSF(E, p)
    pl = models._get_plan(N, 2, _lib.SSF_C128)
    assert pl.lib.ssf_set_profiling(pl.h, 1) == 0
    out = oa.manakovSSF(E, p)
    kt = _lib.KernelTimes()
    assert pl.lib.ssf_get_kernel_times(pl.h, C.byref(kt)) == 0
    pl.lib.ssf_set_profiling(pl.h, 0)
    steps, iters = models.last_run["steps"], models.last_run["iterations"]
    assert kt.row_n >= steps + iters and kt.col_n >= steps + iters and kt.row_ms > 0 and kt.col_ms > 0


@pytest.mark.parametrize("engine", ENGINES)
def test_reference_defaults_end_to_end(engine):
    """A notebook-style call with nothing but Fs set: 5 x 80 km, adaptive step, amp='edfa' (ASE
    generated on the device), default saveSpanN -> one (N, 2) snapshot; seeded runs repeat."""
    N = 1 << 13
    _select(engine, N)
    E = synth_field(N, 2, 80, 0.0)
    outs = []
    for seed in (11, 11, 12):
        p = oa.parameters()
        p.Fs, p.seed, p.prgsBar = 512e9, seed, False
        out = oa.manakovSSF(E, p)
        assert out.shape == (N, 2) and out.dtype == np.complex128 and np.all(np.isfinite(out.view(float)))
        assert p.amp == "edfa" and p.Ltotal == 400 and p.saveSpanN == [5] and p.nlprMethod is True
        outs.append(out)
    assert np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], outs[2])
    # span loss is compensated by the EDFA gain: output power = input power + 5 spans of ASE
    _, p_noise = orc.edfa_noise_power(0.2 * 80, 4.5, 193.1e12, 512e9)
    assert orc.signalPower(outs[0]) == pytest.approx(orc.signalPower(E) + 2 * 5 * p_noise, rel=0.05)


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("case", ["no_span", "hz_longer_than_span", "all_zero_field", "min_length", "three_pairs",
                                  "odd_tiny_length"])
def test_edge_cases_vs_oracle(engine, case):
    import logging
    logging.disable(logging.WARNING)
    try:
        N, ncols, kw = 1024, 2, {}
        if case == "no_span":                      # Ltotal < Lspan: zero spans, nothing propagates
            kw = dict(Ltotal=5, Lspan=10, saveSpanN=[])
        elif case == "hz_longer_than_span":        # one shortened step per span (channels.py:398-400)
            kw = dict(Ltotal=4, Lspan=2, hz=5.0, saveSpanN=[1, 2])
        elif case == "all_zero_field":             # lim = 0/0 = nan never passes: maxIter iterations, warnings
            kw = dict(Ltotal=1, Lspan=1, hz=0.5, maxIter=3, saveSpanN=[])
        elif case == "min_length":
            N = 256
            kw = dict(Ltotal=2, Lspan=1, hz=0.25, saveSpanN=[])
        elif case == "three_pairs":
            ncols = 6
            kw = dict(Ltotal=2, Lspan=1, hz=0.25, saveSpanN=[])
        else:
            N = 97
            kw = dict(Ltotal=2, Lspan=1, hz=0.25, saveSpanN=[])
        _select(engine, N)
        E = synth_field(N, ncols, 90, 6.0) if case != "all_zero_field" else np.zeros((N, 2), complex)
        cfg = _mk_cfg(nlprMethod=False, amp="ideal", **kw)
        tr = {}
        ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
        out, _, run = _run_hip(cfg, E)
        assert out.shape == ref.shape and out.dtype == ref.dtype
        if case == "all_zero_field":
            assert np.all(out == 0) and run["nonconverged_steps"] == tr["nonconverged"] == run["steps"]
        elif case == "no_span":
            assert np.array_equal(out, E) and run["steps"] == 0
        else:
            assert rel_l2(out, ref) <= TOL_C128
        assert run["steps"] == tr.get("steps", 0) and list(run["iters"]) == tr.get("iters", [])
    finally:
        logging.disable(logging.NOTSET)


# ---- round 5: the GPU twin's mixed-dtype calls (optic/models/modelsGPU.py:214-226, 402-404, 505-507) -----------------------------------
@pytest.mark.parametrize("name", golden_names("mix_"))
def test_mixed_input_dtype_and_prec_follow_the_gpu_twin(name):
    """complex64 samples with the default prec: the twin casts the input up, computes in complex128 and hands the result back in
    the input's dtype (`Ech = Ei.copy()`), snapshots in prec; complex128 samples with prec = complex64: cast down, the
    reference's complex64 mode, result back in complex128.  Vectors: the reference on the cast input (tools/gen_golden.py mixed)."""
    d, cfg = load_golden(name)
    out = oa.manakovSSF(d["Ei"].copy(), make_param(oa.parameters, cfg), _trace=True)
    assert out.dtype == d["out"].dtype and out.shape == d["out"].shape
    single = np.dtype(cfg["prec"]) == np.dtype(np.complex64)
    tol = 5e-4 if single else (1e-10 if out.dtype == np.complex128 else 2e-7)     # (complex64 storage of a complex128 result: 6e-8)
    assert rel_l2(out, d["out"]) <= tol
    if not single:
        assert list(oa.last_run["iters"]) == list(d["iters"])
    else:
        assert abs(int(np.sum(oa.last_run["iters"])) - int(d["iters"].sum())) <= 2


# ---- round 5: BASELINE config 1 at its own size (it was only checked in bench.py's "also" leg) -------------------------
CFG1 = dict(func="ssfm", Ltotal=50, Lspan=50, hz=0.5, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, amp=None, prgsBar=False)


@pytest.mark.parametrize("engine", ENGINES)
def test_config1_at_its_own_size(engine):
    """Single-pol ssfm, 2^16 complex samples, 50 km / 100 steps (BASELINE config 1, seed 1, 0 dBm) against the oracle, both engines."""
    N = 1 << 16
    _select(engine, N)
    E = synth_field(N, 1, 1, 0.0)[:, 0].copy()
    ref = orc.ssfm(E, make_param(orc.parameters, CFG1))
    out, _, run = _run_hip(CFG1, E)
    assert out.shape == (N,) and run["steps"] == 100
    assert rel_l2(out, ref) <= TOL_C128


def test_config1_as_sixteen_fields_of_one_plan():
    """... and as bench.py's config-1 leg runs it: 16 independent fields (seeds 1 ... 16) as the rows of ONE scalar-NLSE plan, every
    launch carrying all of them (C ABI: ssf_plan_create(N, nrows = 16) / ssf_upload / ssf_execute / ssf_download): every field against
    the oracle's run of that field."""
    import ctypes as C
    from opticommpy_amd import _lib
    lib = _lib.load()
    N, F = 1 << 16, 16
    fields = np.ascontiguousarray(np.concatenate([synth_field(N, 1, 1 + f, 0.0).T for f in range(F)], axis=0))
    h = C.c_void_p()
    _lib.raise_for(lib, None, lib.ssf_plan_create(0, N, F, _lib.SSF_C128, 0, C.byref(h)))
    try:
        cp = _lib.Params()
        cp.model, cp.direction = _lib.MODEL_NLSE, 1
        cp.Fs, cp.Fc, cp.alpha, cp.D, cp.gamma = 512e9, 193.1e12, 0.2, 16.0, 1.3
        cp.Lspan, cp.Nspans, cp.hz, cp.maxIter, cp.tol = 50.0, 1, 0.5, 10, 1e-5
        cp.nlprMethod, cp.maxNlinPhaseRot, cp.NF, cp.amp = 0, 2e-2, 4.5, _lib.AMP_NONE
        cp.n_save, cp.save_spans = 0, None
        st = _lib.Stats()
        _lib.raise_for(lib, h, lib.ssf_upload(h, fields.ctypes.data_as(C.c_void_p)))
        _lib.raise_for(lib, h, lib.ssf_execute(h, C.byref(cp), 1, 1, None, C.byref(st), None))
        got = np.empty_like(fields)
        _lib.raise_for(lib, h, lib.ssf_download(h, got.ctypes.data_as(C.c_void_p)))
    finally:
        lib.ssf_plan_destroy(h)
    assert int(st.steps) == 100
    for f in range(F):
        ref = orc.ssfm(fields[f].copy(), make_param(orc.parameters, CFG1))
        assert rel_l2(got[f], ref) <= TOL_C128, f
