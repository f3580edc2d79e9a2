"""CPU tests of the fused-engine kernel SOURCE through the emulator (tests/emu): the same
fused_kernels.h / fused_engine.h that hipcc compiles for gfx950, stepped thread by thread on
the host.  This pins the FFT index maps, twiddles, LDS exchange pattern, the device-side
step state machine and the host enqueue logic against the golden vectors and the oracle
before any GPU time is spent.  (The emulator is test infrastructure, never a product path.)"""
import numpy as np
import pytest

import emu_binding as eb
from helpers import golden_names, load_golden, make_param, parity_gate, rel_l2, synth_field
from oracle import ssf_oracle as orc

TOL_C128, TOL_C64 = 1e-10, 5e-4


def _pow2(n):
    return n & (n - 1) == 0


def _assemble(func, cfg, d, out, info):
    ref = d["out"]
    save = cfg.get("saveSpanN", None)
    if func != "ssfm" and (save is None or len(save) > 0):
        got = np.zeros(ref.shape, dtype=ref.dtype)
        for i, s in enumerate(info["snaps"]):
            got[:, 2 * i:2 * i + 2] = s.T
        return got
    return out.T.reshape(ref.shape)


@pytest.mark.parametrize("N", [256, 512, 1024, 2048, 4096, 8192, 1 << 14, 1 << 15, 1 << 16])
def test_linear_channel_all_pass_plans(N):
    """One FFT.H.IFFT: exercises column DIF/DIT, row DIF/DIT, the inter-pass twiddle and the
    bin-index -> operator map for every radix mix (16 | 16,2 | 16,4 | 16,8 | 16,16 | 16,2,16 ...)."""
    E = synth_field(N, 3, N, 0.0)
    p = orc.parameters()
    p.Fs, p.L, p.alpha, p.D, p.Fc = 512e9, 7.0, 0.2, 16, 193.1e12
    ref = orc.linearFiberChannel(E, p)
    out = eb.linear_channel(E, 512e9, 193.1e12, 0.2, 16, 7.0)
    assert rel_l2(out, ref) < 1e-13


def test_linear_channel_c64_and_large():
    E = synth_field(1 << 13, 2, 3, 0.0, np.complex64)
    p = orc.parameters()
    p.Fs, p.L, p.alpha, p.D, p.Fc = 512e9, 2.0, 0.2, 16, 193.1e12
    ref = orc.linearFiberChannel(E.astype(np.complex128), p)
    assert rel_l2(eb.linear_channel(E, 512e9, 193.1e12, 0.2, 16, 2.0, np.complex64), ref) < 2e-6
    E = synth_field(1 << 18, 2, 4, 0.0)          # N1 = 256 columns, N2 = 1024 rows (16,4,16 passes)
    ref = orc.linearFiberChannel(E, p)
    assert rel_l2(eb.linear_channel(E, 512e9, 193.1e12, 0.2, 16, 2.0), ref) < 1e-13


@pytest.mark.parametrize("name", [n for n in golden_names() if not n.startswith(("rx_", "tx_", "long_", "wl_", "chain_", "bfc_", "mix_"))])
def test_golden_vectors_on_emulated_kernels(name):
    d, cfg = load_golden(name)
    N = d["Ei"].shape[0]
    func = cfg["func"]
    if func != "edc" and not _pow2(N):
        pytest.skip("non power-of-two lengths run on the rocFFT engine")
    if func == "edc":
        pytest.skip("edc vectors are covered by test_edc_overlap_save_on_emulated_kernel")
    cfg = dict(cfg)
    noise = None
    if cfg.get("amp") == "edfa" and func != "manakovDBP":
        # reproduce the CPU reference's draws (same seed every span; x and y share the noise)
        from opticommpy_amd import models
        Nspans = int(cfg["Ltotal"] // cfg["Lspan"])
        G = cfg["alpha"] * cfg["Lspan"]
        _, pn = orc.edfa_noise_power(G, cfg["NF"], cfg["Fc"], cfg["Fs"])
        nr = 1 if func == "ssfm" else d["Ei"].shape[1]
        noise = np.stack([models._span_noise(nr, N, pn, cfg["seed"], func != "ssfm", np.complex128)
                          for _ in range(Nspans)])
    if func == "ssfm" and "saveSpanN" not in cfg:
        cfg["saveSpanN"] = []
    out, info = eb.run(func, d["Ei"], cfg, noise=noise)
    got = _assemble(func, cfg, d, out, info)
    c64 = cfg.get("prec") == "complex64"
    gate = parity_gate(func, d["Ei"], cfg, TOL_C64 if c64 else TOL_C128)
    if gate is None:
        assert orc.signalPower(got) == pytest.approx(orc.signalPower(d["Ei"]), rel=1e-9)
    else:
        assert rel_l2(got, d["out"]) <= gate
    if "iters" in d:
        assert list(info["iters"]) == list(d["iters"])
        assert info["steps"] == len(d["iters"]) and info["iterations"] == int(d["iters"].sum())
        if not c64:
            np.testing.assert_allclose(np.concatenate(info["lims"]), d["lims"], rtol=1e-6)
        assert info["nonconverged_steps"] == sum(
            1 for it, l in zip(d["iters"], info["lims"]) if it == cfg.get("maxIter", 10) and l[-1] >= cfg.get("tol", 1e-5))


@pytest.mark.parametrize("N,adaptive", [(1 << 14, False), (1 << 14, True), (1 << 16, False)])
def test_manakov_vs_oracle_mid_size(N, adaptive):
    E = synth_field(N, 2, 31, 8.4)
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5,
               prgsBar=False, Ltotal=0.64, Lspan=0.32, hz=0.08, nlprMethod=adaptive, maxNlinPhaseRot=2e-2,
               amp="ideal", saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
    out, info = eb.run("manakovSSF", E, cfg)
    assert rel_l2(out.T, ref) <= TOL_C128
    assert list(info["iters"]) == tr["iters"]
    np.testing.assert_allclose(info["hz"], tr["hz"], rtol=1e-9)


def test_headline_geometry_on_emulated_kernels():
    """N = 2^20 (N1 = 256 columns x N2 = 4096 rows, three radix-16 passes, LDS twiddle tables incl.
    the two-level W_4096 form): one linear channel and two Manakov steps of BASELINE config 2."""
    E = synth_field(1 << 20, 2, 2, 8.4)
    p = orc.parameters()
    p.Fs, p.L, p.alpha, p.D, p.Fc = 512e9, 2.0, 0.2, 16, 193.1e12
    assert rel_l2(eb.linear_channel(E, 512e9, 193.1e12, 0.2, 16, 2.0), orc.linearFiberChannel(E, p)) < 1e-14
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5,
               prgsBar=False, Ltotal=0.16, Lspan=0.16, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
    out, info = eb.run("manakovSSF", E, cfg)
    assert rel_l2(out.T, ref) <= TOL_C128 and list(info["iters"]) == tr["iters"]


def test_c64_kernels_have_no_coherent_amplitude_bias():
    """600 single-precision steps on the emulated kernels against the complex128 oracle.  With
    float-rounded butterfly constants / float twiddle power trees every transform lost ~1e-7 of
    amplitude in the same direction (0.9974 power after 2002 steps); unbiased constants and
    twiddles keep the power within 1e-4 here."""
    N = 4096
    E = synth_field(N, 2, 3, 0.0)
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=256e9, maxIter=10, tol=1e-5,
               prgsBar=False, Ltotal=48, Lspan=48, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg))
    out, info = eb.run("manakovSSF", E, dict(cfg, prec="complex64"))
    assert info["steps"] == 601
    out = out.T.astype(np.complex128)
    assert np.sum(np.abs(out) ** 2) / np.sum(np.abs(ref) ** 2) == pytest.approx(1.0, abs=1.5e-4)
    assert rel_l2(out, ref) <= 2e-4


def test_complex64_kernels_do_not_drift_over_a_full_span(monkeypatch):
    """1001 steps at N = 4096 on the emulated kernels against the complex128 oracle.  With twiddles rounded to one float
    (round 1; -DSSF_C64_HILO=0) the kernels lose 1.4e-7 of power per step -- the mean magnitude error of the small, fixed
    twiddle set of a 64-point pass, the same at every step -- and the reference's own complex64 path ends 6.8e-5 away from
    its complex128 result; with every factor applied as a hi + lo pair both complex64 pipelines (packed polarisation pairs,
    and one row per polarisation with SSF_C64_PACKED=0) stay below 2e-5 / 3e-5."""
    N = 4096
    E = synth_field(N, 2, 3, 0.0)
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=256e9, maxIter=10, tol=1e-5,
               prgsBar=False, Ltotal=80, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg))
    pw = lambda x: float(np.sum(np.abs(x.astype(np.complex128)) ** 2))     # noqa: E731
    res = {}
    for packed in ("1", "0"):
        monkeypatch.setenv("SSF_C64_PACKED", packed)
        out, info = eb.run("manakovSSF", E, dict(cfg, prec="complex64"), trace=False)
        assert info["steps"] == 1001
        res[packed] = (pw(out.T) / pw(ref) - 1, rel_l2(out.T, ref))
    for packed in ("1", "0"):
        assert abs(res[packed][0]) <= 2e-5 and res[packed][1] <= 3e-5, (packed, res[packed])


def test_launch_sequence_has_no_host_dependence_on_iteration_count():
    """The host enqueues [Row, Col] pairs without reading results inside a chunk: two launches
    per (step + iteration), two more per rebuilt iterate, and the surplus (no-op launches after
    the span finished) stays bounded."""
    d, cfg = load_golden("mk_fix_p8_ideal_2span")
    out, info = eb.run("manakovSSF", d["Ei"], cfg)
    useful = 2 * (info["steps"] + info["iterations"] + info["rebuilt_iterates"])
    assert useful <= info["launches"] <= 1.35 * useful + 64


def test_convergence_is_decided_one_iteration_ahead():
    """lim_i (i >= 1) is evaluated by the stage that builds iterate i (Parseval form), so every
    iteration after the first of a step is decided in advance and nothing is rebuilt."""
    for name in ("mk_fix_p8_ideal_2span", "mk_fix_p13_ideal_k2", "dbp_adp_ideal"):
        d, cfg = load_golden(name)
        _, info = eb.run(cfg["func"], d["Ei"], cfg)
        assert info["decided_ahead"] == info["iterations"] - info["steps"]
        assert info["rebuilt_iterates"] == 0


def test_convergence_at_iterate_zero_rebuilds_it_as_final():
    """A field that barely changes over a step converges at iterate 0 (lim_0 < tol): the pipeline
    has already moved on, notices it one launch later and rebuilds iterate 0 as the final one."""
    E = synth_field(1024, 2, 41, -10.0)
    cfg = dict(func="manakovSSF", alpha=0.0, D=1e-5, gamma=1e-6, Fc=193.1e12, Fs=64e9, maxIter=10, tol=1e-5,
               prgsBar=False, Ltotal=0.4, Lspan=0.2, hz=0.05, nlprMethod=False, amp=None, saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
    assert set(tr["iters"]) == {1}
    out, info = eb.run("manakovSSF", E, cfg)
    assert rel_l2(out.T, ref) <= TOL_C128
    assert list(info["iters"]) == tr["iters"] and info["rebuilt_iterates"] == info["steps"]
    np.testing.assert_allclose(np.concatenate(info["lims"]), np.concatenate(tr["lims"]), rtol=1e-6)


def test_lim0_lower_bound_gives_the_same_run_without_rereading_the_step_start_field():
    """Without a trace lim_0 is only bounded from below (one sample in sixteen, exact denominator): same field,
    same step / iteration totals as the traced (exact) run, nothing rebuilt."""
    for name in ("mk_fix_p8_ideal_2span", "mk_adp_p13_ideal_2span", "mk_fix_p13_ideal_k2"):
        d, cfg = load_golden(name)
        a, ia = eb.run(cfg["func"], d["Ei"], cfg)
        b, ib = eb.run(cfg["func"], d["Ei"], cfg, trace=False)
        assert np.array_equal(a, b)
        assert (ia["steps"], ia["iterations"], ia["nonconverged_steps"]) == (ib["steps"], ib["iterations"], ib["nonconverged_steps"])
        assert ib["rebuilt_iterates"] == 0


def test_lim0_bound_that_cannot_exclude_convergence_is_rechecked_exactly():
    """Two ways the bound falls below tol.  (a) iterate 0 really has converged: the iterate is rebuilt to measure
    lim_0 on all samples, then rebuilt as final -- two rebuilds per step, result as the traced run.  (b) it has
    not (tol between the bound and lim_0): one rebuild per step, then the iteration goes on as if nothing
    happened."""
    E = synth_field(1024, 2, 41, -10.0)
    cfg = dict(func="manakovSSF", alpha=0.0, D=1e-5, gamma=1e-6, Fc=193.1e12, Fs=64e9, maxIter=10, tol=1e-5,
               prgsBar=False, Ltotal=0.4, Lspan=0.2, hz=0.05, nlprMethod=False, amp=None, saveSpanN=[])
    a, ia = eb.run("manakovSSF", E, cfg)
    b, ib = eb.run("manakovSSF", E, cfg, trace=False)
    assert np.array_equal(a, b) and ib["iterations"] == ia["iterations"] == ia["steps"]
    assert ia["rebuilt_iterates"] == ia["steps"] and ib["rebuilt_iterates"] == 2 * ib["steps"]
    # (b): a run whose lim_0 values are known from the trace; put tol just below the smallest of them
    d, cfg = load_golden("mk_fix_p0_none")
    _, it = eb.run("manakovSSF", d["Ei"], cfg)
    lim0 = min(l[0] for l in it["lims"])
    cfg2 = dict(cfg, tol=float(lim0) * 0.6)            # bound ~ lim_0 / 4 < tol < lim_0
    tr = {}
    ref = orc.manakovSSF(d["Ei"], make_param(orc.parameters, cfg2), trace=tr)
    a, ia = eb.run("manakovSSF", d["Ei"], cfg2)
    b, ib = eb.run("manakovSSF", d["Ei"], cfg2, trace=False)
    assert list(ia["iters"]) == tr["iters"] and rel_l2(a.T, ref) <= TOL_C128
    assert np.array_equal(a, b) and ib["iterations"] == ia["iterations"]
    assert ia["rebuilt_iterates"] == 0 and 0 < ib["rebuilt_iterates"] <= ib["steps"]


def test_final_stage_stores_one_sample_in_sixteen_and_the_field_is_recovered_when_lim0_must_be_exact(monkeypatch):
    """Without a trace the final stage of a step leaves the field at one sample in sixteen (all the next step's bound of lim_0
    reads).  When the bound cannot exclude convergence at iterate 0 the whole field at the step start is recovered from E_hd
    (inverse half linear step: column / row / column launches) before lim_0 is measured on every sample; steps that follow one
    that needed the exact lim_0 find the whole field (no second recovery in a row).  Same results, bit for bit, and the same
    step / iteration counts as the traced run, which never stores sparsely; weak nonlinearity (lim_0 < tol at every step)
    recovers nothing, a run whose bound always holds recovers nothing; the short last step of a fixed-step span is foreseen
    (the step before it stores everything)."""
    d, cfg = load_golden("mk_fix_p0_none")
    _, it = eb.run("manakovSSF", d["Ei"], cfg)
    lim0 = sorted(float(l[0]) for l in it["lims"])
    hit = 0
    for f in (0.27, 0.25, 0.22):                           # tol around the bound (~ lim_0 / 4): some steps need the exact lim_0
        cfg2 = dict(cfg, tol=lim0[len(lim0) // 2] * f)
        a, ia = eb.run("manakovSSF", d["Ei"], cfg2)
        b, ib = eb.run("manakovSSF", d["Ei"], cfg2, trace=False)
        assert np.array_equal(a, b)
        assert (ia["steps"], ia["iterations"], ia["nonconverged_steps"]) == (ib["steps"], ib["iterations"], ib["nonconverged_steps"])
        assert ia["recovered_fields"] == 0 and ib["recovered_fields"] <= ib["rebuilt_iterates"] < ib["steps"]
        useful = 2 * (ib["steps"] + ib["iterations"] + ib["rebuilt_iterates"] + ib["recovered_fields"])
        assert useful <= ib["launches"] <= 1.35 * useful + 64
        hit += ib["recovered_fields"]
    assert hit > 0
    for v in ("8", "16"):                                  # both kernel families (the 8-value one samples every other workgroup)
        monkeypatch.setenv("SSF_COL_V", v)
        monkeypatch.setenv("SSF_ROW_V", v)
        bv, iv = eb.run("manakovSSF", d["Ei"], cfg2, trace=False)
        assert rel_l2(bv, b) <= 1e-13 and iv["iterations"] == ib["iterations"] and iv["recovered_fields"] > 0, v
    monkeypatch.delenv("SSF_COL_V")
    monkeypatch.delenv("SSF_ROW_V")
    tr = {}
    ref = orc.manakovSSF(d["Ei"], make_param(orc.parameters, cfg2), trace=tr)
    assert rel_l2(b.T, ref) <= TOL_C128 and ib["iterations"] == sum(tr["iters"])
    # the bound always holds / never holds / short last step (Lspan = 4.2 hz): nothing to recover
    for name, extra in (("mk_fix_p8_ideal_2span", {}), ("mk_adp_p13_ideal_2span", {}), ("mk_fix_p8_ideal_2span", dict(hz=0.33))):
        d, cfg = load_golden(name)
        cfg = dict(cfg, **extra)
        _, ib = eb.run(cfg["func"], d["Ei"], cfg, trace=False)
        assert ib["recovered_fields"] == 0 and ib["rebuilt_iterates"] == 0, name
    # complex64 (packed pairs): recovered to single precision, results inside the single-precision gate
    E = synth_field(1 << 12, 2, 19, 0.0, np.complex64)
    c64 = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5, prgsBar=False,
               Ltotal=4, Lspan=4, hz=0.25, nlprMethod=False, amp=None, saveSpanN=[], prec="complex64")
    _, it = eb.run("manakovSSF", E, c64)
    lim0 = sorted(float(l[0]) for l in it["lims"])
    hit = 0
    for f in (0.255, 0.2525):
        c = dict(c64, tol=lim0[len(lim0) // 2] * f)
        a, ia = eb.run("manakovSSF", E, c)
        b, ib = eb.run("manakovSSF", E, c, trace=False)
        assert ib["iterations"] == ia["iterations"] and rel_l2(b, a) <= 2e-6
        hit += ib["recovered_fields"]
    assert hit > 0


@pytest.mark.parametrize("maxIter", [1, 2])
def test_iteration_cap(maxIter):
    """maxIter = 1: the only iterate is final by the cap and lim_0 alone decides the warning count."""
    E = synth_field(1024, 2, 42, 10.0)
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=maxIter, tol=1e-5,
               prgsBar=False, Ltotal=2, Lspan=1, hz=0.25, nlprMethod=False, amp="ideal", saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
    out, info = eb.run("manakovSSF", E, cfg)
    assert rel_l2(out.T, ref) <= TOL_C128
    assert list(info["iters"]) == tr["iters"] and info["nonconverged_steps"] == tr["nonconverged"] == len(tr["iters"])
    np.testing.assert_allclose(np.concatenate(info["lims"]), np.concatenate(tr["lims"]), rtol=1e-6)


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32_10 (Salmon et al., kat_vectors)."""
    assert eb.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert eb.philox([f, f, f, f], [f, f]) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert eb.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_device_ase_noise_statistics_and_streams():
    n, sigma = 200000, 0.37
    a = eb.gauss(n, 0, 1, 1234, sigma)
    assert abs(a.mean()) < 4 * sigma / np.sqrt(n)
    assert np.var(a.real) == pytest.approx(sigma ** 2, rel=0.02) and np.var(a.imag) == pytest.approx(sigma ** 2, rel=0.02)
    assert abs(np.mean(a.real * a.imag)) < 4 * sigma ** 2 / np.sqrt(n)
    assert abs(np.mean(a[1:] * np.conj(a[:-1]))) < 5 * 2 * sigma ** 2 / np.sqrt(n)      # white
    assert np.array_equal(a, eb.gauss(n, 0, 1, 1234, sigma))                         # deterministic
    for other in (eb.gauss(n, 1, 1, 1234, sigma), eb.gauss(n, 0, 2, 1234, sigma), eb.gauss(n, 0, 1, 1235, sigma)):
        assert abs(np.mean(a * np.conj(other))) < 5 * 2 * sigma ** 2 / np.sqrt(n)    # rows / spans / seeds independent
    k = np.mean(np.abs(a) ** 4) / np.mean(np.abs(a) ** 2) ** 2
    assert k == pytest.approx(2.0, rel=0.03)                                         # circular Gaussian


def test_edfa_with_device_noise_on_emulated_kernels():
    d, cfg = load_golden("mk_fix_p8_ideal_2span")
    cfg = dict(cfg, amp="edfa", NF=5.0, saveSpanN=[])
    clean, _ = eb.run("manakovSSF", d["Ei"], dict(cfg, _rng_seed=0))                 # gain only
    noisy, _ = eb.run("manakovSSF", d["Ei"], dict(cfg, _rng_seed=99))
    _, p_noise = orc.edfa_noise_power(cfg["alpha"] * cfg["Lspan"], 5.0, cfg["Fc"], cfg["Fs"])
    # two spans: the first span's noise propagates (loss + gain = 1) and the second adds its own
    assert np.mean(np.abs(noisy - clean) ** 2) == pytest.approx(2 * p_noise, rel=0.15)


@pytest.mark.parametrize("name", [n for n in golden_names("edc_")])
def test_edc_overlap_save_on_emulated_kernel(name):
    """edc = overlap-save FFT filter (equalization.py:36-122 / core.py:973-1046): the HIP kernel source
    on the emulator against the reference's own outputs (any signal length, 1-D / real inputs)."""
    d, cfg = load_golden(name)
    out = eb.edc(d["Ei"], make_param(orc.parameters, cfg))
    assert out.dtype == d["out"].dtype and out.shape == d["out"].shape
    assert rel_l2(out, d["out"]) <= 1e-12


def test_edc_inverts_the_linear_channel():
    E = synth_field(1 << 14, 2, 77, 0.0)
    p = orc.parameters()
    p.Fs, p.L, p.alpha, p.D, p.Fc, p.Rs = 64e9, 100.0, 0.0, 16, 193.1e12, 32e9
    disp = orc.linearFiberChannel(E, p)
    back = eb.edc(disp, p)
    assert rel_l2(back, orc.edc(disp, p)) <= 1e-12
    # like the reference's own test (tests/test_channels.py:136-150): realign first -- the block-wise
    # convolution leaves a residual delay of a few samples -- then compare away from the edges
    mid = slice(2000, -2000)
    res = min(rel_l2(np.roll(back, k, axis=0)[mid], E[mid]) for k in range(-4, 5))
    assert res ** 2 < 0.02 and res < rel_l2(disp[mid], E[mid]) / 10


@pytest.mark.parametrize("case", ["hz_longer_than_span", "all_zero_field", "min_length", "three_pairs"])
def test_edge_cases_on_emulated_kernels(case):
    import logging
    logging.disable(logging.WARNING)
    try:
        N, ncols = 1024, 2
        kw = dict(Ltotal=2, Lspan=1, hz=0.25, saveSpanN=[])
        if case == "hz_longer_than_span":
            kw = dict(Ltotal=4, Lspan=2, hz=5.0, saveSpanN=[])
        elif case == "all_zero_field":         # lim = 0/0 = nan never passes: maxIter iterations, warnings
            kw = dict(Ltotal=1, Lspan=1, hz=0.5, maxIter=3, saveSpanN=[])
        elif case == "min_length":
            N = 256
        elif case == "three_pairs":
            ncols = 6
        E = synth_field(N, ncols, 90, 6.0) if case != "all_zero_field" else np.zeros((N, 2), complex)
        cfg = dict(dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5,
                        prgsBar=False, nlprMethod=False, amp="ideal"), **kw)
        tr = {}
        with np.errstate(all="ignore"):
            ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
        out, info = eb.run("manakovSSF", E, cfg)
        if case == "all_zero_field":
            assert np.all(out == 0) and info["nonconverged_steps"] == tr["nonconverged"] == info["steps"]
        else:
            assert rel_l2(out.T, ref) <= TOL_C128
        assert list(info["iters"]) == tr["iters"]
    finally:
        logging.disable(logging.NOTSET)


# ------------------------------------------------------------------ lengths with factors 3 and 5
@pytest.mark.parametrize("N,adaptive,K", [(9600, False, 1), (9600, True, 2), (48000, False, 1), (32000, True, 1), (76800, False, 1)])
def test_mixed_radix_rows_vs_oracle(N, adaptive, K):
    """N = 2^a * m (m odd, 5-smooth): power-of-two column transforms with ragged tiles, mixed-radix row
    transforms in LDS (mixed_fft.h).  9600 = 2^7 * 75, 48000 = 2^7 * 375, 32000 = 2^8 * 125; 76800 = 2^10 * 75: more than 256 row
    workgroups, so the later ones deal their butterflies from the middle of the row (row_mixed_body, SSF_MIX_ROT)."""
    assert eb.load().emu_supported(N, 1)
    E = synth_field(N, 2 * K, 3, 8.0)
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5, prgsBar=False,
               Ltotal=1.3, Lspan=0.65, hz=0.25, nlprMethod=adaptive, maxNlinPhaseRot=2e-2, amp="ideal", saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
    out, info = eb.run("manakovSSF", E, cfg)
    assert rel_l2(out.T, ref) <= TOL_C128 and list(info["iters"]) == tr["iters"]
    np.testing.assert_allclose(info["hz"], tr["hz"], rtol=1e-9)
    out2, info2 = eb.run("manakovSSF", E, cfg, trace=False)
    assert np.array_equal(out, out2) and info2["iterations"] == info["iterations"]


def test_mixed_radix_other_entry_points():
    N = 9600
    E = synth_field(N, 2, 5, 6.0)
    cfg = dict(func="ssfm", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, prgsBar=False, Ltotal=2.0, Lspan=1.0, hz=0.25,
               amp="ideal", saveSpanN=[])
    s = E[:, 0].copy()
    out, _ = eb.run("ssfm", s, cfg)
    assert rel_l2(out.reshape(-1), orc.ssfm(s, make_param(orc.parameters, cfg))) <= TOL_C128
    mk = dict(func="manakovDBP", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5, prgsBar=False,
              Ltotal=2.0, Lspan=1.0, hz=0.5, nlprMethod=False, amp="ideal", saveSpanN=[])
    out, info = eb.run("manakovDBP", E, mk)
    tr = {}
    ref = orc.manakovDBP(E, make_param(orc.parameters, mk), trace=tr)
    assert rel_l2(out.T, ref) <= TOL_C128 and list(info["iters"]) == tr["iters"]
    lp = orc.parameters()
    lp.Fs, lp.L, lp.alpha, lp.D, lp.Fc = 512e9, 3.0, 0.2, 16, 193.1e12
    assert rel_l2(eb.linear_channel(E, 512e9, 193.1e12, 0.2, 16, 3.0), orc.linearFiberChannel(E, lp)) < 1e-13
    c64 = dict(mk, func="manakovSSF", prec="complex64")
    out, _ = eb.run("manakovSSF", E.astype(np.complex64), c64)
    ref = orc.manakovSSF(E, make_param(orc.parameters, dict(mk, func="manakovSSF")))
    assert rel_l2(out.T.astype(np.complex128), ref) <= 5e-4


def test_mixed_radix_transform_lengths():
    """The in-LDS transform alone: forward (through the digit-reversal map) and round trip against numpy."""
    import ctypes as C
    e = eb.load()
    e.emu_mixed_fft.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(0)
    for L in (2, 3, 5, 6, 9, 10, 12, 15, 16, 20, 25, 27, 45, 60, 75, 120, 125, 128, 225, 240, 375, 405, 600, 625, 1000, 1125, 1875,
              2025, 3125, 3375, 3750):
        x = rng.normal(size=(2, L)) + 1j * rng.normal(size=(2, L))
        y, z = np.empty_like(x), np.empty_like(x)
        assert e.emu_mixed_fft(L, 2, -1, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)) == 0
        assert e.emu_mixed_fft(L, 2, +1, x.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p)) == 0
        assert rel_l2(y, np.fft.fft(x, axis=1)) < 1e-14 and rel_l2(z / L, x) < 1e-14, L
    assert e.emu_mixed_fft(7, 1, -1, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)) == -6
    # radix plans made for a thread count (the engine's choice: mix_make_plan with the threads a row gets) pick other
    # radices and orders than largest-first: every one must give the same transform
    e.emu_mixed_fft_t.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    for L in (75, 375, 768, 1280, 1536, 1875, 2025, 3072, 3125, 3750, 5120, 5625, 6000, 6144, 7500):
        x = rng.normal(size=(1, L)) + 1j * rng.normal(size=(1, L))
        y, z = np.empty_like(x), np.empty_like(x)
        for T in (64, 128, 256):
            assert e.emu_mixed_fft_t(L, 1, -1, T, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)) == 0
            assert e.emu_mixed_fft_t(L, 1, +1, T, x.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p)) == 0
            assert rel_l2(y, np.fft.fft(x, axis=1)) < 1e-14 and rel_l2(z / L, x) < 1e-14, (L, T)


def test_mixed_radix_planner_covers_every_row_length_the_split_can_ask_for():
    """mix_make_plan for every 5-smooth row length 64 .. 8192 and the thread counts the engine uses: the radices multiply
    to L, there are at most six passes, every radix is implemented, and the last pass (which also applies the row
    operator, mix_apply_op) has a radix <= 16."""
    import ctypes as C
    e = eb.load()
    e.emu_mix_plan.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    ok_radices = {25, 20, 16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2}
    lengths = sorted({2 ** a * 3 ** b * 5 ** c for a in range(14) for b in range(9) for c in range(6)
                      if 64 <= 2 ** a * 3 ** b * 5 ** c <= 8192})
    assert len(lengths) > 100
    r = (C.c_int * 8)()
    for L in lengths:
        for T in (0, 64, 128, 256, 512):
            n = e.emu_mix_plan(L, T, r)
            rad = list(r[:n])
            assert 1 <= n <= 6 and int(np.prod(rad)) == L and set(rad) <= ok_radices and rad[-1] <= 16, (L, T, rad)
    assert e.emu_mix_plan(7 * 64, 128, r) == 0


# ---- round 3: eight values per thread (128-register kernels, four waves per SIMD) ---------------------------------
@pytest.mark.parametrize("N,prec", [(1 << 12, "complex128"), (1 << 14, "complex128"), (1 << 16, "complex128"),
                                    (1 << 17, "complex128"), (1 << 14, "complex64")])
@pytest.mark.parametrize("which", ["rows", "cols", "both", "sixteen"])
def test_rows_with_eight_values_per_thread(monkeypatch, N, prec, which):
    """SSF_ROW_V=8: radix-8 row passes (a fourth LDS exchange per 4096-point transform), operator on the eight bins
    N/8 apart of a last-pass butterfly.  Same results as the oracle, same iteration counts; ssfm and the linear channel too."""
    # (fields that do not fill the chip get the 8-value kernels by default: force each combination, the 16-value pair included)
    monkeypatch.setenv("SSF_ROW_V", "8" if which in ("rows", "both") else "16")
    monkeypatch.setenv("SSF_COL_V", "8" if which in ("cols", "both") else "16")
    dt = np.complex64 if prec == "complex64" else np.complex128
    E = synth_field(N, 2, 41, 8.4).astype(dt)
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5, prgsBar=False,
               Ltotal=0.48, Lspan=0.24, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[], prec=prec)
    tr = {}
    q = make_param(orc.parameters, cfg)
    q.prec = dt
    ref = orc.manakovSSF(E, q, trace=tr)
    out, info = eb.run("manakovSSF", E, cfg)
    if prec == "complex128":
        assert rel_l2(out.T, ref) <= TOL_C128 and list(info["iters"]) == tr["iters"]
        p = orc.parameters()
        p.Fs, p.L, p.alpha, p.D, p.Fc = 512e9, 3.0, 0.2, 16, 193.1e12
        assert rel_l2(eb.linear_channel(E, 512e9, 193.1e12, 0.2, 16, 3.0), orc.linearFiberChannel(E, p)) < 1e-13
    else:
        assert rel_l2(out.T, ref) <= 5e-4


# ---- round 3: independent units in one launch sequence ---------------------------------------------------------------
@pytest.mark.parametrize("adaptive", [False, True])
@pytest.mark.parametrize("N,prec", [(1 << 12, "complex128"), (6000, "complex128"), (1 << 12, "complex64")])
def test_independent_units_share_the_launches_not_the_decisions(monkeypatch, N, prec, adaptive):
    """Three fields of very different power as ONE batch of independent units (grid.y = units): every unit keeps its own
    control block, step sizes and convergence decisions, so the batch is bit-equal to three separate runs -- and not to the
    coupled K = 3 call of the reference, which shares max(phi) and the norms."""
    dt = np.complex64 if prec == "complex64" else np.complex128
    fields = [synth_field(N, 2, 50 + u, p).astype(dt) for u, p in enumerate((-12.0, 6.0, 18.0))]
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=64e9, maxIter=10, tol=1e-5, prgsBar=False,
               Ltotal=8, Lspan=4, hz=1.0, nlprMethod=adaptive, maxNlinPhaseRot=2e-2, amp="edfa", NF=4.5, saveSpanN=[], prec=prec)
    if N == 6000 and adaptive:                                      # (the 18 dBm unit takes ~40 adaptive steps per km on the emulator)
        cfg.update(Ltotal=2, Lspan=1)
    monkeypatch.delenv("SSF_EMU_UNITS", raising=False)
    alone, steps = [], []
    cfg["_rng_seed"] = 1234                                         # device ASE noise: unit u draws rows 2u, 2u + 1 of the stream
    for u, E in enumerate(fields):
        out, info = eb.run("manakovSSF", E, dict(cfg, _rng_row_offset=2 * u), trace=False)
        alone.append(out)
        steps.append((info["steps"], info["iterations"]))
    assert len({s for s in steps}) > 1                              # the units really need different step / iteration counts
    monkeypatch.setenv("SSF_EMU_UNITS", "3")
    out, info = eb.run("manakovSSF", np.concatenate(fields, axis=1), cfg, trace=False)
    for u in range(3):
        assert np.array_equal(out[2 * u:2 * u + 2], alone[u]), u
    assert info["steps"] == sum(s[0] for s in steps) and info["iterations"] == sum(s[1] for s in steps)
    if prec == "complex128" and not adaptive:
        ref = orc.manakovSSF(fields[1], make_param(orc.parameters, dict(cfg, amp="ideal")))
        cfg2 = dict(cfg, amp="ideal")
        out2, _ = eb.run("manakovSSF", np.concatenate(fields, axis=1), cfg2, trace=False)
        assert rel_l2(out2[2:4].T, ref) <= TOL_C128


@pytest.mark.parametrize("N", [1500, 3000, 6000, 375, 8100, 96])
@pytest.mark.parametrize("in_place", [False, True])
def test_one_launch_linear_step_at_short_smooth_lengths(N, in_place):
    """Lengths with fewer than four factors of two (1500 = 4 x 375 ...) have no column x row split; the general-length engine
    used to give them Bluestein transforms (three launches of a 4096- or 8192-point convolution per linear step).  5-smooth
    rows of up to 8192 values are transformed directly in LDS instead: FFT . H . IFFT in one launch, operator from the bin index."""
    rng = np.random.default_rng(N)
    x = rng.normal(size=(2, N)) + 1j * rng.normal(size=(2, N))
    hzh, lin_a, lin_b, w_scale = 0.04, -0.023, -1.02e-23, 2 * np.pi * 64e9
    w = w_scale * np.fft.fftfreq(N)
    ref = np.fft.ifft(np.fft.fft(x, axis=1) * np.exp((lin_a + 1j * lin_b * w ** 2) * hzh), axis=1)
    assert rel_l2(eb.rows_lin(x, hzh, lin_a, lin_b, w_scale, in_place), ref) <= 1e-13
    ref32 = ref.astype(np.complex64)
    assert rel_l2(eb.rows_lin(x.astype(np.complex64), hzh, lin_a, lin_b, w_scale, in_place), ref32) <= 2e-6


# ---- round 4: twiddle tables (fused_kernels.h: TwSrc) ----------------------------------------------------------------------
@pytest.mark.parametrize("l1,prec", [(4, "complex128"), (10, "complex128"), (4, "complex64"), (10, "complex64"), (5, "complex128")])
def test_three_pass_transforms_with_the_lds_twiddle_table_and_kept_bases(monkeypatch, l1, prec):
    """Rows of 1024 (2^14 = 16 x 1024: passes 16 . 4 . 16) and columns of 1024 (1024 x 16) are three-pass transforms: in single
    precision the rows' pass 1 takes its hi + lo factors from the workgroup's LDS table (fused_kernels.h: TwSrc); every other
    pass generates them from a base that the inverse transform keeps for the forward one (conjugate); 2^5 x 512 rows: passes
    16 . 2 . 16.  Same results as the oracle, same iteration counts; the linear channel (one-row kernels) too."""
    monkeypatch.setenv("SSF_SPLIT_L1", str(l1))
    monkeypatch.setenv("SSF_ROW_V", "16")
    monkeypatch.setenv("SSF_COL_V", "16")
    N = 1 << 14
    dt = np.complex64 if prec == "complex64" else np.complex128
    E = synth_field(N, 2, 43, 8.4).astype(dt)
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5, prgsBar=False,
               Ltotal=0.48, Lspan=0.24, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[], prec=prec)
    tr = {}
    q = make_param(orc.parameters, cfg)
    q.prec = dt
    ref = orc.manakovSSF(E, q, trace=tr)
    out, info = eb.run("manakovSSF", E, cfg)
    if prec == "complex128":
        assert rel_l2(out.T, ref) <= TOL_C128 and list(info["iters"]) == tr["iters"]
        p = orc.parameters()
        p.Fs, p.L, p.alpha, p.D, p.Fc = 512e9, 3.0, 0.2, 16, 193.1e12
        assert rel_l2(eb.linear_channel(E, 512e9, 193.1e12, 0.2, 16, 3.0), orc.linearFiberChannel(E, p)) < 1e-13
    else:
        assert rel_l2(out.T, ref) <= 5e-5


# ---- round 5: stage-specialised column kernels along the predicted stage sequence (FusedCore::run_span, fused_kernels.h: stage_group) ----
@pytest.mark.parametrize("prec", ["complex128", "complex64"])
@pytest.mark.parametrize("kw", [dict(), dict(nlprMethod=True, maxNlinPhaseRot=2e-3), dict(p=-20.0), dict(hz=0.3), dict(maxIter=1),
                                dict(p=11.0, Lspan=6.0, Ltotal=6.0, alpha=2.0)],
                         ids=["fixed", "adaptive", "weak_nonlinearity", "long_steps", "one_iteration", "iteration_count_falls"])
def test_stage_specialised_column_kernels_follow_the_state(monkeypatch, prec, kw):
    """The host enqueues H | ADV | FIN column kernels along the stage sequence it predicts from the latest step's iteration
    count; a kernel whose stage the state does not ask for forwards the control block untouched.  Whatever the guess -- the
    iteration count falling along a lossy span, convergence at iterate 0 as the rule (rebuilds ride in the H kernel; the call
    falls back to the general kernel), the rounding-sized last step of a span -- the results are those of the one general
    kernel (SSF_COL_SPLIT=0) bit for bit, with the same step and iteration counts, at a bounded cost in idle launches."""
    monkeypatch.setenv("SSF_COL_V", "16")
    monkeypatch.setenv("SSF_ROW_V", "16")
    kw = dict(kw)
    dt = np.complex64 if prec == "complex64" else np.complex128
    E = synth_field(1 << 12, 2, 43, kw.pop("p", 8.4)).astype(dt)
    cfg = dict(func="manakovSSF", alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Fs=512e9, maxIter=10, tol=1e-5, prgsBar=False,
               Ltotal=1.6, Lspan=0.8, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[], prec=prec)
    cfg.update(kw)
    res = {}
    for split in ("0", "1"):
        monkeypatch.setenv("SSF_COL_SPLIT", split)
        res[split] = eb.run("manakovSSF", E, cfg, trace=False)
    (a, ia), (b, ib) = res["0"], res["1"]
    assert np.array_equal(a, b)
    for k in ("steps", "iterations", "nonconverged_steps", "rebuilt_iterates", "recovered_fields"):
        assert ia[k] == ib[k], k
    assert ib["launches"] <= 1.5 * ia["launches"] + 96, (ia["launches"], ib["launches"])
    if "alpha" in kw:                                   # the case is what it says: the iteration count changes inside the span
        _, it = eb.run("manakovSSF", E, cfg)
        assert len(set(it["iters"])) > 1
