"""Long runs against vectors the REFERENCE produced (tools/gen_golden.py long -> tests/golden/long_*.npz).

SURVEY.md 8c asks for identical per-step iteration counts across the 3 -> 2 crossover of a lossy span; a 50-step
test never sees it.  The long_* fixtures hold, for 501 ... 1001-step reference runs (config-2 shape at 2^14 and 2^16,
adaptive step, two coupled pairs, back-propagation, a 2^a 3^b 5^c length), the per-step iteration list, every lim
value and a decimated output plus a seeded projection of the whole output; the input is regenerated from the seeded
recipe.  long_c64drift_* hold the reference's own complex64 deviation from its complex128 result after 1 / 2 / 4 /
10 spans (BASELINE config 3 is 10 spans = 10 010 steps): the yardstick for the single-precision gate.

CPU (not gpu): the oracle reproduces one of them bit for bit (it is what the other GPU tests compare with).
GPU: the HIP path reproduces all of them (rel-L2 <= 1e-10, identical iteration lists, lims to 1e-6) and its complex64
path stays closer to the reference's complex128 result than the reference's own complex64 path does."""
import numpy as np
import pytest

from helpers import golden_names, load_golden, make_param, rel_l2, synth_field
from oracle import ssf_oracle as orc

LONG = [n for n in golden_names("long_") if "c64drift" not in n and n != "long_c3_n22"]   # (long_c3_n22: its own test below)
DRIFT = golden_names("long_c64drift_")


def projection(out, seed=4242):
    rng = np.random.default_rng(seed)
    r = (rng.normal(size=out.shape[0]) + 1j * rng.normal(size=out.shape[0])) / np.sqrt(2)
    return out.astype(np.complex128).T @ r


def _input(cfg, dtype=np.complex128):
    N, ncols, seed, p_dbm = cfg["synth"]
    return synth_field(int(N), int(ncols), int(seed), float(p_dbm), dtype)


def _run_cfg(cfg):
    return {k: v for k, v in cfg.items() if k not in ("synth", "dec", "steps")}


def _check(d, cfg, out, iters, lims, tol=1e-10):
    assert list(iters) == list(d["iters"])                                   # every step, crossovers included
    flat = np.concatenate([np.asarray(r, dtype=float) for r in lims]) if len(lims) and np.ndim(lims[0]) else np.asarray(lims)
    np.testing.assert_allclose(flat, d["lims"], rtol=1e-6, atol=1e-15)       # (atol: the last, rounding-sized step's lim is ~4e-12)
    dec = int(cfg["dec"])
    assert rel_l2(out[::dec], d["out_dec"]) <= tol
    np.testing.assert_allclose(np.sum(np.abs(out) ** 2, axis=0), d["out_power"], rtol=1e-9)
    assert np.max(np.abs(projection(out) - d["out_proj"])) <= 10 * tol * np.sqrt(np.sum(d["out_power"]))


def test_fixtures_cover_an_iteration_crossover():
    """The point of the long vectors: the iteration count changes inside the run (3 -> 2 as the power decays)."""
    assert LONG, "run tools/gen_golden.py long"
    d, _ = load_golden("long_c2_n14")
    it = d["iters"].astype(int)
    assert len(it) == 1001 and sorted(set(it[:-1])) == [2, 3] and np.count_nonzero(np.diff(it[:-1])) >= 1
    assert float(d["margin"]) > 1e-7                      # no lim within rounding distance of tol: counts are well defined


def test_oracle_reproduces_the_reference_over_a_full_span():
    d, cfg = load_golden("long_c2_n14")
    tr = {}
    out = orc.manakovSSF(_input(cfg), make_param(orc.parameters, _run_cfg(cfg)), trace=tr)
    assert tr["iters"] == list(d["iters"])
    # (the norms are BLAS nrm2 calls: their last bit depends on the BLAS thread count, the field does not)
    np.testing.assert_allclose(np.concatenate([np.asarray(r, dtype=float) for r in tr["lims"]]), d["lims"], rtol=1e-12)
    assert np.array_equal(out[:: int(cfg["dec"])], d["out_dec"])
    np.testing.assert_allclose(projection(out), d["out_proj"], rtol=1e-12)          # (a BLAS product: see above)


@pytest.mark.parametrize("name", ["long_nb_n200000", "long_nb_n2000000"])
def test_oracle_reproduces_the_reference_at_its_own_benchmark_lengths(name):
    """long_nb_*: the REFERENCE on its published GPU benchmark's settings (examples/benchmarck_GPU_processing.ipynb: Fs 128 GS/s, adaptive
    step, maxIter 5) at 2e5 samples (a full 50 km span) and at its top size 2e6 = 2^7 x 5^6 (the first 12 km) -- tools/gen_golden.py
    notebook.  The oracle is held to them bit for bit here; the HIP path -- 2e6 on the mixed-radix column stage -- under -m gpu
    (test_hip_reproduces_the_reference_long_runs)."""
    d, cfg = load_golden(name)
    tr = {}
    out = orc.manakovSSF(_input(cfg), make_param(orc.parameters, _run_cfg(cfg)), trace=tr)
    assert tr["iters"] == list(d["iters"])
    np.testing.assert_allclose(np.concatenate([np.asarray(r, dtype=float) for r in tr["lims"]]), d["lims"], rtol=1e-12)
    np.testing.assert_allclose(tr["hz"], d["hz"], rtol=0, atol=0) if "hz" in d else None
    assert np.array_equal(out[:: int(cfg["dec"])], d["out_dec"])
    np.testing.assert_allclose(projection(out), d["out_proj"], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", LONG)
def test_hip_reproduces_the_reference_long_runs(name):
    import opticommpy_amd as oa
    from opticommpy_amd import models
    d, cfg = load_golden(name)
    f = {"manakovSSF": oa.manakovSSF, "manakovDBP": oa.manakovDBP}[cfg["func"]]
    for engine in ("fused", "rocfft") if name == "long_c2_n14" else ("auto",):
        oa.set_engine(engine)
        try:
            out = f(_input(cfg), make_param(oa.parameters, _run_cfg(cfg)), _trace=True)
            run = dict(models.last_run)
        finally:
            oa.set_engine("auto")
        assert run["steps"] == len(d["iters"])
        if name.startswith("long_nb_"):
            assert run["pipeline"] == "fused-device"          # (2 000 000 = 2^7 x 5^6: the mixed-radix column stage, not Bluestein)
        _check(d, cfg, out, run["iters"], run["lims"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", DRIFT)
def test_c64_stays_closer_to_the_reference_c128_result_than_the_reference_c64_path(name):
    """10 spans x 1001 steps.  complex128: the HIP result IS the reference's (1e-9 after 10 010 steps).  complex64: the
    reference's own single-precision path is 5e-4 away from its double-precision result by then (same rounded twiddles
    at every step: the error grows linearly); the packed-pair kernels apply twiddles and operator as hi + lo pairs and
    must stay inside the SURVEY 8c gate of 5e-4 over the whole run, inside the reference's own deviation at every
    checkpoint, and conserve the power to 2e-4."""
    import opticommpy_amd as oa
    d, cfg = load_golden(name)
    dec = int(cfg["dec"])
    E64 = _input(cfg, np.complex64)
    run = _run_cfg(cfg)
    o128 = oa.manakovSSF(E64.astype(np.complex128), make_param(oa.parameters, dict(run, prec="complex128")))
    assert o128.shape == (E64.shape[0], 8)
    assert rel_l2(o128[::dec], d["out128_dec"]) <= 1e-9
    assert np.max(np.abs(projection(o128) - d["out128_proj"])) <= 1e-8 * np.sqrt(np.sum(d["out128_power"]))
    o64 = oa.manakovSSF(E64, make_param(oa.parameters, dict(run, prec="complex64")))
    assert o64.dtype == np.complex64
    for i, span in enumerate(d["spans"]):
        a = o64[:, 2 * i:2 * i + 2].astype(np.complex128)
        b = o128[:, 2 * i:2 * i + 2]
        dev = rel_l2(a, b)
        pr = np.sum(np.abs(a) ** 2) / np.sum(np.abs(b) ** 2)
        assert dev <= 5e-4, (span, dev)
        assert dev <= max(float(d["ref_c64_rel_l2"][i]), 5e-5), (span, dev, float(d["ref_c64_rel_l2"][i]))
        assert abs(pr - 1) <= 2e-4, (span, pr)


# ------------------------------------------------------------------------------------------ BASELINE config 3's own field
# tests/golden/wl_cfg3_n22.npz (tools/gen_golden.py cfg3): the REFERENCE on config 3's field -- 2^22 samples, seed 3, 8.4 dBm,
# complex64 samples -- for six passes of the step loop, in complex128 (samples cast up) and in its complex64 mode.  The 2^22
# split (1024 x 4096) is a template instantiation of its own: this is what pins it, in both precisions, and what the
# full-length single-precision tests below (HIP complex64 against HIP complex128) stand on.
def test_oracle_reproduces_the_reference_on_config3s_own_field():
    d, cfg = load_golden("wl_cfg3_n22")
    E64 = _input(cfg, np.complex64)
    tr = {}
    out = orc.manakovSSF(E64.astype(np.complex128), make_param(orc.parameters, dict(_run_cfg(cfg), prec="complex128")), trace=tr)
    assert tr["iters"] == list(d["iters128"])
    np.testing.assert_allclose(np.concatenate([np.asarray(r, dtype=float) for r in tr["lims"]]), d["lims128"], rtol=1e-12)
    assert np.array_equal(out[:: int(cfg["dec"])], d["out128_dec"])


@pytest.mark.gpu
def test_config3_first_steps_against_the_reference_at_full_size():
    """HIP complex128 at 2^22: the reference's iteration list, lims to 1e-6, field to 1e-10 (decimated output, per-column power,
    a seeded projection of all 2^23 output samples).  HIP complex64 (packed pairs): inside the 5e-4 gate of SURVEY 8c against
    the reference's complex128 result AND against the reference's own complex64 result."""
    import opticommpy_amd as oa
    from opticommpy_amd import models
    d, cfg = load_golden("wl_cfg3_n22")
    dec, run = int(cfg["dec"]), _run_cfg(cfg)
    E64 = _input(cfg, np.complex64)
    out = oa.manakovSSF(E64.astype(np.complex128), make_param(oa.parameters, dict(run, prec="complex128")), _trace=True)
    r128 = dict(models.last_run)
    assert r128["engine"] == "fused" and r128["steps"] == int(cfg["steps"])
    assert list(r128["iters"]) == list(d["iters128"])
    flat = np.concatenate([np.asarray(r, dtype=float) for r in r128["lims"]])
    np.testing.assert_allclose(flat, d["lims128"], rtol=1e-6, atol=1e-15)
    assert rel_l2(out[::dec], d["out128_dec"]) <= 1e-10
    np.testing.assert_allclose(np.sum(np.abs(out) ** 2, axis=0), d["out128_power"], rtol=1e-9)
    scale = np.sqrt(np.sum(d["out128_power"]))
    assert np.max(np.abs(projection(out) - d["out128_proj"])) <= 1e-9 * scale
    o64 = oa.manakovSSF(E64, make_param(oa.parameters, dict(run, prec="complex64")), _trace=True)
    r64 = dict(models.last_run)
    assert o64.dtype == np.complex64 and r64["steps"] == int(cfg["steps"])
    assert list(r64["iters"]) == list(d["iters64"])                     # (no lim near tol in these six steps)
    assert rel_l2(o64[::dec], d["out128_dec"]) <= 5e-4
    assert rel_l2(o64[::dec], d["out64_dec"]) <= 5e-4
    assert np.max(np.abs(projection(o64) - d["out128_proj"])) <= 5e-4 * scale
    np.testing.assert_allclose(np.sum(np.abs(o64.astype(np.complex128)) ** 2, axis=0), d["out128_power"], rtol=2e-4)
    # and the HIP complex64 result is at least as close to the complex128 truth as the reference's complex64 path (9.3e-7 here)
    assert rel_l2(o64[::dec], d["out128_dec"]) <= max(3.0 * float(d["ref_c64_rel_l2"]), 5e-6)


@pytest.mark.gpu
def test_config3_full_span_against_the_reference_at_full_size():
    """BASELINE config 3's own field (N = 2^22, seed 3, 8.4 dBm, complex64 samples) over one FULL span -- 80 km, hz 0.08: 1001 passes
    of the step loop, through the 3 -> 2 iteration crossover (step 493 of the reference in both precisions) -- against the
    REFERENCE (tests/golden/long_c3_n22.npz, tools/gen_golden.py long_c3: 71 + 64 minutes of reference time).  HIP complex128: the
    reference's iteration list step for step, every lim to 1e-6, the field to 1e-10 (decimated output, per-column power, a seeded
    projection of all 2^23 output samples).  HIP complex64 (packed pairs): inside the 5e-4 gate of SURVEY 8c against the
    reference's complex128 result AND its own complex64 result, power within 2e-4, the reference's iteration list up to a flip
    at the crossover step (SURVEY 8c allows it in single precision; the totals are reported), and at least as close to the
    complex128 truth as the reference's complex64 path (1.1e-4 here)."""
    import opticommpy_amd as oa
    from opticommpy_amd import models
    d, cfg = load_golden("long_c3_n22")
    dec, run = int(cfg["dec"]), _run_cfg(cfg)
    E64 = _input(cfg, np.complex64)
    out = oa.manakovSSF(E64.astype(np.complex128), make_param(oa.parameters, dict(run, prec="complex128")), _trace=True)
    r128 = dict(models.last_run)
    assert r128["engine"] == "fused" and r128["steps"] == int(cfg["steps"]) == 1001
    assert list(r128["iters"]) == list(d["iters128"])
    flat = np.concatenate([np.asarray(r, dtype=float) for r in r128["lims"]])
    np.testing.assert_allclose(flat, d["lims128"], rtol=1e-6, atol=1e-15)
    assert rel_l2(out[::dec], d["out128_dec"]) <= 1e-10
    np.testing.assert_allclose(np.sum(np.abs(out) ** 2, axis=0), d["out128_power"], rtol=1e-9)
    scale = np.sqrt(np.sum(d["out128_power"]))
    assert np.max(np.abs(projection(out) - d["out128_proj"])) <= 1e-9 * scale
    o64 = oa.manakovSSF(E64, make_param(oa.parameters, dict(run, prec="complex64")), _trace=True)
    r64 = dict(models.last_run)
    assert o64.dtype == np.complex64 and r64["steps"] == 1001
    flips = int(np.count_nonzero(np.asarray(r64["iters"]) != d["iters64"]))
    assert flips <= 2, flips
    dev128, dev64 = rel_l2(o64[::dec], d["out128_dec"]), rel_l2(o64[::dec], d["out64_dec"])
    print(f"config 3, one span: HIP c128 vs reference {rel_l2(out[::dec], d['out128_dec']):.2e}; HIP c64 vs reference c128 {dev128:.2e}, "
          f"vs reference c64 {dev64:.2e} (reference c64 vs c128 {float(d['ref_c64_rel_l2_dec']):.2e}); iteration flips {flips}")
    assert dev128 <= 5e-4 and dev64 <= 5e-4
    assert dev128 <= max(float(d["ref_c64_rel_l2_dec"]), 5e-5)
    assert np.max(np.abs(projection(o64) - d["out128_proj"])) <= 5e-4 * scale
    np.testing.assert_allclose(np.sum(np.abs(o64.astype(np.complex128)) ** 2, axis=0), d["out128_power"], rtol=2e-4)


@pytest.mark.gpu
def test_c64_drift_over_config3_step_count():
    """10 010 steps (10 x 80 km, hz 0.08) at N = 2^18: complex64 against the complex128 HIP run."""
    import opticommpy_amd as oa
    N = 1 << 18
    E = synth_field(N, 2, 7, 0.0, np.complex64)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
               Ltotal=800, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    outs = {}
    for prec in ("complex128", "complex64"):
        outs[prec] = oa.manakovSSF(E.astype(prec), make_param(oa.parameters, dict(cfg, prec=prec)))
    a, b = outs["complex64"].astype(np.complex128), outs["complex128"]
    assert rel_l2(a, b) <= 5e-4
    assert abs(np.sum(np.abs(a) ** 2) / np.sum(np.abs(b) ** 2) - 1) <= 2e-4


@pytest.mark.gpu
def test_config3_at_full_size_and_full_length():
    """BASELINE config 3 itself: N = 2^22, 10 x 80 km, hz 0.08 (10 010 steps), complex64 against the complex128 HIP run of
    the same field (whose kernels are pinned to the reference at 2^20 over a full span, long_c2_n20): the single-precision
    gate of SURVEY 8c (5e-4) after every span that is saved, power within 2e-4, identical iteration totals up to the
    crossover steps.  About 10 s of device time."""
    import opticommpy_amd as oa
    from opticommpy_amd import models
    N = 1 << 22
    E = synth_field(N, 2, 3, 8.4, np.complex64)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
               Ltotal=800, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[1, 5, 10])
    outs, its = {}, {}
    for prec in ("complex128", "complex64"):
        outs[prec] = oa.manakovSSF(E.astype(prec), make_param(oa.parameters, dict(cfg, prec=prec)))
        assert models.last_run["steps"] == 10010
        its[prec] = int(models.last_run["iterations"])
    assert outs["complex64"].dtype == np.complex64 and outs["complex64"].shape == (N, 6)
    for i in range(3):
        a, b = outs["complex64"][:, 2 * i:2 * i + 2].astype(np.complex128), outs["complex128"][:, 2 * i:2 * i + 2]
        assert rel_l2(a, b) <= 5e-4, (i, rel_l2(a, b))
        assert abs(np.sum(np.abs(a) ** 2) / np.sum(np.abs(b) ** 2) - 1) <= 2e-4
    assert abs(its["complex64"] - its["complex128"]) <= 20            # a flip per span at the 3 -> 2 crossover step, no more
