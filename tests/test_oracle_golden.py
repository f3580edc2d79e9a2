"""Pin the CPU oracle (oracle/ssf_oracle.py) bit-for-bit against the golden
vectors produced by importing the reference (tools/gen_golden.py), and against
the reference's own TestSSFM properties (reference tests/test_channels.py:154-224)."""
import numpy as np
import pytest

from helpers import golden_names, load_golden, make_param, split_iters
from oracle import ssf_oracle as orc

ALL = [n for n in golden_names() if not n.startswith(("rx_", "tx_", "long_", "wl_", "chain_", "bfc_", "mix_"))]   # rx_*, tx_*: their own test files


def _run(cfg, Ei, trace):
    p = make_param(orc.parameters, cfg)
    if cfg["func"] == "edc":
        return orc.edc(Ei, p), p
    fn = {"ssfm": orc.ssfm, "manakovSSF": orc.manakovSSF, "manakovDBP": orc.manakovDBP}[cfg["func"]]
    return fn(Ei, p, trace=trace), p


def test_fixture_inventory():
    assert len(ALL) >= 25
    assert any(n.startswith("dbp_") for n in ALL) and any(n.startswith("ssfm_") for n in ALL)


@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_reference_bit_for_bit(name):
    d, cfg = load_golden(name)
    trace = {}
    out, p = _run(cfg, d["Ei"], trace)
    assert out.dtype == d["out"].dtype and out.shape == d["out"].shape
    assert np.array_equal(out, d["out"]), f"max abs diff {np.max(np.abs(out - d['out']))}"
    if "lims" in d:
        flat = [v for step in trace["lims"] for v in step]
        assert np.array_equal(np.array(flat), d["lims"])
        assert trace["iters"] == list(d["iters"])
        assert trace["iters"] == split_iters(flat, p.tol, p.maxIter)


def test_ref_property_gamma0_equals_linear_channel():
    d, cfg = load_golden("ssfm_ref_gamma0")
    out, _ = _run(cfg, d["Ei"], None)
    lp = orc.parameters()
    lp.L, lp.alpha, lp.D, lp.Fc, lp.Fs = cfg["Ltotal"], cfg["alpha"], cfg["D"], cfg["Fc"], cfg["Fs"]
    lin = orc.linearFiberChannel(d["Ei"], lp)
    assert np.array_equal(lin, d["extra_linear"])
    np.testing.assert_allclose(out, lin, atol=1e-12)


def test_ref_property_spm_broadens_spectrum():
    d, cfg = load_golden("ssfm_ref_spm")
    out, _ = _run(cfg, d["Ei"], None)
    out0, _ = _run(dict(cfg, gamma=0), d["Ei"], None)
    assert np.array_equal(out0, d["extra_gamma0"])
    assert not np.allclose(np.abs(np.fft.fft(out)), np.abs(np.fft.fft(out0)))


def test_ref_property_power_preserved():
    d, cfg = load_golden("ssfm_ref_power")
    out, _ = _run(cfg, d["Ei"], None)
    assert orc.signalPower(out) == pytest.approx(orc.signalPower(d["Ei"]), rel=1e-9)


def test_defaults_written_back_and_return_parameters():
    d, cfg = load_golden("mk_defaults")
    p = make_param(orc.parameters, cfg)
    p.returnParameters = True
    out, p2 = orc.manakovSSF(d["Ei"], p)
    assert p2 is p
    assert (p.hz, p.alpha, p.D, p.gamma, p.maxIter, p.tol) == (0.5, 0.2, 16, 1.3, 10, 1e-5)
    assert p.nlprMethod is True and p.maxNlinPhaseRot == 2e-2
    assert p.saveSpanN == [p.Ltotal // p.Lspan]


def test_edfa_gain_and_noise_power():
    p = orc.parameters()
    p.G, p.NF, p.Fc, p.Fs, p.seed = 16.0, 4.5, 193.1e12, 512e9, 3
    E = np.ones((1, 200000), dtype=complex)
    out = orc.edfa(E, p)
    G_lin, p_noise = orc.edfa_noise_power(16.0, 4.5, 193.1e12, 512e9)
    noise = out - E * np.sqrt(G_lin)
    assert np.mean(np.abs(noise) ** 2) == pytest.approx(p_noise, rel=2e-2)
    zero = np.zeros_like(E)
    assert np.array_equal(orc.edfa(E, p, noise=zero), E * np.sqrt(G_lin))


def test_dbp_roundtrip_recovers_input():
    d, cfg = load_golden("dbp_roundtrip")
    out, _ = _run(cfg, d["Ei"], {})
    err = np.linalg.norm(out - d["extra_orig"]) / np.linalg.norm(d["extra_orig"])
    assert err < 1e-6


@pytest.mark.parametrize("name", golden_names("mix_"))
def test_mixed_dtype_vectors_are_the_reference_on_the_cast_input(name):
    """tests/golden/mix_*.npz (the GPU twin's mixed-dtype calls): the oracle on the input cast to prec, result cast as the twin does."""
    d, cfg = load_golden(name)
    prec = np.dtype(cfg["prec"]).type
    tr = {}
    out = orc.manakovSSF(d["Ei"].astype(prec), make_param(orc.parameters, cfg), trace=tr)
    out = out.astype(d["Ei"].dtype) if not cfg["saveSpanN"] else out.astype(prec)
    assert out.dtype == d["out"].dtype and np.array_equal(out, d["out"]) and tr["iters"] == list(d["iters"])
