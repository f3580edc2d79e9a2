"""WDM transmitter on the CPU emulator: host glue of opticommpy_amd/wdm_tx.py (constellations, symbol draws,
pulse taps, phase noise) against the reference-generated vectors exactly, and the device signal path
(zero-stuffed overlap-save FIR, peak / power normalisation, IQ modulator, frequency shift, accumulation:
rx_kernels.h bodies sequenced by rx_pipeline.h) against them within 1e-12 of the largest sample."""
import numpy as np
import pytest

import emu_binding as eb
import opticommpy_amd as oa
from helpers import golden_names, load_golden, make_param
from opticommpy_amd import wdm_tx

WDM = golden_names("tx_wdm_")


@pytest.fixture(autouse=True)
def emu_backend(monkeypatch):
    monkeypatch.setattr(wdm_tx, "_backend", eb.EmuTxBackend())


def test_host_glue_matches_reference_tables_exactly():
    d, _ = load_golden("tx_gray_maps")
    for name, (M, t) in dict(qam4=(4, "qam"), qam16=(16, "qam"), qam64=(64, "qam"), psk8=(8, "psk"), pam4=(4, "pam")).items():
        out = oa.grayMapping(M, t)
        assert out.dtype == d[name].dtype and np.array_equal(out, d[name]), name
    d, _ = load_golden("tx_pulses")
    for name, kw in (("rrc", dict(pulseType="rrc", SpS=16, nFilterTaps=1024, rollOff=0.01)),
                     ("rrc_odd", dict(pulseType="rrc", SpS=8, nFilterTaps=513, rollOff=0.25)),
                     ("rc", dict(pulseType="rc", SpS=4, nFilterTaps=64, rollOff=0.5)),
                     ("nrz", dict(pulseType="nrz", SpS=16)), ("rect", dict(pulseType="rect", SpS=8))):
        assert np.array_equal(oa.pulseShape(make_param(oa.parameters, kw)), d[name]), name
    d, cfg = load_golden("tx_phase_noise")
    assert np.array_equal(oa.phaseNoise(cfg["lw"], cfg["N"], cfg["Ts"], seed=cfg["seed"]), d["out"])


@pytest.mark.parametrize("name", WDM)
def test_simple_wdm_tx_on_emulated_kernels(name):
    d, cfg = load_golden(name)
    sig, symb, par = oa.simpleWDMTx(make_param(oa.parameters, cfg))
    assert np.array_equal(symb, d["symb"])                                           # same draws, same constellation
    assert np.array_equal(par.wdmFreqGrid, d["freqGrid"]) and np.array_equal(par.pmf, d["pmf"])
    assert sig.shape == d["out"].shape and sig.dtype == d["out"].dtype
    assert np.max(np.abs(sig - d["out"])) <= 1e-12 * np.max(np.abs(d["out"]))


def test_errors():
    with pytest.raises(ValueError):
        oa.simpleWDMTx(make_param(oa.parameters, dict(probDist="gaussian", prgsBar=False)))
    with pytest.raises(AssertionError):
        oa.simpleWDMTx(make_param(oa.parameters, dict(powerPerChannel=[0, 0], nChannels=3, nBits=1024, prgsBar=False)))
