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


def test_laser_phase_noise_is_generated_on_the_device_when_there_is_no_seed(monkeypatch):
    """Without a seed nothing can be reproduced, so the laser's random walk (optic/dsp/core.py:792-826: phi[0] = 0, steps
    N(0, 2 pi lw Ts)) is generated on the device (Philox, ssf_tx_params.pn_seed): no host draw, no N-sample upload.  One channel,
    one polarisation, no modulation depth (mzmScale -> 0, a CW carrier): the transmitted field's phase IS the walk, so its
    increments can be measured: zero mean, the right variance, uncorrelated, a different walk per key and per channel."""
    lw, Fs = 5e6, 16 * 32e9
    kw = dict(M=4, nBits=2 * 4096, SpS=16, nChannels=1, nPolModes=1, laserLinewidth=lw, mzmScale=1e-9, pulseType="nrz", prgsBar=False)
    seen = []
    monkeypatch.setattr(wdm_tx, "phaseNoise", lambda *a, **k: (_ for _ in ()).throw(AssertionError("host phase-noise draw")))
    for key in (11, 12):
        from opticommpy_amd import models
        monkeypatch.setattr(models, "_device_seed", lambda seed, key=key: key)
        sig, _, _ = oa.simpleWDMTx(make_param(oa.parameters, kw))
        ph = np.unwrap(np.angle(sig[:, 0]))
        ph -= ph[0]
        inc = np.diff(ph)
        s2 = 2 * np.pi * lw / Fs
        assert abs(np.var(inc) / s2 - 1) < 0.02 and abs(np.mean(inc)) < 4 * np.sqrt(s2 / len(inc))
        assert abs(np.corrcoef(inc[:-1], inc[1:])[0, 1]) < 0.02
        assert np.max(np.abs(inc)) < 6 * np.sqrt(s2)          # ... also across the 4096-sample chunks the kernel works in: they join up
        seen.append(inc)
    assert np.max(np.abs(seen[0] - seen[1])) > 1e-6
    # with a seed: the reference's own draws, on the host, as before
    monkeypatch.undo()
    monkeypatch.setattr(wdm_tx, "_backend", eb.EmuTxBackend())
    a, _, _ = oa.simpleWDMTx(make_param(oa.parameters, dict(kw, seed=5)))
    b, _, _ = oa.simpleWDMTx(make_param(oa.parameters, dict(kw, seed=5)))
    assert np.array_equal(a, b)


@pytest.mark.parametrize("nPol,lw,nCh", [(1, 0.0, 3), (1, 2e5, 3), (2, 2e5, 2), (2, 0.0, 2), (1, 2e5, 1)])
def test_seeded_phase_walk_is_drawn_once_and_the_global_stream_ends_where_the_reference_leaves_it(nPol, lw, nCh):
    """tx.py:199 reseeds np.random with the same param.seed before every channel's walk: all channels share one walk, which is drawn
    once and uploaded as one row (ssf_tx_params.phi_rows = 1).  Field, symbols AND np.random's state after the call are the
    oracle's (which draws per channel, as the reference does)."""
    from oracle import tx_oracle
    kw = dict(M=16, nBits=4 * 512, SpS=8, nChannels=nCh, nPolModes=nPol, laserLinewidth=lw, seed=21, nFilterTaps=64, prgsBar=False)
    ref, symb_ref, _ = tx_oracle.simpleWDMTx(make_param(oa.parameters, kw))
    want = np.random.random_sample(4)                      # what a caller drawing from the global stream sees next
    calls = []
    real = wdm_tx.phaseNoise
    wdm_tx.phaseNoise = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        sig, symb, _ = oa.simpleWDMTx(make_param(oa.parameters, kw))
    finally:
        wdm_tx.phaseNoise = real
    got = np.random.random_sample(4)
    assert np.array_equal(symb, symb_ref)
    assert np.max(np.abs(sig - ref)) <= 1e-12 * np.max(np.abs(ref))
    assert np.array_equal(got, want)
    assert len(calls) == (1 if (lw or nPol == 1) else 0)
