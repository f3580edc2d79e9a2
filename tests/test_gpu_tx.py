"""GPU parity tests of the WDM transmitter (run with -m gpu): opticommpy_amd.simpleWDMTx through the C ABI
against the reference-generated vectors (tests/golden/tx_wdm_*) and the transmitter oracle at the
reference's default size; device-resident hand-over to the channel."""
import time

import numpy as np
import pytest

import opticommpy_amd as oa
from helpers import golden_names, load_golden, make_param
from oracle import tx_oracle as otx
from oracle.ssf_oracle import parameters as oparams

pytestmark = pytest.mark.gpu

WDM = golden_names("tx_wdm_")


@pytest.mark.parametrize("name", WDM)
def test_simple_wdm_tx_golden_vectors(name):
    d, cfg = load_golden(name)
    sig, symb, par = oa.simpleWDMTx(make_param(oa.parameters, cfg))
    assert np.array_equal(symb, d["symb"]) and np.array_equal(par.wdmFreqGrid, d["freqGrid"]) and np.array_equal(par.pmf, d["pmf"])
    assert sig.shape == d["out"].shape and sig.dtype == d["out"].dtype
    assert np.max(np.abs(sig - d["out"])) <= 1e-12 * np.max(np.abs(d["out"]))


def test_reference_default_size_vs_oracle_and_device_hand_over(capsys):
    """The reference's defaults (16-QAM, 60000 bits, 16 SpS, 5 channels, 1024 taps) with two polarisations
    and a 100 kHz laser: N = 240 000 samples per polarisation."""
    kw = dict(seed=42, nPolModes=2, laserLinewidth=100e3, prgsBar=False)
    t0 = time.perf_counter()
    ref, rsymb, _ = otx.simpleWDMTx(make_param(oparams, kw))
    t_cpu = time.perf_counter() - t0
    oa.simpleWDMTx(make_param(oa.parameters, kw))
    t0 = time.perf_counter()
    sig, symb, par = oa.simpleWDMTx(make_param(oa.parameters, kw))
    t_gpu = time.perf_counter() - t0
    assert np.array_equal(symb, rsymb)
    assert np.max(np.abs(sig - ref)) <= 1e-12 * np.max(np.abs(ref))
    sd, _, _ = oa.simpleWDMTx(make_param(oa.parameters, kw), device_output=True)
    assert isinstance(sd, oa.DeviceArray) and np.array_equal(sd.get(), sig)
    ch = make_param(oa.parameters, dict(Fs=par.Rs * par.SpS, Ltotal=2, Lspan=2, hz=0.5, amp="ideal", nlprMethod=False,
                                        prgsBar=False, saveSpanN=[]))
    a = oa.manakovSSF(sig, ch)
    b = oa.manakovSSF(sd, ch)                                            # straight from the transmitter, no host copy
    assert np.array_equal(a, b.get())
    with capsys.disabled():
        print(f"\n[tx] simpleWDMTx defaults x 2 pol (10 channel-modes, N=240000): oracle {t_cpu*1e3:.0f} ms, GPU {t_gpu*1e3:.0f} ms "
              f"(of which host symbol / phase-noise draws dominate)")
