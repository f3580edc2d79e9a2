"""The notebook chain end to end, pinned to the REFERENCE.

tests/golden/chain_wdm_240k.npz (tools/gen_golden.py chain) holds the reference's own functions run in the order of
examples/test_WDM_transmission.ipynb (cells 10, 14, 18, 20, 22, 23): simpleWDMTx(seed) -> manakovSSF (two 50 km spans,
adaptive step, amp='ideal') -> basicLaserModel (local oscillator) -> pdmCoherentReceiver -> firFilter (matched filter) ->
decimate -> edc, each stage fed with the previous stage's output; 240 000 samples (simpleWDMTx's defaults: 2^6 * 3 * 5^4,
the mixed-radix pipeline).  CPU: the three oracles chained reproduce it bit for bit.  GPU: the package's functions chained,
through host arrays and through DeviceArrays (nothing leaves HBM between the transmitter and the final .get()), to 1e-9
at every stage with the reference's iteration list."""
import json

import numpy as np
import pytest

from helpers import load_golden, make_param, rel_l2
from oracle import rx_oracle as orx
from oracle import ssf_oracle as orc
from oracle import tx_oracle as otx


def projection(out, seed=4242):
    rng = np.random.default_rng(seed)
    r = (rng.normal(size=out.shape[0]) + 1j * rng.normal(size=out.shape[0])) / np.sqrt(2)
    return np.asarray(out).astype(np.complex128).T @ r


def run_chain(mod, P, cfg, channel, device=False):
    """The notebook's calls on module `mod` (the oracles or opticommpy_amd); returns every stage's output."""
    o = {}
    if device:
        o["tx"], o["symb"], ptx = mod["simpleWDMTx"](make_param(P, cfg["tx"]), device_output=True)
    else:
        o["tx"], o["symb"], ptx = mod["simpleWDMTx"](make_param(P, cfg["tx"]))
    assert np.array_equal(ptx.wdmFreqGrid[cfg["chIndex"]], cfg["lo"]["freqShift"] + 128e6)
    o["ch"] = channel(o["tx"], make_param(P, cfg["ch"]))
    o["lo"] = mod["basicLaserModel"](make_param(P, cfg["lo"]))
    o["rx"] = mod["pdmCoherentReceiver"](o["ch"], o["lo"], make_param(P, cfg["fe"]), make_param(P, cfg["pd"]))
    pulse = mod["pulseShape"](make_param(P, cfg["ps"]))
    o["mf"] = mod["firFilter"](pulse, o["rx"])
    o["dec"] = mod["decimate"](o["mf"], make_param(P, cfg["dec"]))
    o["out"] = mod["edc"](o["dec"], make_param(P, cfg["edc"]))
    return o


def test_oracles_chained_reproduce_the_reference_chain_bit_for_bit():
    d, cfg = load_golden("chain_wdm_240k")
    mod = dict(simpleWDMTx=otx.simpleWDMTx, basicLaserModel=otx.basicLaserModel, pdmCoherentReceiver=orx.pdmCoherentReceiver,
               pulseShape=otx.pulseShape, firFilter=orx.firFilter, decimate=orx.decimate, edc=orc.edc)
    tr = {}
    o = run_chain(mod, orc.parameters, cfg, lambda E, p: orc.manakovSSF(E, p, trace=tr))
    assert tr["iters"] == list(d["iters"])
    dd = int(cfg["d"])
    assert np.array_equal(o["symb"][:64], d["symb_head"])
    for k in ("tx", "ch", "lo", "rx", "mf"):
        assert np.array_equal(o[k][::dd], d[k + "_dec"]), k
    assert np.array_equal(o["out"], d["out"])


def check_against_reference(o, d, cfg, tol=1e-9):
    dd = int(cfg["d"])
    get = lambda x: x.get() if hasattr(x, "get") else np.asarray(x)
    assert np.array_equal(o["symb"][:64], d["symb_head"])
    assert np.max(np.abs(projection(o["symb"].reshape(len(o["symb"]), -1)) - d["symb_proj"])) == 0
    worst = {}
    for k in ("tx", "ch", "rx", "mf"):
        a = get(o[k])
        worst[k] = rel_l2(a[::dd], d[k + "_dec"])
        assert worst[k] <= tol, (k, worst[k])
        scale = np.sqrt(np.sum(np.abs(a) ** 2))
        assert np.max(np.abs(projection(a) - d[k + "_proj"])) <= tol * scale, k
    out = get(o["out"])
    worst["out"] = rel_l2(out, d["out"])
    assert out.shape == d["out"].shape and worst["out"] <= tol, worst
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize("device", [False, True], ids=["host_arrays", "device_arrays"])
def test_notebook_chain_against_the_reference(device):
    import opticommpy_amd as oa
    from opticommpy_amd import device as odev
    from opticommpy_amd import models
    d, cfg = load_golden("chain_wdm_240k")
    mod = {k: getattr(oa, k) for k in ("simpleWDMTx", "basicLaserModel", "pdmCoherentReceiver", "pulseShape", "firFilter", "decimate", "edc")}
    info = {}

    def channel(E, p):
        out = oa.manakovSSF(E, p, _trace=True)
        info.update(models.last_run)
        return out

    d2h0 = odev.transfer_counts()["d2h"]
    o = run_chain(mod, oa.parameters, cfg, channel, device=device)
    if device:
        assert all(isinstance(o[k], oa.DeviceArray) for k in ("tx", "ch", "rx", "mf", "dec", "out"))
        assert odev.transfer_counts()["d2h"] == d2h0, "a DeviceArray went through the host between the transmitter and the final get()"
    assert list(info["iters"]) == list(d["iters"])                   # the reference's adaptive-step iteration list
    flat = np.concatenate([np.asarray(r, dtype=float) for r in info["lims"]])
    np.testing.assert_allclose(flat, d["lims"], rtol=1e-6, atol=1e-15)
    worst = check_against_reference(o, d, cfg)
    print("chain vs reference:", {k: "%.1e" % v for k, v in worst.items()})
