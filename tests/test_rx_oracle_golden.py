"""Pin the receiver-side CPU oracle (oracle/rx_oracle.py) bit-for-bit against golden vectors produced
by importing the reference (tools/gen_golden.py rx -> tests/golden/rx_*.npz)."""
import json

import numpy as np
import pytest

from helpers import golden_names, load_golden, rx_call
from oracle import rx_oracle as rx
from oracle.ssf_oracle import parameters

RX = [n for n in golden_names("rx_") if n not in ("rx_lowpassfir", "rx_pd_noise_seed11")]


def test_rx_fixture_inventory():
    assert len(RX) >= 20
    for prefix in ("rx_fir_", "rx_decimate_", "rx_delay_", "rx_iqmix_", "rx_pd_", "rx_pdm_"):
        assert any(n.startswith(prefix) for n in RX), prefix


@pytest.mark.parametrize("name", RX)
def test_rx_oracle_matches_reference_bit_for_bit(name):
    d, cfg = load_golden(name)
    out = rx_call(rx, parameters, d, cfg)
    assert out.dtype == d["out"].dtype and out.shape == d["out"].shape
    assert np.array_equal(out, d["out"]), f"max abs diff {np.max(np.abs(out - d['out']))}"


def test_lowpassfir_taps():
    d, cfg = load_golden("rx_lowpassfir")
    assert np.array_equal(rx.lowPassFIR(*cfg["rect"], "rect"), d["rect"])
    assert np.array_equal(rx.lowPassFIR(*cfg["gauss"], "gauss"), d["gauss"])


def test_photodiode_noise_draw_order():
    """Seeded reference run == oracle fed with the same unit normals (shot first, then thermal)."""
    d, cfg = load_golden("rx_pd_noise_seed11")
    p = parameters()
    for k, v in cfg.items():
        if k != "func":
            setattr(p, k, v)
    out = rx.photodiode(d["Ei"].copy(), p, noise=(d["extra_shot"], d["extra_thermal"]))
    assert np.array_equal(out, d["out"])
    out2 = rx.photodiode(d["Ei"].copy(), p)                       # same through np.random.seed
    assert np.array_equal(out2, d["out"])
