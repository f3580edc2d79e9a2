"""Round 6: device forms of edfa / linearFiberChannel / pbs / opticalHybrid2x4 (reference optic/models/devices.py:671-726, 223-260,
462-500; optic/models/channels.py:30-109) -- their kernels on the CPU emulator here, the product on the GPU under -m gpu."""
import numpy as np
import pytest

import emu_binding as eb
import opticommpy_amd as oa
from opticommpy_amd import rx as rxmod
from helpers import rel_l2, synth_field
from oracle import rx_oracle as orx
from oracle import ssf_oracle as orc


def bag(cls=None, **kw):
    p = (cls or oa.parameters)()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


# ------------------------------------------------------------------------------------------ emulator (no GPU)
@pytest.fixture
def emu_rx(monkeypatch):
    monkeypatch.setattr(rxmod, "_backend", eb.EmuRxBackend())


@pytest.mark.parametrize("theta", [0.0, 0.3, -1.2])
def test_pbs_kernel_on_the_emulator_against_the_oracle(emu_rx, theta):
    E = synth_field(300, 2, 5, 0.0)
    for inp in (E, E[:, 0].copy()):
        ex, ey = oa.pbs(inp, theta)
        rx_, ry_ = orx.pbs(inp.copy(), theta)
        assert ex.shape == (300,) and ex.dtype == np.complex128
        assert np.max(np.abs(ex - rx_)) <= 1e-15 * np.max(np.abs(E)) and np.max(np.abs(ey - ry_)) <= 1e-15 * np.max(np.abs(E))


def test_optical_hybrid_kernel_on_the_emulator_is_the_reference_matrix_product(emu_rx):
    Es, Elo = synth_field(257, 1, 6, 0.0)[:, 0], synth_field(257, 1, 7, 10.0)[:, 0]
    out = oa.opticalHybrid2x4(Es, Elo)
    assert out.shape == (4, 257) and np.array_equal(out, orx.opticalHybrid2x4(Es, Elo))


def test_edfa_kernel_on_the_emulator():
    E = synth_field(4096, 2, 8, 0.0)
    G_lin, p_noise = orc.edfa_noise_power(20, 4.5, 193.1e12, 64e9)
    nz = orc.gaussianComplexNoise(E.shape, p_noise, 3)
    out = eb.edfa(E, G_lin, p_noise, noise=nz)
    ref = E * np.sqrt(G_lin) + nz                                           # devices.py:724-726
    assert np.max(np.abs(out - ref)) <= 4e-16 * np.max(np.abs(ref))
    assert np.array_equal(eb.edfa(E, G_lin, p_noise), E * np.sqrt(G_lin))   # neither noise array nor seed: the gain alone
    dev = eb.edfa(E, G_lin, p_noise, seed=99) - E * np.sqrt(G_lin)           # Philox on the "device": CN(0, p_noise), columns independent
    assert np.mean(np.abs(dev) ** 2) == pytest.approx(p_noise, rel=0.05)
    assert abs(np.mean(dev[:, 0] * np.conj(dev[:, 1]))) <= 0.05 * p_noise
    assert abs(np.mean(dev)) <= 0.05 * np.sqrt(p_noise)
    assert np.array_equal(eb.edfa(E, G_lin, p_noise, seed=99), eb.edfa(E, G_lin, p_noise, seed=99))
    # rows of the stream: column c of a call with row0 = r draws what column c + r of a call with row0 = 0 draws
    a = eb.edfa(np.zeros((64, 3), complex), G_lin, p_noise, seed=5)
    b = eb.edfa(np.zeros((64, 2), complex), G_lin, p_noise, seed=5, row0=1)
    assert np.array_equal(a[:, 1:], b)


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_edfa_numpy_in_is_the_reference_draw_for_draw_and_device_in_stays_in_hbm():
    from opticommpy_amd import device
    E = synth_field(1 << 14, 2, 11, 0.0)
    p = bag(Fs=64e9, G=17, NF=5, seed=42)
    out = oa.edfa(E, p)
    ref = orc.edfa(E, bag(orc.parameters, Fs=64e9, G=17, NF=5, seed=42))
    assert out.dtype == np.complex128 and np.max(np.abs(out - ref)) <= 4e-16 * np.max(np.abs(ref))
    before = device.transfer_counts()
    Ed = oa.to_device(E)
    mid = device.transfer_counts()
    od = oa.edfa(Ed, p)
    assert isinstance(od, oa.DeviceArray) and device.transfer_counts() == mid and mid["h2d"] == before["h2d"] + 1
    dev = od.get() - E * np.sqrt(10 ** 1.7)
    _, p_noise = orc.edfa_noise_power(17, 5, 193.1e12, 64e9)
    assert np.mean(np.abs(dev) ** 2) == pytest.approx(p_noise, rel=0.03)
    assert np.array_equal(oa.edfa(Ed, p).get(), od.get())                   # seeded: the same stream
    assert not np.array_equal(oa.edfa(Ed, bag(Fs=64e9, G=17, NF=5)).get(), od.get())
    with pytest.raises(AssertionError):
        oa.edfa(E, bag(Fs=64e9, G=17, NF=2))


@pytest.mark.gpu
def test_pbs_and_hybrid_on_device_arrays_without_transfers():
    from opticommpy_amd import device
    E = synth_field(1 << 15, 2, 12, 3.0)
    lo = synth_field(1 << 15, 1, 13, 10.0)[:, 0]
    Ed, Ld = oa.to_device(E), oa.to_device(lo)
    c0 = device.transfer_counts()
    ex, ey = oa.pbs(Ed, 0.4)
    h = oa.opticalHybrid2x4(ex, Ld)
    assert device.transfer_counts() == c0 and isinstance(ex, oa.DeviceArray) and h.shape == (4, 1 << 15)
    rx_, ry_ = orx.pbs(E.copy(), 0.4)
    assert np.max(np.abs(ex.get() - rx_)) <= 1e-15 * np.max(np.abs(E)) and np.max(np.abs(ey.get() - ry_)) <= 1e-15 * np.max(np.abs(E))
    assert np.max(np.abs(h.get() - orx.opticalHybrid2x4(rx_, lo))) <= 1e-15 * np.max(np.abs(lo))
    nx, ny = oa.pbs(E, 0.4)                                                  # numpy in, numpy out: the same kernel
    assert np.array_equal(nx, ex.get()) and np.array_equal(ny, ey.get())
    x1, y1 = oa.pbs(E[:, 0].copy())                                          # (N,): the x polarisation
    assert np.array_equal(x1, E[:, 0]) and not np.any(y1)


@pytest.mark.gpu
def test_linear_fiber_channel_on_device_arrays_and_in_the_reference_layout():
    from opticommpy_amd import device
    E = synth_field(1 << 16, 2, 14, 0.0)
    lp = dict(Fs=512e9, L=40, alpha=0.2, D=17, Fc=193.1e12)
    ref = orc.linearFiberChannel(E, bag(orc.parameters, **lp))
    out = oa.linearFiberChannel(E, bag(**lp))
    assert out.shape == E.shape and rel_l2(out, ref) <= 1e-12
    Ed = oa.to_device(E)
    c0 = device.transfer_counts()
    od = oa.linearFiberChannel(Ed, bag(**lp))
    assert isinstance(od, oa.DeviceArray) and device.transfer_counts() == c0
    assert np.array_equal(od.get(), out)
    o1 = oa.linearFiberChannel(E[:, 0].copy(), bag(**lp))                   # 1-D in, 1-D out
    assert o1.shape == (1 << 16,) and rel_l2(o1, ref[:, 0]) <= 1e-12
    o64 = oa.linearFiberChannel(E.astype(np.complex64), bag(**lp))          # the reference's result is complex128 too (channels.py:97)
    r64 = orc.linearFiberChannel(E.astype(np.complex64), bag(orc.parameters, **lp))
    assert o64.dtype == r64.dtype == np.complex128 and rel_l2(o64, r64) <= 5e-6
    back, prm = oa.linearFiberChannel(od, bag(returnParameters=True, **dict(lp, D=-17, alpha=-0.2)))    # and a chain stays in HBM
    assert prm.returnParameters and rel_l2(back.get(), E) <= 1e-12
