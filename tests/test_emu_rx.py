"""Receiver front-end on the CPU emulator: the kernel bodies of rx_kernels.h / ols_body and the stage
sequencing of rx_pipeline.h are the ones the GPU runs; the host marshalling is opticommpy_amd/rx.py
itself with its backend swapped.  Checked against the reference-generated vectors (tests/golden/rx_*)
and the receiver oracle."""
import numpy as np
import pytest

import emu_binding as eb
import opticommpy_amd as oa
from helpers import golden_names, load_golden, rel_l2, rx_call
from opticommpy_amd import rx as rxmod
from oracle import rx_oracle as orx
from oracle.ssf_oracle import parameters as oparams

RX = [n for n in golden_names("rx_") if n not in ("rx_lowpassfir", "rx_pd_noise_seed11")]
TOL = 1e-12                      # complex128 arithmetic in a different operation order (FFT sizes, fused subtraction)


@pytest.fixture(autouse=True)
def emu_backend(monkeypatch):
    monkeypatch.setattr(rxmod, "_backend", eb.EmuRxBackend())


@pytest.mark.parametrize("name", RX)
def test_rx_golden_vectors_on_emulated_kernels(name):
    d, cfg = load_golden(name)
    out = rx_call(oa, oa.parameters, d, cfg)
    ref = d["out"]
    assert out.dtype == ref.dtype and out.shape == ref.shape
    if cfg["func"] in ("decimate", "opticalHybrid2x4"):
        assert np.array_equal(out, ref)                                    # pure selection / host glue
    else:
        scale = np.max(np.abs(ref))
        assert np.max(np.abs(out - ref)) <= TOL * max(scale, 1e-300), np.max(np.abs(out - ref)) / scale


def test_lowpassfir_matches_reference_vectors():
    d, cfg = load_golden("rx_lowpassfir")
    assert np.array_equal(oa.lowPassFIR(*cfg["rect"], "rect"), d["rect"])
    assert np.array_equal(oa.lowPassFIR(*cfg["gauss"], "gauss"), d["gauss"])


def test_photodiode_with_supplied_noise_matches_seeded_reference_run():
    d, cfg = load_golden("rx_pd_noise_seed11")
    p = oa.parameters()
    for k, v in cfg.items():
        if k != "func":
            setattr(p, k, v)
    un = np.stack([d["extra_shot"], d["extra_thermal"]])[None]             # (1 photodiode, 2 kinds, N)
    out = oa.photodiode(d["Ei"].copy(), p, _unit_normals=un)
    assert np.max(np.abs(out - d["out"])) <= 1e-12 * np.max(np.abs(d["out"]))


def test_device_noise_statistics():
    """Philox noise of a photodiode: variance of shot + thermal terms as devices.py:381-390 defines them."""
    N = 1 << 14
    E = np.full(N, np.sqrt(1e-3), dtype=complex)
    p = oa.parameters()
    p.Fs, p.B, p.bandwidthLimitation, p.seed = 128e9, 30e9, False, 7
    i = oa.photodiode(E, p)
    q, kB = 1.602176634e-19, 1.380649e-23
    var = p.Fs * q * (1e-3 + 5e-9) + p.Fs * 2 * kB * 298.15 / 50
    assert np.mean(i) == pytest.approx(1e-3, rel=2e-3)
    assert np.var(i) == pytest.approx(var, rel=0.05)
    p.seed = 8
    assert not np.array_equal(i, oa.photodiode(E, p))                      # another seed, another stream
    p.seed = 7
    assert np.array_equal(i, oa.photodiode(E, p))                          # same seed, same stream


def test_pdm_receiver_against_oracle_on_a_longer_field_with_noise_arrays():
    """All stages at once (PBS, polarisation delay, PDL, hybrid, 8 noisy photodiodes, low-pass, IQ
    imbalance, skew) with host-supplied unit normals, against the oracle fed with the same normals."""
    rng = np.random.default_rng(5)
    N = 6000
    Es = (rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * 0.02
    Elo = np.sqrt(8e-3) * np.exp(1j * 2 * np.pi * 2e8 * np.arange(N) / 96e9)
    fe = dict(Fs=96e9, polRotation=-0.3, pdl=0.7, polDelay=-4e-12, ampImbX=0.3, phaseImbX=-0.1, timeSkewX=1e-12,
              ampImbY=0.1, phaseImbY=0.05, timeSkewY=0.0)
    pd = dict(Fs=96e9, B=25e9, N=101, currentSaturation=True, IpdSat=6e-3)
    un = rng.normal(size=(8, 2, N))

    def bag(cls, kw):
        o = cls()
        for k, v in kw.items():
            setattr(o, k, v)
        return o
    out = oa.pdmCoherentReceiver(Es, Elo, bag(oa.parameters, fe), bag(oa.parameters, pd), _unit_normals=un)
    # oracle noise layout: ((I pair: (PD1, PD2)), (Q pair: (PD1, PD2))) per polarisation, each PD = (shot, thermal)
    def pd_noise(slot):
        return un[slot][0], un[slot][1]

    def pol(b):
        return (pd_noise(b), pd_noise(b + 1)), (pd_noise(b + 2), pd_noise(b + 3))
    ref = orx.pdmCoherentReceiver(Es, Elo, bag(oparams, fe), bag(oparams, pd), noise=(pol(0), pol(4)))
    assert rel_l2(out, ref) <= 1e-12


def test_a_coherent_receiver_call_is_at_most_three_launches():
    """Round 5: every stage that is not a filter rides in the loads / stores of the filter next to it (rx_kernels.h:
    rx_ols_body).  All stages on = polarisation-delay filters (PBS in the loads), low-pass filter (detection in the loads), skew
    filters (IQ imbalance in the loads, I + jQ in the stores) = 3 launches (4 when the two polarisations' skews need different
    zero padding); the notebook's receiver (polarisation delay, ideal photodiodes) = 1 since round 6 (the detection rides in the delay
    filters' stores: POST_DET; 2 before); defaults (band-limited photodiodes) = 1."""
    e = eb.load()
    e.emu_rx_launches.restype = __import__("ctypes").c_long
    rng = np.random.default_rng(6)
    N = 3000
    Es = (rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * 0.02
    Elo = np.full(N, np.sqrt(8e-3), dtype=complex)

    def bag(kw):
        o = oa.parameters()
        for k, v in kw.items():
            setattr(o, k, v)
        return o
    for fe, pd, want in ((dict(polRotation=0.2, pdl=1.0, polDelay=2e-12, ampImbX=0.5, timeSkewX=1e-12, timeSkewY=-1e-12), dict(B=25e9, seed=1), 3),
                         (dict(polRotation=0.2, polDelay=2e-12, timeSkewX=1e-12), dict(B=25e9, seed=1), 4),
                         (dict(polRotation=np.pi / 3, polDelay=3 / 32e9), dict(B=32e9, ideal=True), 1),
                         (dict(), dict(B=30e9, seed=2), 1), (dict(), dict(B=30e9, ideal=True), 1)):
        out = oa.pdmCoherentReceiver(Es, Elo, bag(dict(Fs=96e9, **fe)), bag(dict(Fs=96e9, **pd)))
        assert out.shape == (N, 2) and e.emu_rx_launches() == want, (fe, e.emu_rx_launches())
    # photodiode / balancedPD: detection in the filter's loads, the real part in its stores (or one element-wise pass): 1 launch
    for pd in (dict(B=30e9, seed=2), dict(B=30e9, ideal=True), dict(B=30e9, bandwidthLimitation=False, seed=3)):
        i = oa.photodiode(Es, bag(dict(Fs=96e9, **pd)))
        assert i.shape == (N,) and i.dtype == np.float64 and e.emu_rx_launches() == 1, (pd, e.emu_rx_launches())
        i = oa.balancedPD(Es[:, 0], Es[:, 1], bag(dict(Fs=96e9, **pd)))
        assert i.shape == (N,) and e.emu_rx_launches() == 1, (pd, e.emu_rx_launches())


@pytest.mark.parametrize("N", [1, 2, 3, 17])
def test_tiny_signals(N):
    """Shorter than every filter involved (31-tap low-pass, 512-tap delay filters)."""
    rng = np.random.default_rng(N)
    Es = (rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * 0.02
    Elo = np.full(N, 0.05 + 0j)
    fe = dict(Fs=64e9, polDelay=3e-12, timeSkewX=2e-12)
    pd = dict(Fs=64e9, B=20e9, shotNoise=False, thermalNoise=False, N=31)

    def bag(cls, kw):
        o = cls()
        for k_, v in kw.items():
            setattr(o, k_, v)
        return o
    a = oa.pdmCoherentReceiver(Es, Elo, bag(oa.parameters, fe), bag(oa.parameters, pd))
    b = orx.pdmCoherentReceiver(Es, Elo, bag(oparams, fe), bag(oparams, pd))
    assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b))
    x = rng.normal(size=N) + 1j * rng.normal(size=N)
    for K in (1, 2, 5):
        h = rng.normal(size=K)
        assert np.max(np.abs(oa.firFilter(h, x) - orx.firFilter(h, x))) <= 1e-13


def test_realness_is_decided_on_values_like_the_reference():
    """blockwiseFFTConv returns a float array when no sample has an imaginary part, whatever the dtype
    (optic/dsp/core.py:1043-1046)."""
    x = np.random.default_rng(3).normal(size=300) + 0j
    a, b = oa.delaySignal(x, 0.3 / 64e9, 64e9), orx.delaySignal(x, 0.3 / 64e9, 64e9)
    assert a.dtype == b.dtype == np.float64 and np.max(np.abs(a - b)) <= 1e-13


def test_error_conventions():
    p = oa.parameters()
    with pytest.raises(AttributeError):
        oa.pdmCoherentReceiver(np.zeros((16, 2), complex), np.zeros(16, complex), p)          # no Fs
    p.Fs = 64e9
    with pytest.raises(AssertionError):
        oa.pdmCoherentReceiver(np.zeros((16, 2), complex), np.zeros(15, complex), p)
    pd = oa.parameters()
    pd.Fs, pd.B = 40e9, 30e9
    with pytest.raises(AssertionError):
        oa.photodiode(np.zeros(16, complex), pd)                                              # Fs < 2 B
    pd = oa.parameters()
    pd.R, pd.ideal = -1, True
    with pytest.raises(AssertionError):
        oa.photodiode(np.zeros(16, complex), pd)
    dp = oa.parameters()
    dp.SpSin, dp.SpSout = 16, 2
    with pytest.raises(ValueError):
        oa.decimate(np.zeros(100), dp)                                                        # 100 % 16 != 0


@pytest.mark.parametrize("lg", [4, 6, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("ncols", [1, 2, 3, 4])
def test_every_overlap_save_instantiation(lg, ncols):
    """One kernel per transform size (256 ... 8192 points) and per columns-side-by-side (two for an even column count up to 4096
    points), the run-time plan below 256 points, the filter in register order (fused_kernels.h: ols_body_x, ols_permute_filter):
    all against np.convolve -- blockwiseFFTConv's definition (optic/dsp/core.py:973-1046) with a random impulse response."""
    import ctypes as C
    emu = eb.load()
    rng = np.random.default_rng(100 * lg + ncols)
    nfft = 1 << lg
    K = max(2, nfft // 5) | 1
    N = 3 * nfft + 37
    x = np.ascontiguousarray(rng.normal(size=(N, ncols)) + 1j * rng.normal(size=(N, ncols)))
    h = rng.normal(size=K) + 1j * rng.normal(size=K)
    H = np.ascontiguousarray(np.fft.fft(np.pad(h, (0, nfft - K))))
    out = np.empty_like(x)
    assert emu.emu_overlap_save(N, ncols, lg, K, H.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                                out.ctypes.data_as(C.c_void_p)) == 0
    D = (K - 1) // 2
    want = np.stack([np.convolve(x[:, c], h)[D:D + N] for c in range(ncols)], axis=1)
    assert rel_l2(out, want) <= 1e-13


@pytest.mark.parametrize("ntaps", [2049, 4096])
def test_fir_filter_with_more_than_2048_taps_uses_8192_point_blocks(ntaps):
    """(4096-point blocks would advance by 4096 - ntaps + 1 samples: one sample per block at 4096 taps)"""
    rng = np.random.default_rng(ntaps)
    N = 20000
    x = rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))
    h = rng.normal(size=ntaps)
    y = oa.firFilter(h, x)                                   # (the autouse fixture routes the kernels to the emulator)
    want = np.stack([np.convolve(x[:, c], h, mode="same") for c in range(2)], axis=1)
    assert rel_l2(y, want) <= 1e-13


def test_block_size_rule_is_the_same_on_both_sides_of_the_abi():
    """edc picks its overlap-save block in Python (models._ols_block), firFilter / the photodiodes' low-pass in C++ (rx_pipeline.h:
    fir_nfft): one rule -- 2048 points up to 682 taps, 4096 up to 2048, 8192 above, never smaller than the filter."""
    from opticommpy_amd import models
    e = eb.load()
    for K in list(range(1, 700)) + [1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096]:
        n = e.emu_fir_nfft(K)
        assert n == models._ols_block(K) and n >= K and n & (n - 1) == 0, K
    assert (e.emu_fir_nfft(255), e.emu_fir_nfft(682), e.emu_fir_nfft(683), e.emu_fir_nfft(2049)) == (2048, 2048, 4096, 8192)
