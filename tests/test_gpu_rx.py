"""GPU parity tests of the receiver side (run with -m gpu on an MI355X): opticommpy_amd's firFilter /
delaySignal / decimate / iqMixing / photodiode / balancedPD / coherentReceiver / pdmCoherentReceiver,
called through the C ABI, against the reference-generated vectors (tests/golden/rx_*) and against the
receiver oracle on seeded inputs at sizes the oracle finishes in seconds.

Tolerance: complex128 results within 1e-12 of the reference relative to the largest output value (the
operation order differs: other FFT sizes, balanced subtraction before the common filter); decimate is a
selection and must be exact."""
import time

import numpy as np
import pytest

import opticommpy_amd as oa
from helpers import golden_names, load_golden, rel_l2, rx_call
from oracle import rx_oracle as orx
from oracle.ssf_oracle import parameters as oparams

pytestmark = pytest.mark.gpu

RX = [n for n in golden_names("rx_") if n not in ("rx_lowpassfir", "rx_pd_noise_seed11")]
TOL = 1e-12


def bag(cls, kw):
    o = cls()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


@pytest.mark.parametrize("name", RX)
def test_rx_golden_vectors(name):
    d, cfg = load_golden(name)
    out = rx_call(oa, oa.parameters, d, cfg)
    ref = d["out"]
    assert out.dtype == ref.dtype and out.shape == ref.shape
    if cfg["func"] in ("decimate", "opticalHybrid2x4"):
        assert np.array_equal(out, ref)
    else:
        assert np.max(np.abs(out - ref)) <= TOL * np.max(np.abs(ref))


def test_photodiode_with_supplied_noise_matches_seeded_reference_run():
    d, cfg = load_golden("rx_pd_noise_seed11")
    p = bag(oa.parameters, {k: v for k, v in cfg.items() if k != "func"})
    un = np.stack([d["extra_shot"], d["extra_thermal"]])[None]
    out = oa.photodiode(d["Ei"].copy(), p, _unit_normals=un)
    assert np.max(np.abs(out - d["out"])) <= 1e-12 * np.max(np.abs(d["out"]))


def test_device_noise_statistics_and_streams():
    N = 1 << 18
    E = np.full(N, np.sqrt(1e-3), dtype=complex)
    p = bag(oa.parameters, dict(Fs=128e9, B=30e9, bandwidthLimitation=False, seed=7))
    i = oa.photodiode(E, p)
    q, kB = 1.602176634e-19, 1.380649e-23
    var = p.Fs * q * (1e-3 + 5e-9) + p.Fs * 2 * kB * 298.15 / 50
    assert np.mean(i) == pytest.approx(1e-3, rel=5e-4)
    assert np.var(i) == pytest.approx(var, rel=0.02)
    assert np.array_equal(i, oa.photodiode(E, p))
    p.seed = 8
    assert not np.array_equal(i, oa.photodiode(E, p))
    # the eight photodiodes of a PDM receiver draw from eight different streams
    Es = np.zeros((4096, 2), complex)
    Elo = np.full(4096, np.sqrt(1e-3), dtype=complex)
    s = oa.pdmCoherentReceiver(Es, Elo, bag(oa.parameters, dict(Fs=128e9)),
                               bag(oa.parameters, dict(Fs=128e9, bandwidthLimitation=False, seed=3)))
    cols = np.stack([s[:, 0].real, s[:, 0].imag, s[:, 1].real, s[:, 1].imag])
    c = np.corrcoef(cols)
    assert np.max(np.abs(c - np.eye(4))) < 0.08


@pytest.mark.parametrize("N", [1 << 16, 1 << 20])
def test_pdm_receiver_vs_oracle_all_stages(N):
    rng = np.random.default_rng(5)
    Es = (rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * 0.02
    Elo = np.sqrt(8e-3) * np.exp(1j * 2 * np.pi * 2e8 * np.arange(N) / 96e9)
    fe = dict(Fs=96e9, polRotation=-0.3, pdl=0.7, polDelay=-4e-12, ampImbX=0.3, phaseImbX=-0.1, timeSkewX=1e-12,
              ampImbY=0.1, phaseImbY=0.05, timeSkewY=0.0)
    pd = dict(Fs=96e9, B=25e9, N=101, currentSaturation=True, IpdSat=6e-3)
    un = rng.normal(size=(8, 2, N))
    out = oa.pdmCoherentReceiver(Es, Elo, bag(oa.parameters, fe), bag(oa.parameters, pd), _unit_normals=un)

    def pd_noise(slot):
        return un[slot][0], un[slot][1]

    def pol(b):
        return (pd_noise(b), pd_noise(b + 1)), (pd_noise(b + 2), pd_noise(b + 3))
    ref = orx.pdmCoherentReceiver(Es, Elo, bag(oparams, fe), bag(oparams, pd), noise=(pol(0), pol(4)))
    assert rel_l2(out, ref) <= 1e-12


def test_fir_filter_sizes_and_limits():
    rng = np.random.default_rng(9)
    x = (rng.normal(size=(1 << 20, 2)) + 1j * rng.normal(size=(1 << 20, 2)))
    for K in (1, 2, 255, 1024, 4096):
        h = rng.normal(size=K) / max(K, 1) ** 0.5
        out = oa.firFilter(h, x)
        ref = orx.firFilter(h, x[: 1 << 15])
        # the first 2^15 - K outputs do not depend on what follows
        assert np.max(np.abs(out[: (1 << 15) - K] - ref[: (1 << 15) - K])) <= 1e-12 * np.max(np.abs(ref)), K
    # more than 4096 taps (the reference takes any length): segment by segment on the device (ssf_fir_long)
    h = rng.normal(size=4097) / 64
    out = oa.firFilter(h, x[:8192])
    ref = orx.firFilter(h, x[:8192])
    assert np.max(np.abs(out - ref)) <= 1e-12 * np.max(np.abs(ref))
    xr = rng.normal(size=5000).astype(np.float32)
    yr = oa.firFilter(np.ones(5) / 5, xr)
    assert yr.dtype == np.float32 and yr.shape == xr.shape
    np.testing.assert_allclose(yr, orx.firFilter(np.ones(5) / 5, xr), atol=1e-6)


@pytest.mark.parametrize("lg", [4, 6, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("ncols", [1, 2, 3, 4])
def test_every_overlap_save_instantiation_through_the_c_abi(lg, ncols):
    """ssf_overlap_save (blockwiseFFTConv with a caller-supplied response, optic/dsp/core.py:973-1046) at every transform size the
    kernel is instantiated for (256 ... 8192 points; the run-time plan below), one and two columns per transform, the filter in
    register order: against np.convolve with a random impulse response.  (The same sweep on the emulator: tests/test_emu_rx.py.)"""
    import ctypes as C
    from opticommpy_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(100 * lg + ncols)
    nfft = 1 << lg
    K = max(2, nfft // 5) | 1
    N = 3 * nfft + 37
    x = np.ascontiguousarray(rng.normal(size=(N, ncols)) + 1j * rng.normal(size=(N, ncols)))
    h = rng.normal(size=K) + 1j * rng.normal(size=K)
    H = np.ascontiguousarray(np.fft.fft(np.pad(h, (0, nfft - K))))
    out = np.empty_like(x)
    rc = lib.ssf_overlap_save(0, N, ncols, _lib.SSF_C128, nfft, K, H.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                              out.ctypes.data_as(C.c_void_p))
    _lib.raise_for(lib, None, rc)
    D = (K - 1) // 2
    want = np.stack([np.convolve(x[:, c], h)[D:D + N] for c in range(ncols)], axis=1)
    assert rel_l2(out, want) <= 1e-13
    if lg >= 6:                                                       # the single-precision instantiation (run-time plan, one column)
        x32, H32, out32 = x.astype(np.complex64), H.astype(np.complex64), np.empty((N, ncols), dtype=np.complex64)
        rc = lib.ssf_overlap_save(0, N, ncols, _lib.SSF_C64, nfft, K, H32.ctypes.data_as(C.c_void_p), x32.ctypes.data_as(C.c_void_p),
                                  out32.ctypes.data_as(C.c_void_p))
        _lib.raise_for(lib, None, rc)
        assert rel_l2(out32, want) <= 2e-5


@pytest.mark.parametrize("ntaps", [683, 2049, 4096])
def test_fir_filter_block_sizes_above_682_taps(ntaps):
    """2048-point blocks up to 682 taps, 4096 up to 2048, 8192 above (rx_pipeline.h: fir_nfft)"""
    rng = np.random.default_rng(ntaps)
    x = rng.normal(size=(30000, 2)) + 1j * rng.normal(size=(30000, 2))
    h = rng.normal(size=ntaps)
    y = oa.firFilter(h, x)
    want = np.stack([np.convolve(x[:, c], h, mode="same") for c in range(2)], axis=1)
    assert rel_l2(y, want) <= 1e-13


def test_decimate_large_exact():
    rng = np.random.default_rng(10)
    N, sps = 1 << 20, 16
    x = rng.normal(size=(N, 4)) + 1j * rng.normal(size=(N, 4))
    x[3::sps, 0] *= 3          # make one sampling phase per column stand out
    x[7::sps, 1] *= 3
    x[0::sps, 2] *= 3
    x[15::sps, 3] *= 3
    p = bag(oa.parameters, dict(SpSin=sps, SpSout=2))
    assert np.array_equal(oa.decimate(x, p), orx.decimate(x, bag(oparams, dict(SpSin=sps, SpSout=2))))


def test_edc_filters_a_column_without_imaginary_part_as_a_real_signal():
    """optic/dsp/core.py:1043-1046 looks at the values: such a column comes back with zero imaginary part."""
    from oracle import ssf_oracle as orc
    x = np.random.default_rng(1).normal(size=(4096, 2)) + 0j
    x[:, 1] += 1j * np.random.default_rng(2).normal(size=4096)
    kw = dict(Fs=64e9, L=20, D=16, Fc=193.1e12, Rs=32e9)
    a, b = oa.edc(x, bag(oa.parameters, kw)), orc.edc(x, bag(oparams, kw))
    assert np.all(a[:, 0].imag == 0) and np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b))


def test_error_conventions():
    p = oa.parameters()
    with pytest.raises(AttributeError):
        oa.pdmCoherentReceiver(np.zeros((16, 2), complex), np.zeros(16, complex), p)
    p.Fs = 64e9
    with pytest.raises(AssertionError):
        oa.pdmCoherentReceiver(np.zeros((16, 2), complex), np.zeros(15, complex), p)
    with pytest.raises(AssertionError):
        oa.photodiode(np.zeros(16, complex), bag(oa.parameters, dict(Fs=40e9, B=30e9)))
    with pytest.raises(ValueError):
        oa.decimate(np.zeros(100), bag(oa.parameters, dict(SpSin=16, SpSout=2)))


def test_receiver_throughput_is_reported(capsys):
    """Not a gate: prints samples/s of the full PDM front-end and of firFilter at N = 2^20."""
    N = 1 << 20
    rng = np.random.default_rng(1)
    Es = (rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * 0.02
    Elo = np.full(N, np.sqrt(8e-3), dtype=complex)
    fe, pd = bag(oa.parameters, dict(Fs=96e9)), bag(oa.parameters, dict(Fs=96e9, B=25e9, seed=1))
    oa.pdmCoherentReceiver(Es, Elo, fe, pd)
    t0 = time.perf_counter()
    oa.pdmCoherentReceiver(Es, Elo, fe, pd)
    t1 = time.perf_counter() - t0
    h = oa.lowPassFIR(25e9, 96e9, 255)
    oa.firFilter(h, Es)
    t0 = time.perf_counter()
    oa.firFilter(h, Es)
    t2 = time.perf_counter() - t0
    with capsys.disabled():
        print(f"\n[rx] pdmCoherentReceiver N=2^20: {t1*1e3:.1f} ms wall ({N/t1/1e6:.0f} MS/s); firFilter 255 taps x 2 cols: {t2*1e3:.1f} ms")
