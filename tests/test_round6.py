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


# ------------------------------------------------------------------------------------------ mixed-radix COLUMN stage (emulator)
# N = N1 x N2 with both factors 2^a 3^b 5^c (fused_kernels.h: col_mixed_body): what takes the lengths whose power-of-two part is too
# small for the radix-2^n columns -- the reference benchmark's 2 000 000 = 2^7 x 5^6 among them -- onto the device-resident pipeline.
def _mix2(N, prec=1):
    import ctypes as C
    e = eb.load()
    e.emu_mixed2_split.argtypes = [C.c_int64, C.c_int] + [C.POINTER(C.c_int)] * 3
    n1, n2, c = C.c_int(0), C.c_int(0), C.c_int(0)
    return (n1.value, n2.value, c.value) if e.emu_mixed2_split(N, prec, C.byref(n1), C.byref(n2), C.byref(c)) else None


def test_the_split_of_the_reference_benchmarks_top_size():
    for N in (2_000_000, 5_000_000, 15_625, 10_125, 9_000):
        sp = _mix2(N)
        assert sp is not None and sp[0] * sp[1] == N and 16 <= sp[0] <= 1024 and 64 <= sp[1] <= 8192 and sp[2] in (2, 4, 8), (N, sp)
    assert _mix2(2_000_000) == (500, 4000, 4)                     # two workgroups per CU in both stages (fused_engine.h)
    assert _mix2(240_000) is None and _mix2(800_000) is None and _mix2(1 << 20) is None      # (the radix-2^n columns keep what they take ...
    assert _mix2(200_000) == (125, 1600, 4)                       #  ... when the length has at least seven factors of two: 2 x 64 rows of 3125 would not fill the chip)
    assert _mix2(7 * 4096) is None                                # (not 5-smooth: Bluestein)


MK = dict(Fs=512e9, Ltotal=4.0, Lspan=2.0, hz=0.25, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, maxIter=10, tol=1e-5, prgsBar=False)


@pytest.mark.parametrize("N, power", [(9000, 8.4), (10125, 13.0), (15625, 3.0)])
@pytest.mark.parametrize("adaptive", [False, True])
def test_manakov_on_the_mixed_radix_column_stage_against_the_oracle(N, power, adaptive):
    assert _mix2(N) is not None
    E = synth_field(N, 2, 31, power)
    cfg = dict(MK, nlprMethod=adaptive, maxNlinPhaseRot=2e-2, amp="ideal", saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, bag(orc.parameters, **cfg), trace=tr)
    out, info = eb.run("manakovSSF", E, cfg)
    assert rel_l2(out.T, ref) <= 1e-11
    assert list(info["iters"]) == list(tr["iters"]) and np.allclose(info["hz"], tr["hz"], rtol=1e-9)
    for a, b in zip(info["lims"], tr["lims"]):
        assert np.allclose(a, b, rtol=1e-6, atol=1e-14)
    back, _ = eb.run("manakovDBP", ref, cfg)                       # ... and backwards
    assert rel_l2(back.T, orc.manakovDBP(ref, bag(orc.parameters, **cfg))) <= 1e-11


def test_mixed_radix_columns_complex64_two_pairs_and_the_rare_stages():
    N = 5625                                                       # = 3^2 x 5^4: odd, 75 x 75
    E4 = np.concatenate([synth_field(N, 2, 41, 8.0), synth_field(N, 2, 42, 2.0)], axis=1)
    cfg = dict(MK, nlprMethod=False, amp=None, saveSpanN=[])
    out, info = eb.run("manakovSSF", E4, cfg)                      # two coupled pairs (K = 2)
    tr = {}
    ref = orc.manakovSSF(E4, bag(orc.parameters, **cfg), trace=tr)
    assert rel_l2(out.T, ref) <= 1e-11 and list(info["iters"]) == list(tr["iters"])
    E = synth_field(N, 2, 43, 8.0)
    c64 = dict(cfg, prec="complex64")
    o64, i64 = eb.run("manakovSSF", E.astype(np.complex64), c64)
    r128 = orc.manakovSSF(E.astype(np.complex64).astype(np.complex128), bag(orc.parameters, **cfg))
    assert rel_l2(o64.T, r128) <= 5e-5
    # weak nonlinearity: lim_0 < tol at every step -> iterate 0 is rebuilt as the final one (ST_REDO0); without a trace the bound of
    # lim_0 cannot exclude it either: rebuilt once to measure, once as final -- the same field either way
    weak = synth_field(N, 2, 44, -10.0)
    wcfg = dict(func="manakovSSF", alpha=0.0, D=1e-5, gamma=1e-6, Fc=193.1e12, Fs=64e9, maxIter=10, tol=1e-5, prgsBar=False,
                Ltotal=0.4, Lspan=0.2, hz=0.05, nlprMethod=False, amp=None, saveSpanN=[])
    tr = {}
    wref = orc.manakovSSF(weak, bag(orc.parameters, **{k: v for k, v in wcfg.items() if k != "func"}), trace=tr)
    assert set(tr["iters"]) == {1}
    w1, wi1 = eb.run("manakovSSF", weak, wcfg)
    w2, wi2 = eb.run("manakovSSF", weak, wcfg, trace=False)
    assert list(wi1["iters"]) == list(tr["iters"]) and rel_l2(w1.T, wref) <= 1e-11 and np.array_equal(w1, w2)
    assert wi1["rebuilt_iterates"] == wi1["steps"] and wi2["rebuilt_iterates"] == 2 * wi2["steps"] and wi2["iterations"] == wi1["iterations"]
    # a tolerance around the bound of lim_0 (~ lim_0 / 4): some steps need the exact lim_0, whose step-start field was stored at one
    # sample in sixteen only -- recovered from E_hd (ST_RECOVER_A / _ROW / _B); same steps, iterations and field as the traced run
    mid = synth_field(N, 2, 45, 0.0)
    mcfg = dict(cfg, Ltotal=10, Lspan=10, hz=0.5)
    _, it = eb.run("manakovSSF", mid, mcfg)
    lim0 = sorted(float(x[0]) for x in it["lims"])
    hit = 0
    for f in (0.26, 0.255):
        c2 = dict(mcfg, tol=lim0[len(lim0) // 2] * f)
        tr = {}
        mref = orc.manakovSSF(mid, bag(orc.parameters, **c2), trace=tr)
        a, ia = eb.run("manakovSSF", mid, c2)
        b, ib = eb.run("manakovSSF", mid, c2, trace=False)
        assert list(ia["iters"]) == list(tr["iters"]) and rel_l2(a.T, mref) <= 1e-11 and rel_l2(b, a) <= 1e-13
        assert (ia["steps"], ia["iterations"]) == (ib["steps"], ib["iterations"]) and ia["recovered_fields"] == 0
        hit += ib["recovered_fields"]
    assert hit > 0


def test_ssfm_and_linear_channel_on_the_mixed_radix_column_stage():
    N = 10125
    E = synth_field(N, 1, 51, 5.0)[:, 0]
    cfg = dict(Fs=512e9, Ltotal=3.0, Lspan=1.5, hz=0.25, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, amp="ideal", prgsBar=False, saveSpanN=[])
    out, info = eb.run("ssfm_final", E, cfg)
    ref = orc.ssfm(E, bag(orc.parameters, **cfg))
    assert rel_l2(out[0], ref) <= 1e-12
    lp = bag(orc.parameters, Fs=512e9, L=3.0, alpha=0.2, D=16, Fc=193.1e12)
    E2 = synth_field(N, 2, 52, 0.0)
    assert rel_l2(eb.linear_channel(E2, 512e9, 193.1e12, 0.2, 16, 3.0), orc.linearFiberChannel(E2, lp)) <= 1e-13


def test_forcing_a_radix_2n_length_onto_the_mixed_columns_gives_the_same_field(monkeypatch):
    """12 000 = 2^5 x 375 runs on the radix-2^n columns; forced onto the mixed-radix column stage (experiment knob SSF_MIX2, read by
    the emulator's experiment build) both pipelines must agree with the oracle and with each other to rounding."""
    N = 12000
    E = synth_field(N, 2, 61, 8.4)
    cfg = dict(MK, nlprMethod=False, amp="ideal", saveSpanN=[])
    a, ia = eb.run("manakovSSF", E, cfg)
    monkeypatch.setenv("SSF_MIX2", "125,8")
    assert _mix2(N) == (125, 96, 8)
    b, ib = eb.run("manakovSSF", E, cfg)
    ref = orc.manakovSSF(E, bag(orc.parameters, **cfg))
    assert rel_l2(a.T, ref) <= 1e-11 and rel_l2(b.T, ref) <= 1e-11 and rel_l2(a, b) <= 1e-12
    assert list(ia["iters"]) == list(ib["iters"])


# ------------------------------------------------------------------------------------------ GPU: the reference benchmark's own lengths
@pytest.mark.gpu
@pytest.mark.parametrize("N", [2_000_000, 200_000, 800_000, 960_000])
def test_reference_notebook_lengths_run_device_resident_and_match_the_oracle(N):
    """examples/benchmarck_GPU_processing.ipynb: 16-QAM, SpS 4, adaptive step, 5e4 ... 5e5 symbols = 2e5 ... 2e6 samples.  2e6 =
    2^7 x 5^6 used to fall to the host-driven Bluestein path; it now runs on the device-resident pipeline (mixed-radix column stage).
    800 000 = 256 x 3125 (the bench leg's middle length) and 960 000 = 256 x 3750: radix-2^n columns x mixed-radix rows at full size."""
    E = synth_field(N, 2, 71, -2.0 + 3.0)
    cfg = dict(Fs=128e9, Ltotal=12.0, Lspan=12.0, hz=0.5, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, maxIter=5, tol=1e-5, nlprMethod=True,
               maxNlinPhaseRot=2e-2, amp="ideal", saveSpanN=[], prgsBar=False)
    tr = {}
    ref = orc.manakovSSF(E, bag(orc.parameters, **cfg), trace=tr)
    out = oa.manakovSSF(E, bag(**cfg), _trace=True)
    assert oa.last_run["pipeline"] == "fused-device"
    assert rel_l2(out, ref) <= 1e-10
    assert [int(x) for x in oa.last_run["iters"]] == [int(x) for x in tr["iters"]]
    assert np.allclose(oa.last_run["hz"], tr["hz"], rtol=1e-9)
    fix = dict(cfg, nlprMethod=False, hz=2.0)                              # fixed step, complex64, and backwards
    o64 = oa.manakovSSF(E.astype(np.complex64), bag(prec=np.complex64, **fix))
    assert oa.last_run["pipeline"] == "fused-device" and o64.dtype == np.complex64
    r128 = orc.manakovSSF(E.astype(np.complex64).astype(np.complex128), bag(orc.parameters, **fix))
    assert rel_l2(o64, r128) <= 5e-4
    back = oa.manakovDBP(ref, bag(**fix))
    assert rel_l2(back, orc.manakovDBP(ref, bag(orc.parameters, **fix))) <= 1e-10


# ------------------------------------------------------------------------------------------ receiver chain in one call
def _chain_case(N, ntaps, sps, L_edc, seed=81, pd=None, **fe_kw):
    E = synth_field(N, 2, seed, 0.0)
    lo = np.sqrt(5e-3) * np.exp(1j * 2 * np.pi * 2e8 * np.arange(N) / 64e9)
    fe = dict(Fs=64e9, polRotation=0.3, polDelay=2e-12, **fe_kw)
    pd = pd or dict(Fs=64e9, ideal=True)
    rng = np.random.default_rng(seed)
    h = rng.normal(size=ntaps) * np.hanning(ntaps)
    dec = dict(SpSin=sps, SpSout=2)
    edcp = dict(Fs=64e9 * 2 / sps, L=L_edc, D=16, Fc=193.1e12, Rs=32e9)
    return E, lo, fe, pd, h, dec, edcp


def _oracle_chain(E, lo, fe, pd, h, dec, edcp):
    s = orx.pdmCoherentReceiver(E, lo, bag(orc.parameters, **fe), bag(orc.parameters, **pd))
    s = orx.firFilter(h, s)
    s = orx.decimate(s, bag(orc.parameters, **dec))
    return orc.edc(s, bag(orc.parameters, **edcp))


@pytest.mark.parametrize("N, ntaps, sps, L", [(1 << 14, 129, 16, 800), (12288, 255, 8, 400), (1 << 13, 33, 4, 5)])
def test_receiver_chain_in_one_call_on_the_emulator(emu_rx, N, ntaps, sps, L):
    """fused geometry (2048-point matched filter, SpS | 128), another one, and one whose transforms have no chain kernel (the stages
    then run one by one inside the same call): always the result of the four reference functions in a row."""
    case = _chain_case(N, ntaps, sps, L)
    E, lo, fe, pd, h, dec, edcp = case
    out = oa.pdmCoherentReceiverChain(E, lo, bag(**fe), bag(**pd), h, bag(**dec), bag(**edcp))
    ref = _oracle_chain(*case)
    assert out.shape == ref.shape and out.dtype == np.complex128
    assert np.max(np.abs(out - ref)) <= 1e-11 * np.max(np.abs(ref))


def test_receiver_chain_with_band_limited_photodiodes_and_with_skew_on_the_emulator(emu_rx):
    """the receiver's detection is NOT left to the matched filter's loads when a filter of its own (the photodiodes' low-pass) or
    the skew filters follow it"""
    for kw in (dict(pd=dict(Fs=64e9, B=20e9, N=127, shotNoise=False, thermalNoise=False)), dict(timeSkewX=2e-12, ampImbY=0.4, phaseImbX=0.1)):
        case = _chain_case(1 << 13, 129, 16, 600, seed=83, **kw)
        E, lo, fe, pd, h, dec, edcp = case
        out = oa.pdmCoherentReceiverChain(E, lo, bag(**fe), bag(**pd), h, bag(**dec), bag(**edcp))
        ref = _oracle_chain(*case)
        assert np.max(np.abs(out - ref)) <= 1e-11 * np.max(np.abs(ref)), kw


@pytest.mark.gpu
def test_receiver_chain_in_one_call_on_the_gpu_equals_the_four_calls():
    from opticommpy_amd import device
    case = _chain_case(1 << 18, 1024, 16, 800, ampImbX=0.5, timeSkewY=1e-12)
    E, lo, fe, pd, h, dec, edcp = case
    Ed, Ld = oa.to_device(E), oa.to_device(lo)
    c0 = device.transfer_counts()
    one = oa.pdmCoherentReceiverChain(Ed, Ld, bag(**fe), bag(**pd), h, bag(**dec), bag(**edcp))
    four = oa.edc(oa.decimate(oa.firFilter(h, oa.pdmCoherentReceiver(Ed, Ld, bag(**fe), bag(**pd))), bag(**dec)), bag(**edcp))
    assert isinstance(one, oa.DeviceArray) and device.transfer_counts() == c0
    a, b = one.get(), four.get()
    assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(b))
    ref = _oracle_chain(*case)
    assert np.max(np.abs(a - ref)) <= 1e-11 * np.max(np.abs(ref))
    host = oa.pdmCoherentReceiverChain(E, lo, bag(**fe), bag(**pd), h, bag(**dec), bag(**edcp))       # numpy in, numpy out
    assert np.array_equal(host, a)


@pytest.mark.gpu
def test_mixed_radix_column_stage_through_the_whole_public_api():
    """Snapshots, the EDFA epilogue (supplied and device noise), a coupled K = 2 call, independent units per launch and ssfm with
    saveSpanN at a length of the mixed-radix column stage (9 000 = 2^3 x 3^2 x 5^3)."""
    from opticommpy_amd import mgpu
    N = 9000
    E = synth_field(N, 2, 91, 6.0)
    base = dict(Fs=512e9, Ltotal=6.0, Lspan=2.0, hz=0.25, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, maxIter=10, tol=1e-5, nlprMethod=True,
                maxNlinPhaseRot=5e-3, prgsBar=False)
    snap = dict(base, amp="ideal", saveSpanN=[1, 3])
    out = oa.manakovSSF(E, bag(**snap))
    assert oa.last_run["pipeline"] == "fused-device" and out.shape == (N, 4)
    assert rel_l2(out, orc.manakovSSF(E, bag(orc.parameters, **snap))) <= 1e-10
    rng = np.random.default_rng(2)
    noise = 1e-4 * (rng.normal(size=(3, 2, N)) + 1j * rng.normal(size=(3, 2, N)))
    ed = dict(base, amp="edfa", NF=5.0, saveSpanN=[])
    assert rel_l2(oa.manakovSSF(E, bag(**ed), _noise=noise), orc.manakovSSF(E, bag(orc.parameters, **ed), noise=noise)) <= 1e-10
    o1 = oa.manakovSSF(E, bag(seed=3, **dict(ed, Ltotal=2.0)))
    oz = oa.manakovSSF(E, bag(**dict(ed, Ltotal=2.0)), _noise=np.zeros((1, 2, N), complex))
    _, p_noise = orc.edfa_noise_power(0.2 * 2.0, 5.0, 193.1e12, 512e9)
    assert np.mean(np.abs(o1 - oz) ** 2) == pytest.approx(p_noise, rel=0.1)
    E4 = np.concatenate([E, synth_field(N, 2, 92, 0.0)], axis=1)               # one coupled call with two pairs
    k2 = dict(base, amp=None, saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E4, bag(orc.parameters, **k2), trace=tr)
    got = oa.manakovSSF(E4, bag(**k2), _trace=True)
    assert rel_l2(got, ref) <= 1e-10 and [int(x) for x in oa.last_run["iters"]] == [int(x) for x in tr["iters"]]
    units = [synth_field(N, 2, 93 + u, 3.0 * u) for u in range(3)]            # independent units: one batch per launch == one call each
    fx = dict(base, nlprMethod=False, amp="ideal", saveSpanN=[])
    batch = mgpu.run_sharded(units, bag(**fx))
    for u, o in zip(units, batch):
        assert np.array_equal(o, oa.manakovSSF(u, bag(**fx)))
    assert rel_l2(batch[2], orc.manakovSSF(units[2], bag(orc.parameters, **fx))) <= 1e-10
    e1 = E[:, 0].copy()
    sc = dict(Fs=512e9, Ltotal=4.0, Lspan=2.0, hz=0.25, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, amp="ideal", prgsBar=False, saveSpanN=[1, 2])
    s = oa.ssfm(e1, bag(**sc))
    assert s.shape == (N, 2) and rel_l2(s, orc.ssfm(e1, bag(orc.parameters, **sc))) <= 1e-10
    back = oa.manakovDBP(ref[:, :2].copy(), bag(**dict(k2, amp="ideal")))
    assert rel_l2(back, orc.manakovDBP(ref[:, :2].copy(), bag(orc.parameters, **dict(k2, amp="ideal")))) <= 1e-10
