"""Round-3 additions: the self-launching bench (--gpus N), the RCCL rendezvous file, per-unit noise streams."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------ rendezvous file (CPU)
def test_rendezvous_file_is_private_fresh_and_keyed_by_the_job(tmp_path, monkeypatch):
    from opticommpy_amd import _lib, mgpu
    R = mgpu.RcclComm
    monkeypatch.delenv("SSF_RCCL_ID_FILE", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29511")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/7")
    p = R.id_path()
    d = os.path.dirname(p)
    assert os.stat(d).st_mode & 0o077 == 0 and os.stat(d).st_uid == os.getuid()
    assert "29511" in p and "job_7" in p and "ppid" not in p       # ranks of one job agree without sharing a parent
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")              # torchrun's static rendezvous names every job "none":
    assert ("ppid%d" % os.getppid()) in R.id_path()                # the launcher's pid keys the file instead
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/7")
    monkeypatch.setenv("SSF_RCCL_ID_FILE", str(tmp_path / "x.id"))
    p = R.id_path()
    assert p == str(tmp_path / "x.id")
    raw = bytes(range(128)) * (_lib.COMM_ID_BYTES // 128)
    # a file a crashed run left behind (valid layout, old time stamp) is not taken for this run's id
    with open(p, "wb") as f:
        f.write(R._pack(raw, stamp=time.time() - 3600.0))
    assert R._read_id(p, mgpu._process_start_time() - R._STALE_S) is None
    # rank 0 replaces it: O_EXCL, 0600, fresh stamp
    R._publish_id(p, raw)
    assert os.stat(p).st_mode & 0o777 == 0o600
    assert R._read_id(p, mgpu._process_start_time() - R._STALE_S) == raw
    # partial / foreign content is "not there yet", never an id
    with open(p, "wb") as f:
        f.write(b"garbage")
    assert R._read_id(p, 0.0) is None
    os.unlink(p)
    assert R._read_id(p, 0.0) is None


def test_rendezvous_times_out_with_a_stale_file(tmp_path, monkeypatch):
    """Rank 1 must not pick up an old id (it would hang in ncclCommInitRank): it reports the missing rendezvous."""
    from opticommpy_amd import _lib, mgpu
    p = tmp_path / "stale.id"
    p.write_bytes(mgpu.RcclComm._pack(b"\0" * _lib.COMM_ID_BYTES, stamp=time.time() - 7200.0))
    for k, v in dict(SSF_RCCL_ID_FILE=str(p), RANK="1", WORLD_SIZE="2", LOCAL_RANK="1").items():
        monkeypatch.setenv(k, v)
    if not os.path.exists(os.path.join(ROOT, "opticommpy_amd", "libssf_hip.so")):
        pytest.skip("library not built")
    with pytest.raises(TimeoutError, match="no fresh id"):
        mgpu.RcclComm.from_env(timeout=0.3)


def test_rendezvous_job_nonce_decides_identity_not_age(tmp_path, monkeypatch):
    """With $SSF_RCCL_NONCE (bench.py --gpus N exports one per launch) a rank takes exactly its own job's id: an hour-old file of
    this job is taken (a rank may start long after rank 0), a fresh file of another job -- a killed run under the same key -- is
    not (advisor, round 3)."""
    from opticommpy_amd import _lib, mgpu
    R = mgpu.RcclComm
    raw = bytes(range(128)) * (_lib.COMM_ID_BYTES // 128)
    p = str(tmp_path / "n.id")
    monkeypatch.setenv("SSF_RCCL_NONCE", "job-A")
    with open(p, "wb") as f:
        f.write(R._pack(raw, stamp=time.time() - 3600.0))
    assert R._read_id(p, time.time()) == raw                      # old, but ours
    monkeypatch.setenv("SSF_RCCL_NONCE", "job-B")
    assert R._read_id(p, 0.0) is None                             # fresh enough for any clock, but another job's
    monkeypatch.delenv("SSF_RCCL_NONCE")
    assert R._read_id(p, 0.0) is None                             # a rank without a nonce does not take a nonce'd job's id


def test_unit_checksum_tells_equal_power_units_apart():
    sys.path.insert(0, ROOT)
    import bench
    rng = np.random.default_rng(0)
    a = rng.normal(size=(2, 256)) + 1j * rng.normal(size=(2, 256))
    b = np.roll(a, 1, axis=1)                                    # same power, different field
    ca, cb = bench.unit_checksum(a), bench.unit_checksum(b)
    assert ca[0] == pytest.approx(cb[0], rel=1e-14) and abs(ca[1] - cb[1]) > 1e-6 * ca[1]
    assert bench.unit_checksum(a) == ca                          # seeded: reproducible


# ------------------------------------------------------------------------------------------ bench.py --gpus N (GPU)
def _bench(args, extra_env=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, timeout=timeout)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_by_itself_config4():
    """`python bench.py --gpus 2 --config 4` with no launcher: two ranks (both on GPU 0 here, over the gloo stand-in, because
    RCCL refuses two ranks on one device), 16 units in two blocks of 8, every unit distinguishable, parity green."""
    r, rec = _bench(["--gpus", "2", "--config", "4", "--steps", "8", "--warmup", "2", "--log2n", "16", "--no-kernel-times"],
                    dict(SSF_BENCH_DEVICE="mod", SSF_BENCH_COMM="gloo"))
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    assert rec["n_gpus"] == 2 and rec["config"]["units_total"] == 16 and rec["config"]["units_per_gpu"] == 8
    cs = rec["unit_checksums"]
    assert len(cs) == 16 and all(len(c) == 3 for c in cs)
    _check_units_against_the_reference(rec, "4", 16)             # every unit, rank 1's block included, is the reference's
    assert rec["parity"]["ok"] and rec["value"] > 0
    assert rec["config"]["unit_steps_total"] == 16 * 8


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,cfg,units", [(4, "5", 8), (8, "4", 16), (8, "5", 8)])
def test_bench_with_four_and_eight_ranks_on_the_stand_in(ranks, cfg, units):
    """The 4- and 8-process shapes of the driver's scaling runs (`bench.py --gpus 8 --config 4 / 5`), as far as ONE GPU can show
    them: eight processes rendezvous, rank 0 synthesises every unit and scatters them, rank r owns the
    contiguous block [r U / G, (r + 1) U / G) (2 / 1 units per rank at eight ranks), the checksums come back through the gather --
    all ranks on this GPU (rank r on device r % count) over the gloo stand-in, because RCCL refuses several ranks on one device.
    EVERY unit's (sum |E|^2, <q, E>, iterations) equals the reference's own run of that unit (wl_units45_n16).  No scaling curve
    can be measured this way: the processes share one GPU."""
    r, rec = _bench(["--gpus", str(ranks), "--config", cfg, "--steps", "8", "--warmup", "2", "--log2n", "16", "--no-kernel-times"],
                    dict(SSF_BENCH_DEVICE="mod", SSF_BENCH_COMM="gloo"), timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    assert rec["n_gpus"] == ranks and rec["scaling"] == "strong"
    assert rec["config"]["units_total"] == units and rec["config"]["units_per_gpu"] == units // ranks
    assert len(rec["unit_checksums"]) == units
    _check_units_against_the_reference(rec, cfg, 16)
    assert rec["parity"]["ok"] and rec["value"] > 0


@pytest.mark.gpu
def test_bench_gpus_2_weak_scaling_default_config():
    r, rec = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--log2n", "16", "--no-kernel-times"],
                    dict(SSF_BENCH_DEVICE="mod", SSF_BENCH_COMM="gloo"))
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and len(rec["unit_checksums"]) == 2
    assert rec["config"]["unit_steps_total"] == 12 and rec["parity"]["ok"]


@pytest.mark.gpu
def test_bench_fails_loudly_when_rccl_cannot_serve_the_ranks():
    """Two ranks on ONE device is something RCCL refuses: the bench must fail, not switch transport silently."""
    r, rec = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--log2n", "14", "--no-kernel-times", "--no-cpu-baseline"],
                    dict(SSF_BENCH_DEVICE="0"), timeout=300)
    assert r.returncode != 0 and rec is None


# ------------------------------------------------------------------------------------------ per-unit / per-rank ASE noise
EDFA_CFG = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=40, Lspan=20,
                hz=2.0, nlprMethod=False, amp="edfa", NF=5.0, seed=77, saveSpanN=[])


@pytest.mark.gpu
def test_sharded_units_with_one_seed_get_independent_noise():
    """Monte-Carlo units that share param.seed must not share their ASE noise (advisor, round 2): unit u draws rows
    u*ncols... of the seed's stream; unit 0 is the stand-alone call; the result does not depend on the lane count."""
    import opticommpy_amd as oa
    from helpers import make_param, synth_field
    from opticommpy_amd import mgpu
    E = synth_field(1 << 12, 2, 5, 0.0)
    p = make_param(oa.parameters, EDFA_CFG)
    outs = mgpu.run_sharded([E, E.copy(), E.copy()], p)
    alone = oa.manakovSSF(E, make_param(oa.parameters, EDFA_CFG))
    assert np.array_equal(outs[0], alone)
    n = [o - np.mean(outs, axis=0) for o in outs]
    d01, d02 = outs[0] - outs[1], outs[0] - outs[2]
    assert np.linalg.norm(d01) > 1e-3 * np.linalg.norm(outs[0])                     # different noise ...
    c = abs(np.vdot(d01, d02)) / (np.linalg.norm(d01) * np.linalg.norm(d02))
    assert 0.3 < c < 0.7                                                            # ... d01, d02 share only unit 0's part
    again = mgpu.run_sharded([E, E.copy(), E.copy()], make_param(oa.parameters, EDFA_CFG))
    assert all(np.array_equal(a, b) for a, b in zip(outs, again))                   # seeded: reproducible


@pytest.mark.gpu
def test_row_offset_continues_the_single_calls_noise_rows():
    """Rows [2, 4) of a K = 2 call draw the same noise as a K = 1 call with rng_row_offset = 2 (what a rank holding the
    second pair of a coupled batch passes): checked at gamma = 0, where the pairs do not interact."""
    import opticommpy_amd as oa
    from helpers import make_param, synth_field
    cfg = dict(EDFA_CFG, gamma=0.0)
    E = np.concatenate([synth_field(1 << 12, 2, 5, 0.0), synth_field(1 << 12, 2, 6, 0.0)], axis=1)
    both = oa.manakovSSF(E, make_param(oa.parameters, cfg))
    p = make_param(oa.parameters, cfg)
    p._rng_row_offset = 2
    second = oa.manakovSSF(np.ascontiguousarray(E[:, 2:]), p)
    assert np.linalg.norm(second - both[:, 2:]) <= 1e-12 * np.linalg.norm(second)
    first = oa.manakovSSF(np.ascontiguousarray(E[:, :2]), make_param(oa.parameters, cfg))
    assert np.linalg.norm(first - both[:, :2]) <= 1e-12 * np.linalg.norm(first)
    p0 = make_param(oa.parameters, cfg)
    wrong = oa.manakovSSF(np.ascontiguousarray(E[:, 2:]), p0)                        # offset 0: the FIRST pair's noise
    assert np.linalg.norm(wrong - both[:, 2:]) > 1e-4 * np.linalg.norm(wrong)


# ------------------------------------------------------------------------------------------ independent units per launch
UNIT_CFG = dict(Fs=64e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=20, Lspan=10,
                hz=0.5, nlprMethod=False, maxNlinPhaseRot=5e-3, amp="edfa", NF=4.5, seed=5, saveSpanN=[])


@pytest.mark.gpu
@pytest.mark.parametrize("N,prec,adaptive", [(1 << 12, "complex128", False), (1 << 14, "complex128", True), (1 << 14, "complex64", False),
                                             (12000, "complex128", False), (1 << 16, "complex128", False)])
def test_units_in_one_launch_are_bit_equal_to_separate_calls(N, prec, adaptive):
    """mgpu.run_sharded sends a rank's small units to the device as ONE batch of independent units (ssf_plan_set_units): same
    launches for all, own control block / step sizes / decisions per unit -> bit-equal to one call per unit, device ASE noise
    included (unit u draws rows 2u, 2u + 1 of the seed's stream either way); and NOT the coupled K = U call."""
    import opticommpy_amd as oa
    from helpers import make_param, synth_field
    from opticommpy_amd import mgpu, models
    dt = np.dtype(prec)
    fields = [synth_field(N, 2, 60 + u, p).astype(dt) for u, p in enumerate((-12.0, 0.0, 6.0, 12.0, 18.0))]
    cfg = dict(UNIT_CFG, nlprMethod=adaptive, prec=prec)
    os.environ["SSF_MGPU_BATCH"] = "0"
    try:
        alone = mgpu.run_sharded(fields, make_param(oa.parameters, cfg))
    finally:
        del os.environ["SSF_MGPU_BATCH"]
    batch = mgpu.run_sharded(fields, make_param(oa.parameters, cfg))
    assert models.last_run["engine"] == "fused"
    for u in range(len(fields)):
        assert batch[u].dtype == dt and np.array_equal(batch[u], alone[u]), u
    coupled = oa.manakovSSF(np.concatenate(fields, axis=1), make_param(oa.parameters, dict(cfg, amp="ideal")))
    ideal = mgpu.run_sharded(fields, make_param(oa.parameters, dict(cfg, amp="ideal")))
    if adaptive:                                                   # the coupled call takes the strongest pair's steps for all
        assert np.linalg.norm(coupled[:, :2] - ideal[0]) > 1e-9 * np.linalg.norm(ideal[0])


@pytest.mark.gpu
def test_units_with_snapshots_and_against_the_oracle():
    import opticommpy_amd as oa
    from helpers import make_param, rel_l2, synth_field
    from opticommpy_amd import mgpu
    from oracle import ssf_oracle as orc
    N = 1 << 13
    fields = [synth_field(N, 2, 70 + u, 3.0 * u) for u in range(4)]
    cfg = dict(UNIT_CFG, amp="ideal", Ltotal=30, saveSpanN=[1, 3])
    outs = mgpu.run_sharded(fields, make_param(oa.parameters, cfg))
    for E, o in zip(fields, outs):
        assert o.shape == (N, 4)
        assert rel_l2(o, orc.manakovSSF(E, make_param(orc.parameters, cfg))) <= 1e-10


@pytest.mark.gpu
def test_units_throughput_at_small_sizes():
    """16 units of 2^14 samples in one launch sequence against one unit at a time: the launches are latency-bound at this size,
    so the batch must be several times faster per unit-step (the round's target: >= 8x; asserted: >= 4x, boxes differ)."""
    import time
    import opticommpy_amd as oa
    from helpers import make_param, synth_field
    from opticommpy_amd import mgpu
    N, U = 1 << 14, 16
    fields = [synth_field(N, 2, 80 + u, 2.0) for u in range(U)]
    cfg = dict(UNIT_CFG, amp="ideal", Ltotal=100, Lspan=100, hz=0.5)
    t = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["SSF_MGPU_BATCH"], os.environ["SSF_MGPU_LANES"] = mode, "1"
        try:
            t0 = time.perf_counter()
            mgpu.run_sharded(fields, make_param(oa.parameters, cfg))
            t[mode] = time.perf_counter() - t0
        finally:
            del os.environ["SSF_MGPU_BATCH"], os.environ["SSF_MGPU_LANES"]
    print("16 units of 2^14, 200 steps each: one at a time %.3f s, batched %.3f s (%.1fx)" % (t["0"], t["1"], t["0"] / t["1"]))
    assert t["0"] / t["1"] >= 4.0


# ------------------------------------------------------------------------------------------ edc with long filters
@pytest.mark.gpu
@pytest.mark.parametrize("L", [9000.0, 20000.0, 40000.0])
def test_edc_filters_longer_than_one_lds_block(L):
    """The reference's edc takes any filter length (optic/dsp/equalization.py:36-122, optic/dsp/core.py:973-1046); round 2
    raised ValueError beyond 8192 taps.  Long impulse responses are now convolved segment by segment."""
    import opticommpy_amd as oa
    from helpers import make_param, rel_l2, synth_field
    from oracle import ssf_oracle as orc
    E = synth_field(1 << 15, 2, 93, 0.0)
    kw = dict(L=L, D=16, Fc=193.1e12, Fs=64e9, Rs=32e9)
    K = oa.models._edc_filter(make_param(oa.parameters, kw), 64e9)[0]
    assert K > 4096
    ref = orc.edc(E, make_param(orc.parameters, kw))
    out = oa.edc(E, make_param(oa.parameters, kw))
    assert rel_l2(out, ref) <= 1e-12
    one = oa.edc(E[:, 0].copy(), make_param(oa.parameters, kw))               # 1-D in, 1-D out
    assert one.shape == (1 << 15,) and rel_l2(one, ref[:, 0]) <= 1e-12
    d = oa.to_device(E)                                                       # device in, device out
    assert rel_l2(oa.edc(d, make_param(oa.parameters, kw)).get(), ref) <= 1e-12


# ------------------------------------------------------------------------------------------ complex64 at notebook lengths
@pytest.mark.gpu
def test_c64_drift_at_a_notebook_length_over_config3_step_count():
    """N = 240 000 = 2^7 * 3 * 5^4 (SpS x Nsymbols), complex64, 10 x 80 km: the mixed-radix rows now apply their pass twiddles,
    radix-3 / 5 constants and the row operator as hi + lo pairs like the power-of-two kernels, so the 5e-4 gate holds over
    10 010 steps (round 2: 7e-4, the drift of the reference's own complex64 path)."""
    import opticommpy_amd as oa
    from helpers import make_param, rel_l2, synth_field
    N = 240000
    E = synth_field(N, 2, 7, 0.0, np.complex64)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
               Ltotal=800, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    outs = {}
    for prec in ("complex128", "complex64"):
        outs[prec] = oa.manakovSSF(E.astype(prec), make_param(oa.parameters, dict(cfg, prec=prec)))
        assert oa.models.last_run["engine"] == "fused"
    a, b = outs["complex64"].astype(np.complex128), outs["complex128"]
    assert rel_l2(a, b) <= 5e-4
    assert abs(np.sum(np.abs(a) ** 2) / np.sum(np.abs(b) ** 2) - 1) <= 2e-4


# ------------------------------------------------------------------------------------------ the longest fields
@pytest.mark.gpu
@pytest.mark.parametrize("N", [1 << 23, (1 << 21) + 1])
def test_longest_fields_stay_on_the_hand_written_kernels(N):
    """2^23 complex128 samples (1024 x 8192 split) and a length whose Bluestein convolution needs 2^23 points (2^21 + 1 =
    3 * 43 * 16 257): round 2 sent both to rocFFT under AUTO.  One step against the oracle."""
    import opticommpy_amd as oa
    from helpers import make_param, rel_l2, synth_field
    from opticommpy_amd import models
    from oracle import ssf_oracle as orc
    E = synth_field(N, 2, 97, 8.4)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=0.08, Lspan=0.08,
               hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
    out = oa.manakovSSF(E, make_param(oa.parameters, cfg))
    assert models.last_run["engine"] == "fused" and models.last_run["iterations"] == tr["iterations"]
    assert rel_l2(out, ref) <= 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("N,engine", [(3000, "auto"), (1500, "auto"), (30030, "auto"), (1 << 12, "rocfft")])
def test_run_sharded_falls_back_to_one_call_per_unit_where_a_plan_cannot_carry_units(N, engine):
    """Small same-shape units are batched into one plan of independent units -- which only the natively split fused pipeline
    takes (ssf_plan_set_units: SSF_ERR_UNSUPPORTED otherwise, 'the caller falls back to one call per unit', include/ssf.h).
    Lengths that AUTO sends to rocFFT or to Bluestein / one-launch rows, and set_engine('rocfft'), must run all the same
    (advisor, round 3: they raised RuntimeError), with the results of the stand-alone calls."""
    import opticommpy_amd as oa
    from helpers import make_param, rel_l2, synth_field
    from opticommpy_amd import mgpu
    from oracle import ssf_oracle as orc
    fields = [synth_field(N, 2, 80 + u, 3.0 * u) for u in range(3)]
    cfg = dict(UNIT_CFG, amp="ideal")
    oa.set_engine(engine)
    try:
        outs = mgpu.run_sharded(fields, make_param(oa.parameters, cfg))
        alone = [oa.manakovSSF(E, make_param(oa.parameters, cfg)) for E in fields]
    finally:
        oa.set_engine("auto")
    for u in range(3):
        assert np.array_equal(outs[u], alone[u]), u
    assert rel_l2(outs[2], orc.manakovSSF(fields[2], make_param(orc.parameters, cfg))) <= 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["complex128", "complex64"])
def test_device_side_coupling_over_a_one_rank_rccl_communicator(monkeypatch, prec):
    """ssf_set_coupling_comm: the device-resident pipeline reduces its partial sums / maxima per rank, all-gathers them over RCCL on
    the plan's stream and reduces over the ranks, host out of the loop (round 4; two ranks: tests/test_multi_gpu_rccl.py).  With
    ONE rank the gathered value is the rank's own: the K = 2 coupled call must come out as without a communicator -- same step
    and iteration counts, field to rounding (the sums are taken in another order) -- adaptive and fixed step, and against the oracle."""
    import opticommpy_amd as oa
    from helpers import make_param, rel_l2, synth_field
    from opticommpy_amd import mgpu, models
    from oracle import ssf_oracle as orc
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("LOCAL_RANK", "0")
    dt = np.dtype(prec)
    E = np.concatenate([synth_field(1 << 13, 2, 11, 3.0), synth_field(1 << 13, 2, 12, 12.0)], axis=1).astype(dt)
    with mgpu.RcclComm.from_env() as comm:
        for adaptive in (True, False):
            cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=8, Lspan=4, hz=0.5,
                       nlprMethod=adaptive, maxNlinPhaseRot=1e-2, amp="ideal", saveSpanN=[], prec=prec)
            plain = oa.manakovSSF(E, make_param(oa.parameters, cfg))
            run0 = (models.last_run["steps"], models.last_run["iterations"], models.last_run["pipeline"])
            out = oa.manakovSSF(E, make_param(oa.parameters, cfg), _coupling=comm)
            run1 = (models.last_run["steps"], models.last_run["iterations"], models.last_run["pipeline"])
            assert run1 == run0 and run1[2] == "fused-device"
            assert rel_l2(out, plain) <= (1e-12 if prec == "complex128" else 1e-6)
            if prec == "complex128":
                tr = {}
                ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
                assert rel_l2(out, ref) <= 1e-10 and run1[:2] == (tr["steps"], tr["iterations"])
        # lengths without a native split keep the host-driven engine + reducer callback
        E3 = np.concatenate([synth_field(3000, 2, 11, 3.0), synth_field(3000, 2, 12, 12.0)], axis=1)
        cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=4, Lspan=4, hz=0.5,
                   nlprMethod=True, maxNlinPhaseRot=1e-2, amp="ideal", saveSpanN=[])
        out3 = oa.manakovSSF(E3, make_param(oa.parameters, cfg), _coupling=comm)
        assert rel_l2(out3, orc.manakovSSF(E3, make_param(orc.parameters, cfg))) <= 1e-10


# ------------------------------------------------------------------------------------------ configs 4 and 5 at workload size
def _check_units_against_the_reference(rec, which, log2n):
    """Every unit's (sum |E|^2, |<q, E>|, iterations) against what the REFERENCE produced for that unit (tests/golden/wl_units45_n*.npz,
    tools/gen_golden.py units45: seeds, launch powers and the DBP leg of bench.py's configs 4 / 5, eight steps): a unit propagated
    with the wrong launch power, the wrong direction or the wrong field shows, not only a duplicated one."""
    from helpers import load_golden
    d, cfg = load_golden("wl_units45_n%d" % log2n)
    assert rec["steps"] == cfg["steps"]
    ref = d["c4"] if which == "4" else d["c5"]
    its = d["c4_iterations"] if which == "4" else d["c5_iterations"].sum(axis=1)
    cs = rec["unit_checksums"]
    assert len(cs) == len(ref)
    for u, (c, r) in enumerate(zip(cs, ref)):
        assert c[0] == pytest.approx(r[0], rel=1e-9), (u, c, r)
        assert abs(c[1] - abs(complex(r[1], r[2]))) <= 1e-9 * np.sqrt(r[0]), (u, c, r)
        assert int(c[2]) == int(its[u]), (u, c, its[u])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,units", [("4", 16), ("5", 8)])
def test_configs_4_and_5_at_workload_size_on_one_gpu(cfg, units):
    """BASELINE configs 4 (16 WDM units of 2^20) and 5 (8 units, forward + manakovDBP chained in HBM) at their full unit size,
    all units on this GPU's two lanes (more GPUs only change who owns which block): EVERY unit's checksum and iteration count
    equal to the reference's own run of that unit (reference-generated fixture), in the bench's parity leg and again here."""
    r, rec = _bench(["--config", cfg, "--steps", "8", "--warmup", "2", "--no-kernel-times", "--parity", "fixture_units"], timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    assert rec["config"]["units_total"] == units and rec["config"]["units_per_gpu"] == units and rec["config"]["lanes_per_gpu"] == 2
    assert rec["parity"]["ok"] and rec["parity"]["units"] == units and rec["parity"]["worst_unit_checksum_err"] <= 1e-9
    _check_units_against_the_reference(rec, cfg, 20)
    assert rec["metric"].endswith("2^20 samples)") and rec["config"]["unit_steps_total"] >= units * 8
