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
    monkeypatch.setenv("SSF_RCCL_ID_FILE", str(tmp_path / "x.id"))
    p = R.id_path()
    assert p == str(tmp_path / "x.id")
    raw = bytes(range(128)) * (_lib.COMM_ID_BYTES // 128)
    # a file a crashed run left behind (valid layout, old time stamp) is not taken for this run's id
    with open(p, "wb") as f:
        f.write(R._MAGIC + np.array([time.time() - 3600.0]).tobytes() + raw)
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
    p.write_bytes(mgpu.RcclComm._MAGIC + np.array([time.time() - 7200.0]).tobytes() + b"\0" * _lib.COMM_ID_BYTES)
    for k, v in dict(SSF_RCCL_ID_FILE=str(p), RANK="1", WORLD_SIZE="2", LOCAL_RANK="1").items():
        monkeypatch.setenv(k, v)
    if not os.path.exists(os.path.join(ROOT, "opticommpy_amd", "libssf_hip.so")):
        pytest.skip("library not built")
    with pytest.raises(TimeoutError, match="no fresh id"):
        mgpu.RcclComm.from_env(timeout=0.3)


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
    r, rec = _bench(["--gpus", "2", "--config", "4", "--steps", "6", "--warmup", "2", "--log2n", "16", "--no-kernel-times"],
                    dict(SSF_BENCH_DEVICE="0", SSF_BENCH_COMM="gloo"))
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    assert rec["n_gpus"] == 2 and rec["config"]["units_total"] == 16 and rec["config"]["units_per_gpu"] == 8
    cs = rec["unit_checksums"]
    assert len(cs) == 16 and all(len(c) == 2 for c in cs)
    assert len({round(c[1], 9) for c in cs}) == 16               # no swapped / duplicated unit
    assert rec["parity"]["ok"] and rec["value"] > 0
    assert rec["config"]["unit_steps_total"] == 16 * 6


@pytest.mark.gpu
def test_bench_gpus_2_weak_scaling_default_config():
    r, rec = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--log2n", "16", "--no-kernel-times"],
                    dict(SSF_BENCH_DEVICE="0", SSF_BENCH_COMM="gloo"))
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
