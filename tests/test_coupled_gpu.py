"""A K > 1 batch of ONE reference call spread over two processes (mgpu.run_coupled / ssf_set_coupling): every process
holds one polarisation pair, the engine's max(phi) and norm sums are all-reduced before they are used, and each process
returns its columns of the single coupled call (reference optic/models/channels.py:394, 517-519).  Rank r runs on
GPU r % count: on a one-GPU box both share GPU 0 (the product's communicator, RCCL, refuses two ranks on one device, so the
gloo stand-in carries the 8- and 16-byte all-reduces; tests/test_multi_gpu_rccl.py runs the same over RCCL where two GPUs exist)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SSF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SSF_ROOT"], "tests"))
import opticommpy_amd as oa
from opticommpy_amd import mgpu, models
from helpers import synth_field, make_param
from comm_gloo import GlooComm
comm = GlooComm()
from opticommpy_amd import _lib
oa.set_device(comm.rank % max(1, _lib.load().ssf_device_count()))       # (rank r on device r % count)
E = np.concatenate([synth_field(4096, 2, 11, 3.0), synth_field(4096, 2, 12, 12.0)], axis=1)      # two pairs, 9 dB apart
cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=8, Lspan=4, hz=0.5,
           nlprMethod=True, maxNlinPhaseRot=1e-2, amp="ideal", saveSpanN=[])
blk = np.ascontiguousarray(E[:, 2 * comm.rank: 2 * comm.rank + 2])
out = mgpu.run_coupled(blk, make_param(oa.parameters, cfg), comm)
np.save(os.path.join(os.environ["SSF_OUT"], f"coupled_rank{comm.rank}.npy"), out)
np.save(os.path.join(os.environ["SSF_OUT"], f"steps_rank{comm.rank}.npy"), np.array([models.last_run["steps"], models.last_run["iterations"]]))
# EDFA with a fixed seed: every rank keys the same noise stream and must draw ITS rows of it
cfg2 = dict(cfg, amp="edfa", NF=5.0, seed=9, nlprMethod=False, hz=1.0, gamma=0.0)
out2 = mgpu.run_coupled(blk, make_param(oa.parameters, cfg2), comm)
np.save(os.path.join(os.environ["SSF_OUT"], f"edfa_rank{comm.rank}.npy"), out2)
comm.close()
'''


def test_two_processes_reproduce_the_single_coupled_call(tmp_path):
    import opticommpy_amd as oa
    from helpers import make_param, rel_l2, synth_field
    from opticommpy_amd import models
    from oracle import ssf_oracle as orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SSF_ROOT=ROOT, SSF_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()[-3000:]
    got = np.concatenate([np.load(tmp_path / f"coupled_rank{r}.npy") for r in range(2)], axis=1)
    steps = [np.load(tmp_path / f"steps_rank{r}.npy") for r in range(2)]
    E = np.concatenate([synth_field(4096, 2, 11, 3.0), synth_field(4096, 2, 12, 12.0)], axis=1)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=8, Lspan=4, hz=0.5,
               nlprMethod=True, maxNlinPhaseRot=1e-2, amp="ideal", saveSpanN=[])
    tr = {}
    ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)                 # the single coupled call
    assert rel_l2(got, ref) <= 1e-10
    assert all(int(s[0]) == tr["steps"] and int(s[1]) == tr["iterations"] for s in steps)       # both ranks: the coupled step sequence
    # ... which is not what the pairs do on their own: the weak pair alone takes far fewer (longer) steps
    alone = orc.manakovSSF(E[:, :2].copy(), make_param(orc.parameters, cfg), trace=(tr0 := {}))
    assert tr0["steps"] < tr["steps"] and rel_l2(got[:, :2], alone) > 1e-6
    # amp='edfa' with a fixed seed: the two ranks' pairs get the noise rows of the single K = 2 call (gamma = 0: no coupling
    # through the field), not the same rows twice (advisor, round 2)
    cfg2 = dict(cfg, amp="edfa", NF=5.0, seed=9, nlprMethod=False, hz=1.0, gamma=0.0)
    single = oa.manakovSSF(E, make_param(oa.parameters, cfg2))
    got2 = np.concatenate([np.load(tmp_path / f"edfa_rank{r}.npy") for r in range(2)], axis=1)
    assert rel_l2(got2, single) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,npart", [(2, 512), (8, 128), (3, 777)])
def test_rank_order_reduction_of_a_coupled_batch_on_synthetic_rank_buffers(nranks, npart):
    """The arithmetic behind ssf_set_coupling_comm with partials that DIFFER between the ranks (with one RCCL
    rank the gathered value is the rank's own): every rank's per-workgroup partials are reduced on the device (k_couple_local),
    the per-rank results in rank order (k_couple_finish).  Checked bit for bit against a numpy model of exactly that order --
    the property that makes every rank take the same decisions -- and to rounding against the single K-pair call's reduction
    over all workgroups at once (the sums agree to ~1e-16 relative, the maximum exactly)."""
    import ctypes as C
    from opticommpy_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(nranks * 1000 + npart)
    parts = rng.random((nranks, 5, npart)) * np.array([1e-2, 1e-9, 1e-3, 1e-9, 1e-3])[None, :, None]      # pmax, pnum, pden, pnum0, pden0
    out = np.zeros(5)
    _lib.raise_for(lib, None, lib.ssf_couple_reduce_selftest(0, nranks, npart, parts.ctypes.data_as(C.POINTER(C.c_double)),
                                                            out.ctypes.data_as(C.POINTER(C.c_double))))

    def local(v, is_max):                       # k_couple_local: 256 threads stride the partials, thread 0..255 combined in order
        lanes = []
        for t in range(256):
            x = v[t::256]
            if is_max:
                lanes.append(np.max(x) if len(x) else -np.inf)
            else:
                s = 0.0
                for y in x:
                    s += y
                lanes.append(s)
        r = lanes[0]
        for y in lanes[1:]:
            r = max(r, y) if is_max else r + y
        return r
    want = []
    for q, src in enumerate((3, 4, 1, 2, 0)):   # out5 = (sum pnum0, sum pden0, sum pnum, sum pden, max pmax)
        per_rank = [local(parts[r, src], q == 4) for r in range(nranks)]
        r0 = per_rank[0]
        for y in per_rank[1:]:
            r0 = max(r0, y) if q == 4 else r0 + y
        want.append(r0)
    assert out.tolist() == want                 # bit for bit: the same on every rank, whatever its own partials are
    single = [parts[:, 3].sum(), parts[:, 4].sum(), parts[:, 1].sum(), parts[:, 2].sum(), parts[:, 0].max()]
    np.testing.assert_allclose(out, single, rtol=1e-14)
    assert out[4] == single[4]
