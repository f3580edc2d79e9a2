"""world_size-2 test of the N > 1 path on CPU: the sharding rule, the independence of units, the broadcast of the
parameter block, the scatter of the inputs from the root and the gather of the results.  The product's
communicator is RCCL inside libssf_hip.so (mgpu.RcclComm); here a gloo-backed stand-in with the same interface
(tests/comm_gloo.py) carries the messages and the CPU oracle stands in for the propagation (`compute=`) -- this test
is about the host-side distribution logic; the HIP path is covered by the -m gpu tests."""
import os
import socket
import subprocess
import sys

import numpy as np

from opticommpy_amd import mgpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SSF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SSF_ROOT"], "tests"))
from opticommpy_amd import mgpu
from oracle import ssf_oracle as orc
from helpers import synth_field
from comm_gloo import GlooComm
comm = GlooComm()
rank, world = comm.rank, comm.world
U = 5
fields = [synth_field(256, 2, 100 + u, 8.4 - 0.5 * u) for u in range(U)]
p = orc.parameters()
p.Fs, p.Ltotal, p.Lspan, p.hz, p.amp, p.nlprMethod, p.prgsBar, p.saveSpanN = 512e9, 2, 1, 0.25, "ideal", False, False, []
calls = []
def compute(E, q):
    calls.append(1)
    return orc.manakovSSF(E, q)
outs = mgpu.run_sharded(fields, p, compute=compute, comm=comm)
assert len(calls) == len(mgpu.shard_range(U, world, rank)), (rank, len(calls))
assert not hasattr(p, "maxIter")            # the caller's param object is untouched (deep copies per unit)
np.save(os.path.join(os.environ["SSF_OUT"], f"out_rank{rank}.npy"), np.stack(outs))
# root mode: only rank 0 holds fields and parameters; everything else arrives through the communicator
comm.calls.clear()
outs2 = mgpu.run_sharded(fields if rank == 0 else None, p if rank == 0 else None, compute=compute, comm=comm, root=0,
                         gather="root")
if rank == 0:
    assert all(np.array_equal(a, b) for a, b in zip(outs, outs2))
    sent = [c for c in comm.calls if c[0] == "send"]
    assert len(sent) == len(mgpu.shard_range(U, world, 1)) and all(c[1] == 1 for c in sent)      # inputs of rank 1's block only
else:
    mine = list(mgpu.shard_range(U, world, rank))
    assert all((outs2[u] is not None) == (u in mine) for u in range(U))
    assert sum(1 for c in comm.calls if c[0] == "recv") == len(mine)
t = comm.allreduce(np.array([float(rank + 1)]), "max")
assert t[0] == world
assert mgpu.bcast_object(comm, {"powers": [1.5, 2.5]} if rank == 0 else None, 0) == {"powers": [1.5, 2.5]}
comm.close()
'''


def test_shard_rule_is_contiguous_and_complete():
    for U in (1, 2, 5, 16, 17):
        for G in (1, 2, 4, 8):
            blocks = [list(mgpu.shard_range(U, G, r)) for r in range(G)]
            assert sum(blocks, []) == list(range(U))
            assert max(map(len, blocks)) - min(map(len, blocks)) <= 1
            assert all(mgpu.owner_of(u, U, G) == r for r, b in enumerate(blocks) for u in b)


def test_no_group_runs_everything_locally():
    outs = mgpu.run_sharded([np.ones((4, 2), complex) * k for k in range(3)], object(),
                            compute=lambda E, p: 2 * E)
    assert [o[0, 0] for o in outs] == [0, 2, 4]


def test_two_ranks_gloo(tmp_path):
    from helpers import synth_field
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
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-2000:]
    a = np.load(tmp_path / "out_rank0.npy")
    b = np.load(tmp_path / "out_rank1.npy")
    assert np.array_equal(a, b)                                  # every rank holds the full result
    par = orc.parameters()
    par.Fs, par.Ltotal, par.Lspan, par.hz, par.amp, par.nlprMethod, par.prgsBar, par.saveSpanN = \
        512e9, 2, 1, 0.25, "ideal", False, False, []
    for u in range(5):
        ref = orc.manakovSSF(synth_field(256, 2, 100 + u, 8.4 - 0.5 * u), par.copy())
        assert np.array_equal(a[u], ref)                         # sharded == serial, bit for bit
