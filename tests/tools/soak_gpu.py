#!/usr/bin/env python3
"""Soak test on the GPU: plans created and destroyed in a loop with every feature that owns threads, streams or pinned
memory (snapshot sink to host and device, device arrays, traces, both complex64 pipelines, the general-length engine, the
RCCL communicator), watching the free device memory.  Usage (GPU box): python tests/tools/soak_gpu.py [rounds]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import make_param, synth_field  # noqa: E402
from opticommpy_amd import device, mgpu, models  # noqa: E402


def free_mem():
    hip = C.CDLL("libamdhip64.so")
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    base = None
    for r in range(rounds):
        for N, prec in ((1 << 14, "complex128"), (1 << 15, "complex64"), (12000, "complex128"), (1009, "complex128")):
            E = synth_field(N, 2, r, 6.0, np.dtype(prec).type)
            cfg = dict(Fs=512e9, Ltotal=3, Lspan=1, hz=0.25, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, amp="edfa", seed=r + 1, NF=5,
                       nlprMethod=bool(r & 1), prgsBar=False, maxIter=10, tol=1e-5, prec=prec, saveSpanN=[1, 3])
            a = oa.manakovSSF(E, make_param(oa.parameters, cfg), _trace=True)
            b = oa.manakovSSF(oa.to_device(E), make_param(oa.parameters, cfg)).get()
            assert a.shape == (N, 4) and np.all(np.isfinite(a)) and np.array_equal(a, b), (r, N, prec)
            oa.manakovDBP(a[:, 2:4].copy(), make_param(oa.parameters, dict(cfg, saveSpanN=[], amp="ideal")))
            oa.ssfm(E[:, 0].copy(), make_param(oa.parameters, dict(cfg, saveSpanN=[2], hz=0.5)))
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        with mgpu.RcclComm.from_env() as comm:
            comm.barrier()
        models.release_plans()
        device.release_pool()
        f = free_mem()
        if r == 1:
            base = f                      # (after the first rounds: runtime pools are warm)
        if r % 5 == 0:
            print(f"round {r}: free device memory {f / 2**20:.0f} MiB", flush=True)
    print(f"done: {rounds} rounds; free memory drift {(base - f) / 2**20:+.1f} MiB")
    assert base - f < 256 << 20, "device memory leak"


if __name__ == "__main__":
    main()
