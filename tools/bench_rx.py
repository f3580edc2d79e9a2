#!/usr/bin/env python3
"""Receiver-side workloads on the GPU (secondary to bench.py): wall time per call including the
transfers, for pdmCoherentReceiver / firFilter / decimate at N = 2^20 and 2^22.  Run under
`rocprofv3 --kernel-trace --stats` for the kernel times quoted in DESIGN.md."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticommpy_amd as oa  # noqa: E402


def bag(**kw):
    p = oa.parameters()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def timeit(f, reps=10):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps


def main():
    rng = np.random.default_rng(1)
    for lg in (20, 22):
        N = 1 << lg
        Es = (rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * 0.02
        Elo = np.full(N, np.sqrt(8e-3), dtype=complex)
        fe = bag(Fs=96e9, polRotation=0.2, polDelay=2e-12, timeSkewX=1e-12)
        pd = bag(Fs=96e9, B=25e9, seed=1)
        t = timeit(lambda: oa.pdmCoherentReceiver(Es, Elo, fe, pd))
        moved = (2 + 1 + 2) * 16 * N
        print(f"pdmCoherentReceiver 2^{lg}: {t*1e3:8.2f} ms/call  {N/t/1e6:7.0f} MS/s  (host<->device {moved/2**20:.0f} MiB)")
        h = oa.lowPassFIR(25e9, 96e9, 255)
        t = timeit(lambda: oa.firFilter(h, Es))
        print(f"firFilter 255 taps x 2   2^{lg}: {t*1e3:8.2f} ms/call  {2*N/t/1e6:7.0f} MS/s")
        t = timeit(lambda: oa.decimate(Es, bag(SpSin=16, SpSout=2)))
        print(f"decimate 16 -> 2 x 2     2^{lg}: {t*1e3:8.2f} ms/call  {2*N/t/1e6:7.0f} MS/s")


if __name__ == "__main__":
    main()
