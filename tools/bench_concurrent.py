#!/usr/bin/env python3
"""Throughput of several independent config-2 fields on ONE GPU, run back to back on one plan
vs concurrently on 2..4 plans/streams (host threads inside ssf_mgpu_run with a repeated device id)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from opticommpy_amd import _lib, mgpu
sys.path.insert(0, ROOT)
from bench import make_params, synth_field

def main():
    N, U, steps = 1 << 20, 8, 200
    fields = np.stack([synth_field(N, 2, 100 + u, 8.4 - 0.5 * u).T for u in range(U)])
    cp = make_params(_lib, steps, 0.08)
    for devs in ([0], [0, 0], [0, 0, 0], [0, 0, 0, 0]):
        mgpu.run_threads(fields[:len(devs)], cp, devs)          # warm-up (plan creation)
        t0 = time.perf_counter()
        outs, stats = mgpu.run_threads(fields, cp, devs)
        dt = time.perf_counter() - t0
        tot = sum(s["steps"] for s in stats)
        print(f"{len(devs)} concurrent plan(s): {U} fields x {steps} steps in {dt*1e3:8.1f} ms wall (incl. plan creation + PCIe) -> "
              f"{tot/dt:8.0f} field-steps/s; sum device_ms {sum(s['device_ms'] for s in stats):8.1f}", flush=True)

if __name__ == "__main__":
    main()
