#!/usr/bin/env python3
"""Single- vs double-precision agreement over long runs on the GPU (the c128 path is the one
pinned against the oracle; this measures how far c64 drifts from it).  Usage (GPU box):
    python tests/tools/c64_drift.py [log2N] [Ltotal_km]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import rel_l2, synth_field  # noqa: E402
from opticommpy_amd import models  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    Ltotal = float(sys.argv[2]) if len(sys.argv) > 2 else 160.0
    E = synth_field(1 << lg, 2, 7, 0.0)
    out = {}
    for prec in (np.complex128, np.complex64):
        p = oa.parameters()
        for k, v in dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, Ltotal=Ltotal, Lspan=80, hz=0.08,
                         maxIter=10, tol=1e-5, nlprMethod=False, amp="ideal", prgsBar=False, prec=prec).items():
            setattr(p, k, v)
        out[prec] = models.manakovSSF(E, p)
        r = models.last_run
        print(f"{np.dtype(prec).name}: steps={r['steps']} iters={r['iterations']} dev={r['device_ms']:.1f} ms "
              f"engine={r['engine']}", flush=True)
    a, b = out[np.complex64], out[np.complex128]
    pw = lambda x: float(np.sum(np.abs(x.astype(np.complex128)) ** 2))
    print(f"log2N={lg} Ltotal={Ltotal}: c64/c128 power ratio {pw(a) / pw(b):.6f}  rel-L2 {rel_l2(a, b):.3e}")


if __name__ == "__main__":
    main()
