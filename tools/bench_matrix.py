#!/usr/bin/env python3
"""Secondary workloads on the GPU (not the headline bench): steps/s and algorithmic GB/s for
the other BASELINE.json configurations and a few shapes around them.  Usage (on a GPU box):
    python tools/bench_matrix.py [--engine auto|fused|rocfft]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import synth_field  # noqa: E402
from opticommpy_amd import models  # noqa: E402


def run(label, func, E, **kw):
    p = oa.parameters()
    base = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, amp="ideal",
                saveSpanN=[])
    base.update(kw)
    for k, v in base.items():
        setattr(p, k, v)
    func(E, p)                      # warm-up (plan creation, first launches)
    t0 = time.perf_counter()
    func(E, p)
    wall = time.perf_counter() - t0
    r = models.last_run
    dev = r["device_ms"] * 1e-3
    print(f"{label:58s} steps={r['steps']:6d} it/step={r['iterations']/max(r['steps'],1):4.2f} "
          f"dev={dev*1e3:9.2f} ms  {r['steps']/dev:10.0f} steps/s  {r['bytes_algorithmic']/dev/1e9:8.0f} GB/s(alg) "
          f"wall={wall*1e3:9.2f} ms  engine={r['engine']}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", default="auto")
    a = ap.parse_args()
    oa.set_engine(a.engine)
    E16 = synth_field(1 << 16, 1, 1, 0.0).reshape(-1) * np.sqrt(2)
    run("C1 ssfm 2^16 c128 50km/100 steps", oa.ssfm, E16, Ltotal=50, Lspan=50, hz=0.5, amp=None)
    E20s = synth_field(1 << 20, 1, 1, 0.0).reshape(-1) * np.sqrt(2)
    run("   ssfm 2^20 c128 200 steps", oa.ssfm, E20s, Ltotal=16, Lspan=16, hz=0.08, amp=None)
    E20 = synth_field(1 << 20, 2, 2, 8.4)
    run("C2 manakovSSF 2^20 c128 fixed hz=0.08, 200 steps", oa.manakovSSF, E20, Ltotal=15.96, Lspan=15.96, hz=0.08,
        nlprMethod=False)
    run("   manakovSSF 2^20 c128 adaptive (2e-2 rad), 8 km", oa.manakovSSF, E20, Ltotal=8, Lspan=8, hz=0.08,
        nlprMethod=True, maxNlinPhaseRot=2e-2)
    run("C5 manakovDBP 2^20 c128 hz=10, 80 km", oa.manakovDBP, E20, Ltotal=80, Lspan=80, hz=10, nlprMethod=False)
    run("   manakovDBP 2^20 c128 hz=0.08, 100 steps", oa.manakovDBP, E20, Ltotal=7.96, Lspan=7.96, hz=0.08,
        nlprMethod=False)
    E22 = synth_field(1 << 22, 2, 3, 8.4, np.complex64)
    run("C3 manakovSSF 2^22 c64 fixed hz=0.08, 100 steps", oa.manakovSSF, E22, Ltotal=7.96, Lspan=7.96, hz=0.08,
        nlprMethod=False, prec=np.complex64)
    E20c = synth_field(1 << 20, 2, 3, 8.4, np.complex64)
    run("   manakovSSF 2^20 c64 fixed hz=0.08, 200 steps", oa.manakovSSF, E20c, Ltotal=15.96, Lspan=15.96, hz=0.08,
        nlprMethod=False, prec=np.complex64)
    E22d = synth_field(1 << 22, 2, 3, 8.4)
    run("   manakovSSF 2^22 c128 fixed hz=0.08, 50 steps", oa.manakovSSF, E22d, Ltotal=3.96, Lspan=3.96, hz=0.08,
        nlprMethod=False)
    for lg in (12, 14, 16, 18):
        E = synth_field(1 << lg, 2, 4, 8.4)
        run(f"   manakovSSF 2^{lg} c128 fixed hz=0.08, 200 steps", oa.manakovSSF, E, Ltotal=15.96, Lspan=15.96, hz=0.08,
            nlprMethod=False)
    E20k = synth_field(1 << 20, 8, 5, 8.4)
    run("C4 manakovSSF 2^20 c128 K=4 pairs in one call, 50 steps", oa.manakovSSF, E20k, Ltotal=3.96, Lspan=3.96, hz=0.08,
        nlprMethod=False)
    En = synth_field(960000, 2, 6, 8.4)
    run("   manakovSSF N=960000 (not 2^m) c128, 50 steps", oa.manakovSSF, En, Ltotal=3.96, Lspan=3.96, hz=0.08,
        nlprMethod=False)


if __name__ == "__main__":
    main()
