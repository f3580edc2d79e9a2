#!/usr/bin/env python3
"""Two-factor mixed-radix splits N = N1 x N2 (column stage col_mixed_body x mixed-radix rows) measured at notebook lengths against
the rule's own choice (radix-2^n columns x mixed-radix rows where a power-of-two column exists).  Needs the experiment library
(SSF_MIX2 = "N1,C" is read at plan creation by experiment builds only):
    SSF_LIB=opticommpy_amd/libssf_hip_exp.so python tools/exp/mix2_notebook_sweep.py N [N ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SSF_LIB", os.path.join(ROOT, "opticommpy_amd", "libssf_hip_exp.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import make_param, synth_field  # noqa: E402
from opticommpy_amd import _lib, models  # noqa: E402


def smooth(n):
    for q in (2, 3, 5):
        while n % q == 0:
            n //= q
    return n == 1


def run(N, E, cfg):
    models.release_plans()
    oa.manakovSSF(E, make_param(oa.parameters, cfg))
    pl = models._get_plan(N, 2, _lib.SSF_C128)
    pl.lib.ssf_set_profiling(pl.h, 1)
    oa.manakovSSF(E, make_param(oa.parameters, cfg))
    r = dict(models.last_run)
    kt = _lib.KernelTimes()
    pl.lib.ssf_get_kernel_times(pl.h, C.byref(kt))
    pl.lib.ssf_set_profiling(pl.h, 0)
    best = 0.0
    for _ in range(2):
        oa.manakovSSF(E, make_param(oa.parameters, cfg))
        best = max(best, models.last_run["steps"] / (models.last_run["device_ms"] * 1e-3))
    return r["pipeline"], best, kt.row_ms / max(kt.row_n, 1) * 1e3, kt.col_ms / max(kt.col_n, 1) * 1e3


def main():
    for N in [int(a) for a in sys.argv[1:]]:
        E = synth_field(N, 2, 2, 8.4)
        cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, amp="ideal", saveSpanN=[],
                   Ltotal=15.96, Lspan=15.96, hz=0.08, nlprMethod=False, prec=np.complex128)
        os.environ.pop("SSF_MIX2", None)
        pipe, rate, row, col = run(N, E, cfg)
        print(f"N={N} rule: {pipe} {rate:7.0f} steps/s  row {row:6.1f} us  col {col:6.1f} us", flush=True)
        res = []
        for n1 in range(32, 1025):
            if N % n1 or not smooth(n1):
                continue
            n2 = N // n1
            if n2 < 256 or n2 > 8192 or not smooth(n2):
                continue
            for c in (8, 4):
                if 8192 + 2 * c * n1 * 16 > 156 * 1024:
                    continue
                os.environ["SSF_MIX2"] = f"{n1},{c}"
                try:
                    pipe, r, row, col = run(N, E, cfg)
                except Exception as ex:      # a split the engine refuses
                    print(f"   {n1} x {n2} C={c}: {ex}", flush=True)
                    continue
                res.append((r, n1, n2, c, row, col))
                print(f"   {n1:5d} x {n2:5d} C={c}: {r:7.0f} steps/s  row {row:6.1f} us  col {col:6.1f} us  ({r / rate:.2f} x the rule)", flush=True)
        os.environ.pop("SSF_MIX2", None)
        if res:
            r, n1, n2, c, row, col = max(res)
            print(f"N={N} best: {n1} x {n2} C={c} {r:.0f} steps/s = {r / rate:.2f} x the rule's {rate:.0f}", flush=True)


if __name__ == "__main__":
    main()
