#!/usr/bin/env python3
"""Every pass plan of a mixed-radix row length, measured (experiment library: SSF_MIX_PLAN is read at plan creation).
Prints the average row launch (HIP events, ssf_set_profiling) per plan, fastest first.
    SSF_LIB=opticommpy_amd/libssf_hip_exp.so python tools/exp/mix_plan_sweep2.py N L [maxpasses] [col]
With "col": L is the COLUMN length of the mixed-radix column stage (SSF_MIX_PLAN1), the column launch is reported."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import make_param, synth_field  # noqa: E402
from opticommpy_amd import _lib, models  # noqa: E402

RAD = [25, 20, 16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2]


COL = "col" in sys.argv[1:]
ENV = "SSF_MIX_PLAN1" if COL else "SSF_MIX_PLAN"


def plans(L, maxp, cur=()):
    if L == 1:
        if cur and (COL or cur[-1] <= 16):
            yield cur
        return
    if len(cur) == maxp:
        return
    for r in RAD:
        if L % r == 0:
            yield from plans(L // r, maxp, cur + (r,))


def main():
    N, L = int(sys.argv[1]), int(sys.argv[2])
    maxp = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "col" else 4
    E = synth_field(N, 2, 2, 8.4)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, amp="ideal", saveSpanN=[],
               Ltotal=4.0, Lspan=4.0, hz=0.08, nlprMethod=False)
    res = []
    cand = [()] + [p for p in plans(L, maxp) if min(p) >= 3 or len(p) <= 3]
    for p in cand:
        os.environ.pop(ENV, None)
        if p:
            os.environ[ENV] = ",".join(map(str, p))
        models.release_plans()
        oa.manakovSSF(E, make_param(oa.parameters, cfg))
        pl = models._get_plan(N, 2, _lib.SSF_C128)
        pl.lib.ssf_set_profiling(pl.h, 1)
        oa.manakovSSF(E, make_param(oa.parameters, cfg))
        kt = _lib.KernelTimes()
        pl.lib.ssf_get_kernel_times(pl.h, C.byref(kt))
        pl.lib.ssf_set_profiling(pl.h, 0)
        res.append(((kt.col_ms / kt.col_n if COL else kt.row_ms / kt.row_n) * 1e3, p))
    auto = res[0][0]
    print(f"N={N} L={L}: engine's own plan {auto:.2f} us per {'column' if COL else 'row'} launch; {len(res) - 1} plans tried")
    for us, p in sorted(res)[:12]:
        print(f"   {us:8.2f} us  {','.join(map(str, p)) or 'auto'}")
    print("   ...")
    for us, p in sorted(res)[-3:]:
        print(f"   {us:8.2f} us  {','.join(map(str, p)) or 'auto'}")


if __name__ == "__main__":
    main()
