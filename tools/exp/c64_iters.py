#!/usr/bin/env python3
"""Why does complex64 need more iterations per step than complex128 late in the span?  Traced lim values, side by side."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa
from helpers import make_param, synth_field
N = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
E = synth_field(N, 2, 2, 8.4)
cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, amp="ideal",
           saveSpanN=[], Ltotal=80, Lspan=80, hz=0.08, nlprMethod=False)
res = {}
for prec in (np.complex128, np.complex64):
    oa.manakovSSF(E.astype(prec), make_param(oa.parameters, dict(cfg, prec=prec)), _trace=True)
    r = oa.last_run
    res[prec] = (np.asarray(r["iters"]), r["lims"])
    print(prec.__name__, "iterations/step", r["iterations"] / r["steps"], "steps", r["steps"])
i128, l128 = res[np.complex128]
i64, l64 = res[np.complex64]
diff = np.nonzero(i128 != i64)[0]
print("steps with different counts:", len(diff), "first", diff[:5])
for s in list(diff[:3]) + list(diff[-2:]):
    print(f"step {s}: c128 it={i128[s]} lims={np.array2string(np.asarray(l128[s]), precision=3)}  c64 it={i64[s]} lims={np.array2string(np.asarray(l64[s]), precision=3)}")
for prec in (np.complex128, np.complex64):
    for tr in (True, False):
        oa.manakovSSF(E.astype(prec), make_param(oa.parameters, dict(cfg, prec=prec)), _trace=tr)
        r = oa.last_run
        print(prec.__name__, "traced" if tr else "untraced", {k: r[k] for k in r if k in ("steps", "iterations", "rebuilt_iterates", "decided_ahead", "nonconverged_steps", "device_ms")})
