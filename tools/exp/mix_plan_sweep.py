#!/usr/bin/env python3
"""Row-pass radix plans of the mixed-radix path, measured: SSF_MIX_PLAN / SSF_MIX_TPR are read at plan creation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa
from helpers import make_param, synth_field
from opticommpy_amd import models
CASES = {960000: ["", "25,25,3", "15,5,5,5", "25,15,5", "15,25,5", "5,5,5,5,3", "3,25,25", "5,15,25"],
         240000: ["", "25,25,3", "15,5,5,5", "25,15,5"],
         786432: ["", "16,16,3", "8,8,4,3", "3,16,16", "16,8,6"],
         48000: ["", "25,15", "15,5,5", "5,5,5,3"]}
for N, plans in CASES.items():
    E = synth_field(N, 2, 2, 8.4)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
               amp="ideal", saveSpanN=[], Ltotal=15.96, Lspan=15.96, hz=0.08, nlprMethod=False)
    for tpr in ("128", "256", "64"):
        for plan in plans:
            os.environ["SSF_MIX_TPR"] = tpr
            os.environ.pop("SSF_MIX_PLAN", None)
            if plan: os.environ["SSF_MIX_PLAN"] = plan
            models.release_plans()
            try:
                oa.manakovSSF(E, make_param(oa.parameters, cfg))
                oa.manakovSSF(E, make_param(oa.parameters, cfg))
                r = models.last_run
                print(f"N={N} tpr={tpr} plan={plan or 'auto':12s} {r['steps'] / (r['device_ms'] * 1e-3):8.0f} steps/s", flush=True)
            except Exception as ex:
                print(f"N={N} tpr={tpr} plan={plan}: {ex}", flush=True)
