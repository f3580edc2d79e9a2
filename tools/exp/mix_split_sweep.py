#!/usr/bin/env python3
"""Column length of the mixed-radix split, measured (SSF_MIX_L1 is read at plan creation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa
from helpers import make_param, synth_field
from opticommpy_amd import models
for N in (1310720, 1920000, 1536000, 1440000, 2880000, 1572864):
    E = synth_field(N, 2, 2, 8.4)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
               amp="ideal", saveSpanN=[], Ltotal=15.96, Lspan=15.96, hz=0.08, nlprMethod=False)
    for l in ("", "7", "8", "9", "10"):
        os.environ.pop("SSF_MIX_L1", None)
        if l: os.environ["SSF_MIX_L1"] = l
        models.release_plans()
        try:
            oa.manakovSSF(E, make_param(oa.parameters, cfg))
            oa.manakovSSF(E, make_param(oa.parameters, cfg))
            r = models.last_run
            print(f"N={N} l1={l or 'auto':5s} {r['engine']} {r['steps'] / (r['device_ms'] * 1e-3):8.0f} steps/s", flush=True)
        except Exception as ex:
            print(f"N={N} l1={l}: {ex}", flush=True)
