#!/usr/bin/env python3
"""ssf_mgpu_run with one or two lanes per device: 8 independent 2-pol fields of 2^20 samples on GPU 0."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa
from helpers import make_param, synth_field
from opticommpy_amd import _lib, mgpu, models
N, U = 1 << 20, 8
fields = np.stack([synth_field(N, 2, 60 + u, 8.4).T for u in range(U)])
cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, amp="ideal",
           saveSpanN=[], Ltotal=15.96, Lspan=15.96, hz=0.08, nlprMethod=False, NF=4.5)
cp = models._fill_params(_lib.MODEL_MANAKOV, +1, make_param(oa.parameters, cfg), 512e9, 1, np.zeros(0, np.int32))
ref = None
for lanes in ("1", "2", "1", "2", "3"):
    os.environ["SSF_MGPU_LANES"] = lanes
    t0 = time.perf_counter()
    outs, stats = mgpu.run_threads(fields, cp, devices=[0])
    dt = time.perf_counter() - t0
    steps = sum(s["steps"] for s in stats)
    same = "" if ref is None else f" identical={np.array_equal(outs, ref)}"
    ref = outs if ref is None else ref
    print(f"lanes={lanes}: {U} fields, {steps} field-steps in {dt*1e3:.1f} ms wall = {steps/dt:.0f} field-steps/s{same}", flush=True)
