#!/usr/bin/env python3
"""Which workgroups get their data late?  (phase build; see tools/phase_timing.py)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SSF_LIB", os.path.join(ROOT, "opticommpy_amd", "libssf_hip_phase.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa
from helpers import synth_field
from opticommpy_amd import _lib, models
E = synth_field(1 << 20, 2, 2, 8.4)
p = oa.parameters()
for k, v in dict(Fs=512e9, Ltotal=4.0, Lspan=4.0, hz=0.08, maxIter=10, tol=1e-5, nlprMethod=False, amp="ideal", prgsBar=False, saveSpanN=[]).items():
    setattr(p, k, v)
models.manakovSSF(E, p)
lib = _lib.load()
buf = np.zeros((4, 4096, 8), dtype=np.uint64)
lib.ssf_debug_marks.argtypes = [C.c_void_p]
lib.ssf_debug_marks(buf.ctypes.data_as(C.c_void_p))
for kind, name in ((0, "row"), (1, "col")):
    m = buf[kind].astype(np.int64)
    used = m[:, 0] > 0
    bids = np.nonzero(used)[0]
    t0 = m[used, 0].min()
    ld = (m[used, 1] - t0) / 100.0
    end = (m[used, 5] - t0) / 100.0
    print(name, "loads-done by XCD (bid % 8):", " ".join(f"{np.median(ld[bids % 8 == x]):.1f}/{ld[bids % 8 == x].max():.1f}" for x in range(8)))
    print(name, "loads-done by bid octile:   ", " ".join(f"{np.median(ld[(bids * 8 // len(bids)) == x]):.1f}/{ld[(bids * 8 // len(bids)) == x].max():.1f}" for x in range(8)))
    late = bids[ld > np.percentile(ld, 90)]
    print(name, "latest 10 %: bids", late[:40], "...", "bid%8 hist", np.bincount(late % 8, minlength=8), "end med/max", np.median(end), end.max())
    order = np.argsort(-end)[:6]
    print(name, "last finishers (bid: loads-done/end us):", " ".join(f"{bids[i]}:{ld[i]:.1f}/{end[i]:.1f}" for i in order),
          "| bid 0:", f"{ld[bids == 0][0]:.1f}/{end[bids == 0][0]:.1f}", "| 99th pct end", f"{np.percentile(end, 99):.1f}")
