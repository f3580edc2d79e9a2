#!/usr/bin/env python3
"""In-kernel phase timing of the fused engine (diagnostic build: make -C opticommpy_amd/csrc phase).
Runs config-2-like steps with libssf_hip_phase.so, then prints, for the last row launch and the
last column launch, when each phase ended (µs after the first workgroup entered), as
min / median / max over the workgroups.  mark() drains vmcnt/lgkmcnt, so phases do not overlap as
they do in the product build: this shows where the time goes, not the product kernel's duration.
    SSF_LIB=opticommpy_amd/libssf_hip_phase.so python tools/phase_timing.py [log2N] [c64]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SSF_LIB", os.path.join(ROOT, "opticommpy_amd", "libssf_hip_phase.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import synth_field  # noqa: E402
from opticommpy_amd import _lib, models  # noqa: E402

ROW = ["entry", "loads+ctrl done", "fwd FFT", "x lin", "inv FFT", "stores done"]
COL = ["entry", "G loads done", "inv FFT", "time-domain work", "fwd FFT", "stores done"]


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    prec = np.complex64 if len(sys.argv) > 2 and sys.argv[2] == "c64" else np.complex128
    N = int(os.environ.get("PHASE_N", 1 << lg))
    E = synth_field(N, 2, 2, 8.4).astype(prec)
    p = oa.parameters()
    for k, v in dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, Ltotal=4.0, Lspan=4.0, hz=0.08, maxIter=10,
                     tol=1e-5, nlprMethod=False, amp="ideal", prgsBar=False, prec=prec, saveSpanN=[]).items():
        setattr(p, k, v)
    models.manakovSSF(E, p)
    print(models.last_run["engine"], models.last_run["steps"], "steps", models.last_run["device_ms"], "ms")
    lib = _lib.load()
    buf = np.zeros((4, 4096, 8), dtype=np.uint64)
    lib.ssf_debug_marks.argtypes = [C.c_void_p]
    rc = lib.ssf_debug_marks(buf.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    for kind, names in ((0, ROW), (1, COL)):
        m = buf[kind].astype(np.int64)
        if kind == 1 and (m[:, 6] > 0).any():             # column kernel: two extra stamps inside the I stage
            m = m[:, [0, 1, 2, 6, 7, 3, 4, 5]]
            names = names[:3] + ["I: E_hd in, powers swapped", "I: phases done"] + names[3:]
        used = m[:, 0] > 0
        m = m[used][:, : len(names)]
        t0 = m[:, 0].min()
        us = (m - t0) / 100.0                 # 100 MHz wall clock
        print(f"{'row' if kind == 0 else 'col'} kernel: {used.sum()} workgroups")
        for i, n in enumerate(names):
            col = us[:, i]
            col = col[m[:, i] > 0]
            if len(col):
                bids = np.nonzero(used)[0][m[:, i] > 0]
                late = bids[np.argsort(col)[-4:]][::-1]
                print(f"   {n:20s} min {col.min():7.2f}  med {np.median(col):7.2f}  p90 {np.percentile(col, 90):7.2f}  p99 {np.percentile(col, 99):7.2f}"
                      f"  max {col.max():7.2f} us   latest workgroups {list(late)}")
        last = us[:, len(names) - 1]
        bids_all = np.nonzero(used)[0]
        print("   end of workgroup by XCD (bid % 8), median / max: " +
              " ".join(f"{np.median(last[bids_all % 8 == x]):.1f}/{last[bids_all % 8 == x].max():.1f}" for x in range(8)))
        q = max(1, len(last) // 4)
        print("   end of workgroup by quarter of the grid, median / max: " +
              " ".join(f"{np.median(last[i * q:(i + 1) * q]):.1f}/{last[i * q:(i + 1) * q].max():.1f}" for i in range(4)))
        G = int(os.environ.get("PHASE_GROUPS", "1"))
        if G > 1:                                         # experiment: statistics per workgroup class
            bids = np.nonzero(used)[0]
            mode = int(os.environ.get("PHASE_GROUP_MODE", "0"))
            grp = bids % G if mode == 0 else (bids // 8) % G if mode == 1 else bids // ((used.sum() + G - 1) // G)
            for gi in range(G):
                sel = grp == gi
                print(f"   group {gi}: " + " | ".join(f"{names[i]} {np.median(us[sel, i]):.2f}" for i in range(len(names))))
        d = np.diff(us, axis=1)
        print("   per-phase median durations:", " | ".join(f"{names[i + 1]} {np.median(d[:, i]):.2f}" for i in range(len(names) - 1)))


if __name__ == "__main__":
    main()
