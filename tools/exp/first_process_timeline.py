#!/usr/bin/env python3
"""Does a process's first tens of milliseconds of GPU work contain a stall (power / memory-clock state change)?  One plan at
config-2 size, chunks of 5 steps back to back, device time and wall time of every chunk.  Usage (GPU box):
    python tools/exp/first_process_timeline.py [idle seconds before the first chunk] [chunks]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from helpers import synth_field  # noqa: E402
from opticommpy_amd import _lib  # noqa: E402


def main():
    idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    lib = _lib.load()
    N = 1 << 20
    w = bench.workload(2, 0, "", 1)
    E = np.ascontiguousarray(synth_field(N, 2, 2, 8.4).T)
    h = C.c_void_p()
    _lib.raise_for(lib, None, lib.ssf_plan_create(0, N, 2, _lib.SSF_C128, 0, C.byref(h)))
    _lib.raise_for(lib, h, lib.ssf_upload(h, E.ctypes.data_as(C.c_void_p)))
    time.sleep(idle)
    cp = bench.make_params(_lib, w, 5)
    rows = []
    t_start = time.perf_counter()
    for i in range(nchunks):
        st = _lib.Stats()
        t0 = time.perf_counter()
        _lib.raise_for(lib, h, lib.ssf_execute(h, C.byref(cp), 1, 1, None, C.byref(st), None))
        t1 = time.perf_counter()
        rows.append((t0 - t_start, t1 - t0, st.device_ms))
    slow = [(i, r) for i, r in enumerate(rows) if r[1] > 3 * np.median([x[1] for x in rows])]
    med = np.median([x[1] for x in rows])
    print("idle %.0f s before the first chunk; %d chunks of 5 steps; median chunk %.3f ms wall" % (idle, nchunks, med * 1e3))
    print("first 12 chunks (ms wall): " + " ".join("%.2f" % (r[1] * 1e3) for r in rows[:12]))
    for i, r in slow:
        print("  chunk %3d at t = %7.1f ms: %.2f ms wall, %.2f ms device" % (i, r[0] * 1e3, r[1] * 1e3, r[2]))
    if not slow:
        print("  no chunk above 3 x the median")
    lib.ssf_plan_destroy(h)


if __name__ == "__main__":
    main()
