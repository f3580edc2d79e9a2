#!/usr/bin/env python3
"""Per-kernel HIP-event times (ssf_set_profiling) of manakovSSF at one length: which pipeline the plan runs on, steps/s, and the
average row / column launch.  Usage (on a GPU box):  python tools/profile_length.py N [c64] [adaptive]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import make_param, synth_field  # noqa: E402
from opticommpy_amd import _lib, models  # noqa: E402


def main():
    N = int(sys.argv[1])
    c64 = "c64" in sys.argv[2:]
    adaptive = "adaptive" in sys.argv[2:]
    dt = np.complex64 if c64 else np.complex128
    E = synth_field(N, 2, 2, 8.4).astype(dt)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, amp="ideal", saveSpanN=[],
               Ltotal=15.96, Lspan=15.96, hz=0.08, nlprMethod=adaptive, maxNlinPhaseRot=2e-3, prec=dt)
    oa.manakovSSF(E, make_param(oa.parameters, cfg))
    pl = models._get_plan(N, 2, _lib.SSF_C64 if c64 else _lib.SSF_C128)
    pl.lib.ssf_set_profiling(pl.h, 1)
    oa.manakovSSF(E, make_param(oa.parameters, cfg))
    r = dict(models.last_run)
    kt = _lib.KernelTimes()
    pl.lib.ssf_get_kernel_times(pl.h, C.byref(kt))
    pl.lib.ssf_set_profiling(pl.h, 0)
    oa.manakovSSF(E, make_param(oa.parameters, cfg))
    r2 = models.last_run
    s = 8 if c64 else 16
    per = 2 * s * N * 2
    print(f"N={N} {dt.__name__} pipeline={r['pipeline']} steps={r['steps']} it/step={r['iterations'] / r['steps']:.2f} "
          f"steps/s={r2['steps'] / (r2['device_ms'] * 1e-3):.0f} (profiled run: {r['steps'] / (r['device_ms'] * 1e-3):.0f})")
    for name, ms, n in (("row", kt.row_ms, kt.row_n), ("col", kt.col_ms, kt.col_n), ("other", kt.other_ms, kt.other_n)):
        if n:
            us = ms / n * 1e3
            print(f"  {name:5s}: {n:6d} launches, {us:8.2f} us each, {per / us / 1e6:7.3f} TB/s algorithmic = {per / us / 1e6 / 8:.3f} of peak")


if __name__ == "__main__":
    main()
