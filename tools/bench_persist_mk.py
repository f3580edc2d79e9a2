#!/usr/bin/env python3
"""Manakov span as ONE persistent launch (k_mk_span, experiment) against the launch sequence, single field, complex128.
    python tools/bench_persist_mk.py [log2N ...]        (env: SSF_PERSIST_MK=<workers>, SSF_PERSIST_XCD=1, SSF_COL_HALF=128)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import make_param, rel_l2, synth_field  # noqa: E402
from opticommpy_amd import models  # noqa: E402


def main():
    steps = int(os.environ.get("STEPS", "200"))
    for lg in [int(s) for s in sys.argv[1:]] or [12, 14, 16]:
        N = 1 << lg
        E = synth_field(N, 2, 2, 8.4)
        cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, amp="ideal", saveSpanN=[],
                   Ltotal=(steps - 0.5) * 0.08, Lspan=(steps - 0.5) * 0.08, hz=0.08, nlprMethod=False)
        try:
            oa.manakovSSF(E, make_param(oa.parameters, cfg))
            out = oa.manakovSSF(E, make_param(oa.parameters, cfg))
            r = models.last_run
            print(f"N=2^{lg}: {r['steps'] / (r['device_ms'] * 1e-3):9.0f} steps/s device ({r['steps']} steps, {r['iterations']} iterations) "
                  f"checksum {np.sum(np.abs(out) ** 2):.12e} proj {abs(np.vdot(np.arange(out.size).reshape(out.shape) % 7 - 3.0, out)):.12e}", flush=True)
        except Exception as e:
            print(f"N=2^{lg}: FAILED {e}", flush=True)
        models.release_plans()


if __name__ == "__main__":
    main()
