#!/usr/bin/env python3
"""Throughput of manakovSSF at sequence lengths that are not powers of two (the lengths the reference's
notebooks use: 2^a 3^b 5^c), fused engine against the rocFFT engine.  Usage (on a GPU box):
    python tools/bench_lengths.py [N ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import make_param, synth_field  # noqa: E402
from opticommpy_amd import models  # noqa: E402


def main():
    sizes = [int(s) for s in sys.argv[1:]] or [48000, 240000, 960000, 1 << 20, 3 << 18, 5 << 18, 1440000]
    for N in sizes:
        E = synth_field(N, 2, 2, 8.4)
        cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False,
                   amp="ideal", saveSpanN=[], Ltotal=15.96, Lspan=15.96, hz=0.08, nlprMethod=False)
        line = f"N={N:8d}"
        outs = {}
        rate = {}
        for eng in ("auto", "fused", "rocfft"):                  # auto: the product's choice; fused: hand-written kernels forced
            oa.set_engine(eng)
            oa.manakovSSF(E, make_param(oa.parameters, cfg))
            outs[eng] = oa.manakovSSF(E, make_param(oa.parameters, cfg))
            r = models.last_run
            rate[eng] = r['steps'] / (r['device_ms'] * 1e-3)
            line += f"  {eng}->{r['engine']:>6s}: {rate[eng]:8.0f} steps/s"
        d = np.linalg.norm(outs["fused"] - outs["rocfft"]) / np.linalg.norm(outs["rocfft"])
        ok = rate["auto"] >= 0.97 * max(rate["fused"], rate["rocfft"])
        print(line + f"  engines differ by {d:.1e}  auto is the faster one: {ok}", flush=True)


if __name__ == "__main__":
    main()
