#!/usr/bin/env python3
"""Differential fuzzing on the GPU through the public API: random sizes / fibre / solver parameters, both
engines, traced and untraced runs, against the oracle (field and per-step iteration counts).
Usage (GPU box): python tests/tools/fuzz_gpu.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import make_param, rel_l2, synth_field  # noqa: E402
from opticommpy_amd import models  # noqa: E402
from oracle import ssf_oracle as orc  # noqa: E402

FUNCS = {"ssfm": oa.ssfm, "manakovSSF": oa.manakovSSF, "manakovDBP": oa.manakovDBP}
ORC = {"ssfm": orc.ssfm, "manakovSSF": orc.manakovSSF, "manakovDBP": orc.manakovDBP}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(cases):
        lg = int(rng.integers(8, 15))
        N = 1 << lg
        pick = rng.integers(0, 5)
        if pick == 0:
            N = int(rng.choice([1500, 3000, 6000, 10000, 97, 1009, 1234, 6006, 31]))   # general-length engine (Bluestein on the fused kernels)
        elif pick == 1:                                             # fused engine, mixed-radix rows
            N = int(rng.choice([128 * 75, 128 * 81, 128 * 125, 256 * 45, 512 * 27, 1024 * 15, 128 * 225, 256 * 135,
                                48000, 128 * 405, 512 * 125, 1024 * 75]))
        K = int(rng.choice([1, 1, 1, 2, 3]))
        c64 = bool(rng.integers(0, 4) == 0) and pick >= 2        # complex64: packed polarisation pairs (power-of-two lengths)
        func = str(rng.choice(["manakovSSF", "manakovSSF", "manakovDBP", "ssfm"]))
        p_dbm = float(rng.choice([-20, -5, 0, 6, 10, 14]))
        adaptive = bool(rng.integers(0, 2)) and func != "ssfm"
        Lspan = float(rng.choice([0.4, 1.0, 2.0, 3.3, 5.0]))
        nsp = int(rng.integers(1, 4))
        cfg = dict(func=func, alpha=float(rng.choice([0.0, 0.2, 0.5])), D=float(rng.choice([1e-3, 4, 16, 17])),
                   gamma=float(rng.choice([0.0, 1e-3, 1.3, 2.0])), Fc=193.1e12, Fs=float(rng.choice([64e9, 256e9, 512e9])),
                   maxIter=int(rng.choice([1, 2, 3, 10])), tol=float(rng.choice([1e-3, 1e-5, 1e-7, 1e-9])), prgsBar=False,
                   Ltotal=Lspan * nsp + float(rng.choice([0.0, 0.3])), Lspan=Lspan, hz=float(rng.choice([0.05, 0.1, 0.37, 0.5, 1.0, 2.5])),
                   nlprMethod=adaptive, maxNlinPhaseRot=float(rng.choice([5e-3, 2e-2, 1e-1])),
                   amp=rng.choice(["ideal", None, "ideal"]), saveSpanN=[])
        if func == "ssfm":
            E = synth_field(N, 1, case, p_dbm).reshape(-1) * np.sqrt(2)
            for k in ("maxIter", "tol", "nlprMethod", "maxNlinPhaseRot"):
                cfg.pop(k)
        else:
            E = synth_field(N, 2 * K, case, p_dbm)
        tr = {}
        gate = 1e-9
        if c64:                                                   # single-precision run against the double-precision oracle
            gate = 5e-4
            with np.errstate(all="ignore"):
                ref = ORC[func](E, make_param(orc.parameters, cfg), trace=tr)
            cfg = dict(cfg, prec="complex64")
            E = E.astype(np.complex64)
        else:
            with np.errstate(all="ignore"):
                ref = ORC[func](E, make_param(orc.parameters, cfg), trace=tr)
        engines = ["rocfft"] + (["fused"] if models.engine_supported("fused", N) else [])
        for eng in engines:
            oa.set_engine(eng)
            out = FUNCS[func](E, make_param(oa.parameters, cfg), _trace=True)
            run = dict(models.last_run)
            out2 = FUNCS[func](E, make_param(oa.parameters, cfg))
            run2 = dict(models.last_run)
            ok = np.all(np.isfinite(ref)) and rel_l2(out, ref) <= gate and np.array_equal(out, out2) if eng == "fused" else rel_l2(out, ref) <= max(gate, 2e-3 if c64 else 0) and rel_l2(out2, ref) <= max(gate, 2e-3 if c64 else 0)
            if func != "ssfm" and not c64:
                ok = ok and list(run["iters"]) == tr["iters"] and run2["iterations"] == run["iterations"]
            if not ok:
                bad += 1
                print("MISMATCH case", case, eng, cfg, "N", N, "K", K, "p", p_dbm, "rel", rel_l2(out, ref), "untraced equal", np.array_equal(out, out2),
                      list(run.get("iters", []))[:8], tr.get("iters", [])[:8], flush=True)
        if case % 25 == 0:
            print(f"case {case} done ({func}, N={N}, K={K})", flush=True)
    oa.set_engine("auto")
    print("done:", cases, "cases,", bad, "mismatches")


if __name__ == "__main__":
    main()
