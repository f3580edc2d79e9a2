#!/usr/bin/env python3
"""Differential fuzzing on the GPU through the public API: random sizes / fibre / solver parameters, both
engines, traced and untraced runs, against the oracle (field and per-step iteration counts).
Usage (GPU box): python tests/tools/fuzz_gpu.py [cases] [seed] [near]
("near": the fused engine runs every Manakov case once more with tol placed around its own bound of lim_0, traced against untraced:
the recovery of a sparsely stored step-start field, fused_kernels.h ST_RECOVER_A; both precisions.)"""
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
    near = len(sys.argv) > 3 and sys.argv[3] == "near"
    bad = recovered = 0
    for case in range(cases):
        lg = int(rng.integers(8, 15))
        N = 1 << lg
        pick = rng.integers(0, 6)
        if pick == 5:                                               # fused engine, mixed-radix COLUMN stage (round 6): no power-of-two column split
            N = int(rng.choice([9000, 10125, 15625, 16875, 25000, 28125, 40500, 50625, 8 * 3125 * 3,
                                20000, 36000, 100000]))      # (the last three: fewer than seven factors of two -> the two-factor split by rule)
        elif pick == 0:
            N = int(rng.choice([1500, 3000, 6000, 10000, 97, 1009, 1234, 6006, 31]))   # general-length engine (Bluestein on the fused kernels)
        elif pick == 1:                                             # fused engine, mixed-radix rows
            N = int(rng.choice([128 * 75, 128 * 81, 128 * 125, 256 * 45, 512 * 27, 1024 * 15, 128 * 225, 256 * 135,
                                48000, 128 * 405, 512 * 125, 1024 * 75]))
        K = int(rng.choice([1, 1, 1, 2, 3]))
        c64 = bool(rng.integers(0, 4) == 0) and pick >= 2        # complex64: packed polarisation pairs (power-of-two lengths); one row per polarisation on the mixed-radix column stage
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
            if ok and near and eng == "fused" and func != "ssfm" and cfg["maxIter"] > 1 and cfg["gamma"] > 0 and len(run.get("lims", [])):
                l0 = np.array([float(l[0]) for l in run["lims"] if len(l)])
                l0 = l0[np.isfinite(l0) & (l0 > 0)]
                if len(l0):
                    cfg2 = dict(cfg, tol=float(rng.choice(l0)) * float(rng.uniform(0.2, 0.34)))
                    a = FUNCS[func](E, make_param(oa.parameters, cfg2), _trace=True)
                    ra = dict(models.last_run)
                    b = FUNCS[func](E, make_param(oa.parameters, cfg2))
                    rb = dict(models.last_run)
                    recovered += rb["recovered_fields"]
                    ok2 = np.array_equal(a, b) and (ra["steps"], ra["iterations"]) == (rb["steps"], rb["iterations"]) and ra["recovered_fields"] == 0
                    if not c64:
                        tr2 = {}
                        with np.errstate(all="ignore"):
                            ref2 = ORC[func](E, make_param(orc.parameters, cfg2), trace=tr2)
                        ok2 = ok2 and list(ra["iters"]) == tr2["iters"] and rel_l2(a, ref2) <= gate
                    if not ok2:
                        bad += 1
                        print("MISMATCH (tol near the bound) case", case, cfg2, "N", N, "K", K, "p", p_dbm, "equal", np.array_equal(a, b),
                              (ra["steps"], ra["iterations"]), (rb["steps"], rb["iterations"]), "recovered", rb["recovered_fields"], flush=True)
        if case % 25 == 0:
            print(f"case {case} done ({func}, N={N}, K={K})", flush=True)
    oa.set_engine("auto")
    print("done:", cases, "cases,", bad, "mismatches" + (f", {recovered} recovered fields in the near-the-bound runs" if near else ""))


if __name__ == "__main__":
    main()
