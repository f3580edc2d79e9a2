#!/usr/bin/env python3
"""Differential fuzzing of the fused engine's kernels and state machine on the CPU emulator against the
oracle: random sizes, fibre / solver parameters, step modes, amplifier modes, polarisation-pair counts,
traced and untraced (lim_0 bound) runs.  Usage: python tests/tools/fuzz_emu.py [cases] [seed] [near]
("near": every Manakov case is run a second time with tol placed around its own bound of lim_0 -- between a fifth and a third of a
measured lim_0 -- so that some steps need the exact lim_0 after steps that stored the field sparsely: the recovery path,
fused_kernels.h ST_RECOVER_A.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu_binding as eb  # noqa: E402
from helpers import make_param, rel_l2, synth_field  # noqa: E402
from oracle import ssf_oracle as orc  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    near = len(sys.argv) > 3 and sys.argv[3] == "near"
    bad = recovered = 0
    for case in range(cases):
        lg = int(rng.integers(8, 13))
        N = 1 << lg
        if rng.integers(0, 4) == 0:                                 # mixed-radix rows (2^a 3^b 5^c)
            N = int(rng.choice([128 * 75, 128 * 81, 256 * 75, 128 * 90, 256 * 100, 128 * 125]))
        K = int(rng.choice([1, 1, 1, 2, 3]))
        func = str(rng.choice(["manakovSSF", "manakovSSF", "manakovDBP", "ssfm"]))
        p_dbm = float(rng.choice([-20, -5, 0, 6, 10, 14]))
        adaptive = bool(rng.integers(0, 2)) and func != "ssfm"
        hz = float(rng.choice([0.05, 0.1, 0.37, 0.5, 1.0, 2.5]))
        Lspan = float(rng.choice([0.4, 1.0, 2.0, 3.3, 5.0]))
        nsp = int(rng.integers(1, 4))
        cfg = dict(func=func, alpha=float(rng.choice([0.0, 0.2, 0.5])), D=float(rng.choice([1e-3, 4, 16, 17])),
                   gamma=float(rng.choice([0.0, 1e-3, 1.3, 2.0])), Fc=193.1e12, Fs=float(rng.choice([64e9, 256e9, 512e9])),
                   maxIter=int(rng.choice([1, 2, 3, 10])), tol=float(rng.choice([1e-3, 1e-5, 1e-7, 1e-9])), prgsBar=False,
                   Ltotal=Lspan * nsp + float(rng.choice([0.0, 0.3])), Lspan=Lspan, hz=hz, nlprMethod=adaptive,
                   maxNlinPhaseRot=float(rng.choice([5e-3, 2e-2, 1e-1])), amp=rng.choice(["ideal", None, "ideal"]),
                   saveSpanN=[])
        if func == "ssfm":
            E = synth_field(N, 1, case, p_dbm).reshape(-1) * np.sqrt(2)
            for k in ("maxIter", "tol", "nlprMethod", "maxNlinPhaseRot"):
                cfg.pop(k)
        else:
            E = synth_field(N, 2 * K, case, p_dbm)
        tr = {}
        fn = {"ssfm": orc.ssfm, "manakovSSF": orc.manakovSSF, "manakovDBP": orc.manakovDBP}[func]
        with np.errstate(all="ignore"):
            ref = fn(E, make_param(orc.parameters, cfg), trace=tr)
        out, info = eb.run(func, E, cfg, max_steps=1 << 15)
        out2, info2 = eb.run(func, E, cfg, max_steps=1 << 15, trace=False)
        got = out.T.reshape(ref.shape) if func != "ssfm" else out.reshape(ref.shape)
        ok = np.all(np.isfinite(ref)) and rel_l2(got, ref) <= 1e-9
        if func != "ssfm":
            ok = ok and list(info["iters"]) == tr["iters"] and info["nonconverged_steps"] == tr.get("nonconverged", info["nonconverged_steps"])
            ok = ok and np.array_equal(out, out2) and info2["iterations"] == info["iterations"]
        if not ok:
            bad += 1
            print("MISMATCH case", case, cfg, "N", N, "K", K, "p", p_dbm, "rel", rel_l2(got, ref) if np.all(np.isfinite(ref)) else "nan-ref",
                  "iters", list(info.get("iters", []))[:8], tr.get("iters", [])[:8], flush=True)
        if ok and near and func != "ssfm" and cfg["maxIter"] > 1 and len(info["lims"]) and cfg["gamma"] > 0:
            l0 = np.array([float(l[0]) for l in info["lims"] if len(l)])
            l0 = l0[np.isfinite(l0) & (l0 > 0)]
            if len(l0):
                cfg2 = dict(cfg, tol=float(rng.choice(l0)) * float(rng.uniform(0.2, 0.34)))
                tr2 = {}
                with np.errstate(all="ignore"):
                    ref2 = fn(E, make_param(orc.parameters, cfg2), trace=tr2)
                a, ia = eb.run(func, E, cfg2, max_steps=1 << 15)
                b, ib = eb.run(func, E, cfg2, max_steps=1 << 15, trace=False)
                ok2 = (np.array_equal(a, b) and ib["iterations"] == ia["iterations"] and list(ia["iters"]) == tr2["iters"]
                       and rel_l2(a.T.reshape(ref2.shape), ref2) <= 1e-9 and ia["recovered_fields"] == 0)
                recovered += ib["recovered_fields"]
                if not ok2:
                    bad += 1
                    print("MISMATCH (tol near the bound) case", case, cfg2, "N", N, "K", K, "p", p_dbm, "equal", np.array_equal(a, b),
                          "iterations", ia["iterations"], ib["iterations"], "recovered", ib["recovered_fields"], flush=True)
        if ok and case % 20 == 0:
            print(f"case {case} ok ({func}, N={N}, K={K}, steps={info['steps']}, it={info['iterations']}, rebuilt={info['rebuilt_iterates']}/{info2['rebuilt_iterates']})", flush=True)
    print("done:", cases, "cases,", bad, "mismatches" + (f", {recovered} recovered fields in the near-the-bound runs" if near else ""))


if __name__ == "__main__":
    main()
