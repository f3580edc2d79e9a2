#!/usr/bin/env python3
"""HBM traffic per SSFM step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: separate --pmc passes; on gfx950
FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced reads => x2; WRITE_SIZE is
taken as is.  Both corrections are re-checked in the same run on k_amp (a plain read-modify-write of
a known 32 MiB / 64 MiB buffer).  Units of both counters: KiB.

usage: traffic_from_pmc.py fetch.db write.db TOTAL_UNIT_STEPS ENGINE [out.json CONFIG ALGORITHMIC_BYTES_PER_STEP ITER_PER_STEP SOURCE]"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    q = ("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name")
    return {n: (c, v) for n, c, v in db.execute(q, (counter,))}


def main(fetch_db, write_db, total_steps, engine, out=None, config="2", alg_bytes=536870912, it_step=3.0, source=""):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    total = 0.0
    rows = []
    for name in sorted(set(f) | set(w)):
        if any(s in name for s in ("k_amp", "copyBuffer", "fillBuffer", "k_aos", "k_soa", "k_burst_copy")):
            cal = (f.get(name, (0, 0)), w.get(name, (0, 0)))
            rows.append((name.split("(")[0][-40:], "calibration/aux", cal))
            continue
        rd = 2.0 * f.get(name, (0, 0))[1] * 1024.0
        wr = w.get(name, (0, 0))[1] * 1024.0
        total += rd + wr
        rows.append((name[:70], f.get(name, (0, 0))[0], rd, wr))
    per_step = total / float(total_steps)
    for r in rows:
        print(r)
    print(f"HBM-side bytes per step ({engine}): {per_step:.4g}  ({per_step/2**20:.1f} MiB) over {total_steps} steps")
    if out:
        try:
            d = json.load(open(out))
        except Exception:
            d = {}
        # bench.py scales this to its own run: traffic is linear in (1 + iterations per step); the profiled
        # run (fixed step, 8.4 dBm, first 200 steps of config 2) needs 3 iterations in every step
        d["config%s" % config] = {"bytes_per_step": per_step, "iterations_per_step": float(it_step), "steps": total_steps,
                                  "algorithmic_bytes_per_step": int(alg_bytes), "engine": engine, "source": source}
        json.dump(d, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], *sys.argv[5:])
