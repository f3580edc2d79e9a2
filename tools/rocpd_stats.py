#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel stats table.
Usage: python tools/rocpd_stats.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':90s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for n, c, s, a, mn, mx in rows:
        n = n if len(n) <= 90 else n[:87] + "..."
        print(f"{n:90s} {c:8d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/total:6.2f}")
    print(f"{'TOTAL':90s} {sum(r[1] for r in rows):8d} {total/1e6:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
