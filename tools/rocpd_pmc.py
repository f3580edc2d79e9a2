#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in one or more rocprofv3 rocpd databases.
Only dispatches longer than --min-us are averaged (drops the idle launches of the state machine).
Usage: python tools/rocpd_pmc.py [--min-us 8] a.db b.db ..."""
import sqlite3
import sys


def main(argv):
    min_us = 8.0
    if argv and argv[0] == "--min-us":
        min_us = float(argv[1])
        argv = argv[2:]
    for path in argv:
        db = sqlite3.connect(path)
        cur = db.cursor()
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(end-start) from counters_collection "
             "where (end-start) > ? group by kernel_name, counter_name order by kernel_name, counter_name")
        print(f"# {path}  (dispatches longer than {min_us} us)")
        for name, ctr, n, avg, dur in cur.execute(q, (min_us * 1e3,)):
            short = name.replace("void ssf::(anonymous namespace)::", "").replace("ssf::(anonymous namespace)::", "").replace("ssf::fused::", "")
            short = short.split("(")[0]
            print(f"{short:42s} {ctr:24s} n={n:5d} avg={avg:14.4f} avg_dur_us={dur/1e3:8.2f}")


if __name__ == "__main__":
    main(sys.argv[1:])
