#!/bin/bash
# evidence for the two-lane overlap: kernel trace of config 4 on one GPU, 1 lane vs 2 lanes: sum of kernel durations vs busy wall time
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for l in 1 2; do
  timeout 300 rocprofv3 --kernel-trace -d $O/kt_l$l -o kt -- python $OLDPWD/bench.py --config 4 --steps 50 --warmup 5 --lanes $l --no-cpu-baseline --no-kernel-times > $O/kt_l$l.log 2>&1
done
cd $OLDPWD
python - <<PY > $O/lane_overlap.txt 2>&1
import sqlite3, glob
for l in (1, 2):
    db = glob.glob("$O/kt_l%d/**/*.db" % l, recursive=True)[0]
    con = sqlite3.connect(db); cur = con.cursor()
    rows = cur.execute("select name, start, end from kernels where name like '%k_row%' or name like '%k_col%' order by start").fetchall()
    # the timed region = the last 16 * 50 steps worth of launches: take the second half of the launches
    rows = rows[len(rows) // 2:]
    total = sum(e - s for _, s, e in rows)
    # union of busy intervals
    busy, cs, ce = 0, None, None
    for _, s, e in rows:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: busy += ce - cs; cs, ce = s, e
    busy += ce - cs
    span = rows[-1][2] - rows[0][1]
    ov = sum(1 for i in range(1, len(rows)) if rows[i][1] < rows[i-1][2])
    print("lanes=%d: %d launches, sum of kernel durations %.2f ms, union of busy time %.2f ms, first-to-last %.2f ms, launches that start before the previous one ended: %d"
          % (l, len(rows), total / 1e6, busy / 1e6, span / 1e6, ov))
PY
find $O -name "*.db" -delete
cat $O/lane_overlap.txt
