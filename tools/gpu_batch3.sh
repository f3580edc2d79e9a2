#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py"
timeout 300 $B --config 3 --steps 200 --warmup 20 > $O/b_c3.json 2> $O/b_c3.err
SSF_BENCH_FORCE_COMM=1 timeout 300 $B --config 4 --steps 50 --warmup 5 > $O/b_c4_rccl1.json 2> $O/b_c4_rccl1.err
# snapshot streaming at config 3's size: 10 spans x 11 steps, every span saved: kernel + memcpy trace
cat > /tmp/snap10.py <<PY
import sys, time, numpy as np
sys.path.insert(0, "$PWD"); sys.path.insert(0, "$PWD/tests")
import opticommpy_amd as oa
from helpers import synth_field, make_param
E = synth_field(1 << 22, 2, 3, 8.4, np.complex64)
cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Lspan=8, Ltotal=80, hz=0.08,
           nlprMethod=False, amp="ideal", prec=np.complex64, saveSpanN=list(range(1, 11)))
for rep in range(2):
    t0 = time.time(); out = oa.manakovSSF(E, make_param(oa.parameters, cfg)); t1 = time.time()
    print("saveSpanN=[1..10], 2^22 c64, 10 x 101 steps: %.3f s wall, device %.1f ms, out %s" % (t1 - t0, oa.last_run["device_ms"], out.shape), flush=True)
cfg["saveSpanN"] = []
t0 = time.time(); out = oa.manakovSSF(E, make_param(oa.parameters, cfg)); t1 = time.time()
print("saveSpanN=[]: %.3f s wall, device %.1f ms" % (t1 - t0, oa.last_run["device_ms"]))
PY
timeout 300 python /tmp/snap10.py > $O/snap10.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OLDPWD/$O/snap_trace -o st -- python /tmp/snap10.py > $OLDPWD/$O/snap_trace.log 2>&1
cd $OLDPWD
python - <<PY > $O/snap_trace_summary.txt 2>&1
import sqlite3, glob
db = glob.glob("$O/snap_trace/**/*.db", recursive=True)
con = sqlite3.connect(db[0]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if "copy" in t.lower() or "kernel" in t.lower()])
for t in tabs:
    if "memory_cop" in t.lower() and "rocpd" not in t.lower()[:0]:
        try:
            cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
            print(t, cols)
        except Exception as e: print(e)
PY
find $O -name "*.db" -size +30M -delete
tail -5 $O/pytest.log; cat $O/snap10.log; tail -c 600 $O/b_c3.json; echo; tail -c 300 $O/b_c4_rccl1.json
