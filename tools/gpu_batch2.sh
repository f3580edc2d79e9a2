#!/bin/bash
# round-2 GPU batch 2: full GPU test suite (long runs, RCCL single rank, packed c64), bench configurations, packed geometry sweep
cd "$(dirname "$0")/.."
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py"
timeout 300 $B --steps 200 --warmup 20 > $O/b_c2.json 2> $O/b_c2.err
timeout 300 $B --config 1 --steps 1000 --warmup 100 > $O/b_c1.json 2> $O/b_c1.err
timeout 300 $B --config 3 --steps 200 --warmup 20 > $O/b_c3.json 2> $O/b_c3.err
timeout 300 $B --config 4 --steps 100 --warmup 10 > $O/b_c4.json 2> $O/b_c4.err
timeout 300 $B --config 4 --steps 100 --warmup 10 --lanes 1 > $O/b_c4_lane1.json 2> $O/b_c4_lane1.err
timeout 300 $B --config 5 --steps 100 --warmup 10 > $O/b_c5.json 2> $O/b_c5.err
timeout 300 $B --config 5 --steps 100 --warmup 10 --dbp-hz 10 > $O/b_c5_hz10.json 2> $O/b_c5_hz10.err
SSF_BENCH_FORCE_COMM=1 timeout 300 $B --config 4 --steps 50 --warmup 5 > $O/b_c4_rccl1.json 2> $O/b_c4_rccl1.err
# the driver's own command lines
timeout 300 $B --gpus 1 --steps 20 --warmup 5 > $O/b_driver.json 2> $O/b_driver.err
# packed geometry sweep (config 3 shape), per-kernel times
Q="python bench.py --no-cpu-baseline --config 3 --steps 100 --warmup 10"
for l1 in 9 10 11; do SSF_SPLIT_L1=$l1 timeout 200 $Q > $O/c3_l1_$l1.json 2> $O/c3_l1_$l1.err; done
SSF_SPLIT_L1=10 SSF_COL_HALF=256 timeout 200 $Q > $O/c3_l1_10_h256.json 2>&1
SSF_SPLIT_L1=11 SSF_ROW_FPW=1 timeout 200 $Q > $O/c3_l1_11_fpw1.json 2>&1
# under-filled sizes: one row per workgroup
R="python bench.py --no-cpu-baseline --no-kernel-times --steps 400 --warmup 20"
for lg in 18 19; do timeout 100 $R --log2n $lg > $O/c128_$lg.json 2>&1; SSF_ROW_FPW=1 timeout 100 $R --log2n $lg > $O/c128_${lg}_fpw1.json 2>&1; done
timeout 100 $R --log2n 20 --prec c64 > $O/c64_20.json 2>&1; SSF_ROW_FPW=1 timeout 100 $R --log2n 20 --prec c64 > $O/c64_20_fpw1.json 2>&1
SSF_SPLIT_L1=9 timeout 100 $R --log2n 20 --prec c64 > $O/c64_20_l1_9.json 2>&1
tail -3 $O/pytest.log
for f in $O/*.json; do echo "$f: $(python - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); k=d["roofline"].get("kernels") or {}
    print(d["value"] and round(d["value"],1), d["unit"], "frac", round(d["roofline"]["frac"],4), "it/step", round(d["config"]["iterations_per_step"],3),
          "row", k.get("row",{}).get("avg_us") and round(k["row"]["avg_us"],1), "col", k.get("col",{}).get("avg_us") and round(k["col"]["avg_us"],1),
          "parity", d.get("parity",{}).get("rel_l2_vs_oracle"), "cpu", d.get("cpu_baseline",{}).get("value"), d.get("comm","")[:20])
except Exception as e: print("ERR", e, open("$f").read()[-300:])
PY
)"; done
