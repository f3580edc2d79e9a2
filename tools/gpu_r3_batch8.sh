#!/bin/bash
# round 3, GPU call 8: are the 8-values-per-thread kernels the better ones where a launch is a latency chain (small fields)?
cd "$(dirname "$0")/.."
O=gpurun_out/r3h; mkdir -p $O
for v in "SSF_X=0" "SSF_ROW_V=8" "SSF_COL_V=8" "SSF_ROW_V=8 SSF_COL_V=8"; do
  echo "== $v"
  env $v UNITS=4 python tools/bench_units.py 10 12 14 16 18 2>&1 | grep -E "N=2|config 1" | sed -e 's/bit-equal.*//' 
done > $O/small_v8.txt 2>&1
cat $O/small_v8.txt
