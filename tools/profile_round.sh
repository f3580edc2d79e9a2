#!/bin/bash
# One call on a GPU box: the bench line, the rocprofv3 kernel-trace summary of the same command and the two PMC passes the
# HBM traffic figure comes from (FETCH_SIZE / WRITE_SIZE separately, never together with other tracing domains).
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh r1e'   ->  gpurun_out/<tag>/{bench.json,kernel_stats.txt,traffic.txt,traffic.json}
# Copy what should be judged into profiles/<tag>_*.
TAG=${1:-prof}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-kernel-times"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- $BENCH --steps 200 --warmup 10 > "$OUT/kt.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pf" -o pf -- $BENCH --steps 200 --warmup 0 > "$OUT/pf.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pw" -o pw -- $BENCH --steps 200 --warmup 0 > "$OUT/pw.log" 2>&1
cd "$REPO"
KT=$(find "$OUT/kt" -name '*.db' | head -1); PF=$(find "$OUT/pf" -name '*.db' | head -1); PW=$(find "$OUT/pw" -name '*.db' | head -1)
python tools/rocpd_stats.py "$KT" > "$OUT/kernel_stats.txt" 2>&1
python tools/traffic_from_pmc.py "$PF" "$PW" 200 fused "$OUT/traffic.json" > "$OUT/traffic.txt" 2>&1
find "$OUT" -name '*.db' -size +40M -delete          # (gpurun_out merges back at most 64 MiB)
tail -1 "$OUT/bench.json" | cut -c1-400; head -6 "$OUT/kernel_stats.txt"; tail -3 "$OUT/traffic.txt"
