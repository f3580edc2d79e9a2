#!/bin/bash
# One call on a GPU box: the bench line, the rocprofv3 kernel-trace summary of the same command and the two PMC passes the
# HBM traffic figure comes from (FETCH_SIZE / WRITE_SIZE separately, never together with other tracing domains).
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh r2 2'   ->  gpurun_out/<tag>_c<config>/{bench.json,kernel_stats.txt,traffic.txt,traffic.json}
# Copy what should be judged into profiles/<tag>_c<config>_*.
TAG=${1:-prof}; CFG=${2:-2}; STEPS=${3:-200}
REPO=$PWD
OUT=$REPO/gpurun_out/${TAG}_c$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python bench.py --config $CFG --no-also --steps $STEPS --warmup 20 > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp
BENCH="python $REPO/bench.py --config $CFG --no-cpu-baseline --no-kernel-times --no-also"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- $BENCH --steps $STEPS --warmup 10 > "$OUT/kt.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pf" -o pf -- $BENCH --steps $STEPS --warmup 0 > "$OUT/pf.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pw" -o pw -- $BENCH --steps $STEPS --warmup 0 > "$OUT/pw.log" 2>&1
cd "$REPO"
KT=$(find "$OUT/kt" -name '*.db' | head -1); PF=$(find "$OUT/pf" -name '*.db' | head -1); PW=$(find "$OUT/pw" -name '*.db' | head -1)
python tools/rocpd_stats.py "$KT" > "$OUT/kernel_stats.txt" 2>&1
UNITS=$(python -c "print({4: 16, 5: 8}.get($CFG, 1))")
ALG=$(python -c "print({2: 536870912, 3: 1073741824, 4: 536870912, 5: 1073741824}.get($CFG, 536870912))")
python tools/traffic_from_pmc.py "$PF" "$PW" $((STEPS * UNITS)) fused "$OUT/traffic.json" $CFG $ALG 3.0 "$TAG" > "$OUT/traffic.txt" 2>&1
find "$OUT" -name '*.db' -exec gzip -f {} \;          # keep the raw databases (gpurun_out merges back at most 64 MiB: they are a few MiB)
tail -1 "$OUT/bench.json" | cut -c1-300; head -6 "$OUT/kernel_stats.txt" | cut -c1-160; tail -3 "$OUT/traffic.txt"
