#!/bin/bash
# Profiles of the current binary: bench lines, rocprofv3 kernel-trace summaries, HBM traffic (separate PMC passes), SQ
# counters.  The raw rocpd databases stay under gpurun_out/<tag>_prof/ (gzip'ed), the summaries are copied into profiles/<tag>_*.
#   gpurun --timeout 1500 -- 'bash tools/gpu_profiles.sh 2 r4'      (config 2 | 3, tag)
CFG=${1:-2}; TAG=${2:-r4}
cd "$(dirname "$0")/.."
REPO=$PWD
O=$REPO/gpurun_out/${TAG}_prof/c$CFG; mkdir -p $O
export TMPDIR=/tmp
STEPS=$([ $CFG = 2 ] && echo 200 || echo 100)
python bench.py --config $CFG --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --config $CFG --no-also --steps $((STEPS * 2)) --warmup 30 > $O/bench.json 2> $O/bench.err
cd /tmp
BENCH="python $REPO/bench.py --config $CFG --no-cpu-baseline --no-kernel-times --no-also"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $BENCH --steps $STEPS --warmup 10 > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o pf -- $BENCH --steps $STEPS --warmup 0 > $O/pf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o pw -- $BENCH --steps $STEPS --warmup 0 > $O/pw.log 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/sq$i -o p -- $BENCH --steps 50 --warmup 0 > $O/sq$i.log 2>&1
done
cd $REPO
KT=$(find $O/kt -name '*.db' | head -1); PF=$(find $O/pf -name '*.db' | head -1); PW=$(find $O/pw -name '*.db' | head -1)
python tools/rocpd_stats.py "$KT" > $O/kernel_stats.txt 2>&1
ALG=$(python -c "print({2: 536870912, 3: 1073741824}[$CFG])")
python tools/traffic_from_pmc.py "$PF" "$PW" $STEPS fused $O/traffic.json $CFG $ALG 3.0 "profiles/${TAG}_c${CFG}_traffic.txt" > $O/traffic.txt 2>&1
python tools/rocpd_pmc.py --min-us 12 $(find $O/sq* -name "*.db" | sort) > $O/sq_pmc.txt 2>&1
find $O -name '*.db' -exec gzip -f {} \;
du -sh $O; tail -1 $O/bench.json | cut -c1-260; head -6 $O/kernel_stats.txt | cut -c1-170; tail -3 $O/traffic.txt
