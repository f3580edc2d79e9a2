#!/bin/bash
# round 5, batch 1: interleaved LDS columns A/B + LDS counters of both layouts
cd "$(dirname "$0")/.."
REPO=$PWD
bash tools/gpu_ab_env.sh r5_b1_il SSF_COL_IL 0 1
export TMPDIR=/tmp
cd /tmp
export SSF_LIB=$REPO/opticommpy_amd/libssf_hip_exp.so
for c in 2 3; do for il in 0 1; do
  O=$REPO/gpurun_out/r5_b1_il/pmc_c${c}_il$il; mkdir -p $O
  SSF_COL_IL=$il timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O -o p -- python $REPO/bench.py --config $c --no-cpu-baseline --no-kernel-times --no-also --parity none --steps 50 --warmup 0 > $O/log.txt 2>&1
  python $REPO/tools/rocpd_pmc.py --min-us 12 $(find $O -name "*.db") > $REPO/gpurun_out/r5_b1_il/pmc_c${c}_il$il.txt 2>&1
  find $O -name '*.db' -delete
done; done
cd $REPO; tail -n 12 gpurun_out/r5_b1_il/pmc_c*_il*.txt
