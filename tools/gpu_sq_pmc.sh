#!/bin/bash
# SQ counters (separate passes, kernel-trace only) for configs 2 and 3
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in 2 3; do
  B="python $OLDPWD/bench.py --config $c --steps 50 --warmup 0 --no-cpu-baseline --no-kernel-times"
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/c${c}_p$i -o p -- $B > $O/c${c}_p$i.log 2>&1
  done
done
cd $OLDPWD
for c in 2 3; do python tools/rocpd_pmc.py --min-us 12 $(find $O/c${c}_p* -name "*.db" | sort) > $O/sq_c$c.txt 2>&1; done
find $O -name "*.db" -delete
cat $O/sq_c2.txt | grep -v "^#" | grep -E "k_row|k_col" | head -40; cat $O/sq_c3.txt | grep -v "^#" | grep -E "k_row|k_col" | head -40
