#!/bin/bash
# SQ counters of the kernels a given length runs on (tools/profile_length.py N as the workload), four separate rocprofv3 --pmc passes.
#   gpurun --timeout 900 -- 'bash tools/gpu_sq_length.sh 960000 r6_sq_960000'
cd "$(dirname "$0")/.."
REPO=$PWD; N=${1:-960000}; O=$REPO/gpurun_out/${2:-sq_$N}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/sq$i -o p -- python $REPO/tools/profile_length.py $N > $O/sq$i.log 2>&1
done
cd $REPO
python tools/rocpd_pmc.py --min-us 12 $(find $O/sq* -name "*.db" | sort) > $O/sq_pmc.txt 2>&1
find $O -name '*.db' -delete
grep -E "k_row_mixed|k_col_ragged|k_col_mixed|k_row<|k_col<" $O/sq_pmc.txt | cut -c1-200
