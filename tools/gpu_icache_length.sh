#!/bin/bash
# Instruction-cache counters of the kernels a given length runs on (tools/profile_length.py N as the workload), separate rocprofv3 --pmc passes.
#   gpurun --timeout 900 -- 'bash tools/gpu_icache_length.sh 2000000 r6_icache_2000000'
cd "$(dirname "$0")/.."
REPO=$PWD; N=${1:-2000000}; O=$REPO/gpurun_out/${2:-icache_$N}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/sq$i -o p -- python $REPO/tools/profile_length.py $N > $O/sq$i.log 2>&1
done
cd $REPO
python tools/rocpd_pmc.py --min-us 12 $(find $O/sq* -name "*.db" | sort) > $O/sq_pmc.txt 2>&1
find $O -name '*.db' -delete
grep -E "k_row_mixed|k_col_ragged|k_col_mixed|k_row<|k_col<" $O/sq_pmc.txt | cut -c1-200
