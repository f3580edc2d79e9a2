#!/bin/bash
# rocprofv3 kernel-trace summaries of the current binary, configs 2 and 3 (the kernel-trace part of tools/gpu_profiles.sh):
#   gpurun --timeout 900 -- 'bash tools/gpu_kernel_stats.sh r5_final'    -> gpurun_out/<tag>_kt/c{2,3}_kernel_stats.txt
cd "$(dirname "$0")/.."
REPO=$PWD; TAG=${1:-r5}; O=$REPO/gpurun_out/${TAG}_kt; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for CFG in 2 3; do
  STEPS=$([ $CFG = 2 ] && echo 200 || echo 100)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt$CFG -o kt -- python $REPO/bench.py --config $CFG --no-cpu-baseline --no-kernel-times --no-also --steps $STEPS --warmup 10 > $O/kt$CFG.log 2>&1
  python $REPO/tools/rocpd_stats.py "$(find $O/kt$CFG -name '*.db' | head -1)" > $O/c${CFG}_kernel_stats.txt 2>&1
  rm -rf $O/kt$CFG
  head -8 $O/c${CFG}_kernel_stats.txt | cut -c1-170
done
