#!/bin/bash
# Receiver / transmitter side of the current binary: call times and the rocprofv3 kernel summary of the same script (profiles/<tag>_rx_tx.txt)
#   gpurun --timeout 900 -- 'bash tools/gpu_rx_profile.sh r5_final'
cd "$(dirname "$0")/.."
REPO=$PWD
TAG=${1:-r5}
O=$REPO/gpurun_out/${TAG}_rx; mkdir -p $O
export TMPDIR=/tmp
python tools/bench_rx_device.py 20 22 --reps 20 > $O/calls.txt 2> $O/calls.err
cd /tmp
for lg in 20 22; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt$lg -o kt -- python $REPO/tools/bench_rx_device.py $lg --reps 5 > $O/kt$lg.log 2>&1
  python $REPO/tools/rocpd_stats.py "$(find $O/kt$lg -name '*.db' | head -1)" > $O/kernel_stats_$lg.txt 2>&1
  rm -rf $O/kt$lg
done
cd $REPO
{ echo "# device-resident calls, tools/bench_rx_device.py (20 calls per figure)"; cat $O/calls.txt
  for lg in 20 22; do echo "# rocprofv3 --kernel-trace --stats of the same script at 2^$lg, 5 repetitions per call"; cut -c1-170 $O/kernel_stats_$lg.txt; done; } > $O/summary.txt
cat $O/summary.txt
