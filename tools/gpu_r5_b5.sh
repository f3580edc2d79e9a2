#!/bin/bash
# round 5, batch 5: GPU suite on the fused receiver + receiver / transmitter profile after the fusion
cd "$(dirname "$0")/.."
REPO=$PWD; O=$REPO/gpurun_out/r5_b5; mkdir -p $O
timeout 1900 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|chain vs|config 3, one span" $O/pytest.log | tail -15
python tools/bench_rx_device.py 20 22 > $O/rx_after.txt 2>&1; cat $O/rx_after.txt
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_rx -o kt -- python $REPO/tools/bench_rx_device.py 20 --reps 5 > $O/kt_rx.log 2>&1
python $REPO/tools/rocpd_stats.py "$(find $O/kt_rx -name '*.db' | head -1)" > $O/rx_kernel_stats_after.txt 2>&1
find $O/kt_rx -name '*.db' -delete
head -30 $O/rx_kernel_stats_after.txt | cut -c1-170
