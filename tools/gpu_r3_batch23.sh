#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3w; mkdir -p $O
for i in 1 2 3; do python tools/exp/first_process_timeline.py 0 150 2>&1 | tail -8; done | tee $O/timeline.txt
python tools/exp/first_process_timeline.py 20 150 2>&1 | tail -8 | tee -a $O/timeline.txt
python tools/exp/first_process_timeline.py 20 150 2>&1 | tail -8 | tee -a $O/timeline.txt
