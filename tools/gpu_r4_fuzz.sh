#!/bin/bash
# differential fuzzing + soak + full-size configurations of the current binary (gpurun_out/r4d)
cd "$(dirname "$0")/.."
O=gpurun_out/r4d; mkdir -p $O
timeout 500 python tests/tools/fuzz_gpu.py 300 61 > $O/fuzz_61.log 2>&1; echo "rc=$?" >> $O/fuzz_61.log; tail -2 $O/fuzz_61.log
timeout 500 python tests/tools/fuzz_gpu.py 300 67 > $O/fuzz_67.log 2>&1; echo "rc=$?" >> $O/fuzz_67.log; tail -2 $O/fuzz_67.log
timeout 300 python tests/tools/soak_gpu.py 20 > $O/soak.log 2>&1; echo "rc=$?" >> $O/soak.log; tail -2 $O/soak.log
timeout 400 python tests/tools/full_configs.py > $O/full_configs.log 2>&1; echo "rc=$?" >> $O/full_configs.log; tail -8 $O/full_configs.log
