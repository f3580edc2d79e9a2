#!/bin/bash
# last checks of the final binary: BASELINE configurations at full size through the public Python API, more fuzzing seeds
cd "$(dirname "$0")/.."
O=gpurun_out/r3_final2; mkdir -p $O
timeout 500 python tests/tools/full_configs.py > $O/full_configs.log 2>&1; echo "full configs rc=$?"; tail -12 $O/full_configs.log | cut -c1-220
for seed in 47 53; do timeout 300 python tests/tools/fuzz_gpu.py 250 $seed > $O/fuzz_$seed.log 2>&1; echo "fuzz seed $seed rc=$? $(tail -1 $O/fuzz_$seed.log)"; done
