#!/bin/bash
# differential fuzzing + soak of the FINAL binary (new seeds)
cd "$(dirname "$0")/.."
O=gpurun_out/r3_fuzz2; mkdir -p $O
for seed in 41 43; do timeout 400 python tests/tools/fuzz_gpu.py 300 $seed > $O/fuzz_$seed.log 2>&1; echo "fuzz seed $seed rc=$? $(tail -1 $O/fuzz_$seed.log)"; done
timeout 400 python tests/tools/soak_gpu.py > $O/soak.log 2>&1; echo "soak rc=$? $(tail -1 $O/soak.log)"
