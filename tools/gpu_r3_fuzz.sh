#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3_fuzz; mkdir -p $O
timeout 900 python -m pytest tests/test_round3.py -m gpu -q -k "workload_size" > $O/pytest_c45.log 2>&1; echo "c4/c5 tests rc=$?"; tail -2 $O/pytest_c45.log
for seed in 31 37; do timeout 600 python tests/tools/fuzz_gpu.py 350 $seed > $O/fuzz_$seed.log 2>&1; echo "fuzz seed $seed rc=$? $(tail -1 $O/fuzz_$seed.log)"; done
timeout 600 python tests/tools/soak_gpu.py > $O/soak.log 2>&1; echo "soak rc=$? $(tail -1 $O/soak.log)"
