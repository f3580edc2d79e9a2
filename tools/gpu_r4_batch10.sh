#!/bin/bash
# round 4, batch 10: P / Theta of the packed column stage laid out by owner thread (base) against by sample (ptold); config 3
cd "$(dirname "$0")/.."
O=gpurun_out/r4j; mkdir -p $O
AB_CONFIGS="3" bash tools/gpu_ab.sh r4j/ab ptold base ptold base
timeout 600 python -m pytest tests -m gpu -x -q -k "c64 or complex64 or config3 or packed or units or drift" > $O/pytest_c64.log 2>&1
tail -4 $O/pytest_c64.log
