#!/bin/bash
# round 4, batch 11: P / Theta by owner thread also in the polarisation-split (double precision) column stage (base) against the
# packed stage only (pt1); config 2, four interleaved repetitions; then the double-precision part of the GPU suite
cd "$(dirname "$0")/.."
O=gpurun_out/r4k; mkdir -p $O
AB_CONFIGS="2" bash tools/gpu_ab.sh r4k/ab pt1 base pt1 base
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log
