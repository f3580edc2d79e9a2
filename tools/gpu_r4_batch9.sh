#!/bin/bash
# round 4, batch 9: final stage stores one sample in sixteen (sparse) against the build before it (prev), then the GPU suite
cd "$(dirname "$0")/.."
O=gpurun_out/r4i; mkdir -p $O
bash tools/gpu_ab.sh r4i/ab prev base
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1
tail -15 $O/pytest.log
