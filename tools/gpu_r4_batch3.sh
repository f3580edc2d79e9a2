#!/bin/bash
# round 4, third GPU call: whole GPU suite on the final kernels, then the profiles of configs 2 and 3 (profiles/r4_*)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4c
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c/pytest.log
tail -5 gpurun_out/r4c/pytest.log
bash tools/gpu_profiles.sh 2 r4
bash tools/gpu_profiles.sh 3 r4
