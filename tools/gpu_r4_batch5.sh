#!/bin/bash
# round 4, fifth GPU call: A/B of "the final stage does not rebuild the column spectrum" against the previous build, long-run parity tests
cd "$(dirname "$0")/.."
O=gpurun_out/r4f; mkdir -p $O
bash tools/gpu_ab.sh r4f/ab base prev
timeout 900 python -m pytest tests -m gpu -x -q -k "long_runs or config3 or golden_vectors or smoke or configs_4_and_5" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
