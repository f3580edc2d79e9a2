#!/bin/bash
# round 4, last change (own quarter-turn reduction for the twiddle bases of the double-precision translation unit): A/B against the build before it, then the whole GPU suite
cd "$(dirname "$0")/.."
O=gpurun_out/r4u; mkdir -p $O
bash tools/gpu_ab.sh r4u/ab prev base prev base
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | tail -2
