#!/bin/bash
# round 4, seventh GPU call: same-box A/B of the concurrent observer (base) against the build before it (prev); parity tests; a stress of the
# cross-stream protocol (many short spans, units, lanes)
cd "$(dirname "$0")/.."
O=gpurun_out/r4h; mkdir -p $O
bash tools/gpu_ab.sh r4h/ab base prev
timeout 1200 python -m pytest tests -m gpu -x -q -k "long_runs or config3 or golden_vectors or configs_4_and_5 or units or snapshots or test_gpu_parity or round2 or smoke" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
