#!/bin/bash
# round 4, first GPU call: instruction-rate microbenchmark, A/B of the twiddle-table / write-through variants, the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4a
tools/exp/valu_rates > gpurun_out/r4a/valu_rates.txt 2>&1; tail -30 gpurun_out/r4a/valu_rates.txt
bash tools/gpu_ab.sh r4a/ab base tw0 tw1 wt7 wt0
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a/pytest.log
tail -15 gpurun_out/r4a/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a/bench_driver_cmd.json 2> gpurun_out/r4a/bench_driver_cmd.err; tail -c 3000 gpurun_out/r4a/bench_driver_cmd.json
