#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/profile_round.sh r2 2 200
bash tools/profile_round.sh r2 3 100
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_c2/bench_driver_cmd.json 2>&1
python bench.py --steps 1000 --warmup 50 > gpurun_out/r2_c2/bench_1000.json 2>&1
for c in 1 4 5; do mkdir -p gpurun_out/r2_c$c; python bench.py --config $c --steps $([ $c = 1 ] && echo 1000 || echo 100) --warmup 10 > gpurun_out/r2_c$c/bench.json 2>&1; done
python bench.py --config 5 --steps 100 --warmup 10 --dbp-hz 10 > gpurun_out/r2_c5/bench_dbp_hz10.json 2>&1
# (round 2 deleted the raw databases here; round 3: tools/gpu_r3_profiles.sh keeps them, gzip'ed, under gpurun_out/r3_prof/)
