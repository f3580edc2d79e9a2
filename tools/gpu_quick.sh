#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2j; mkdir -p $O
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-times > $O/d$i.json 2>&1; done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-times > $O/s200.json 2>&1
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-kernel-times > $O/s1000.json 2>&1
python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-times > $O/c3.json 2>&1
for f in $O/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), round(d['roofline']['frac'],4), d['config']['iterations_per_step'])"; done
