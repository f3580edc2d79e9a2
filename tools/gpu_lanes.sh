#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2h; mkdir -p $O
for l in 1 2 3 4; do
  python bench.py --config 4 --steps 100 --warmup 10 --lanes $l --no-cpu-baseline --no-kernel-times > $O/c4_l$l.json 2>&1
  python bench.py --config 3 --steps 100 --warmup 10 --lanes $l --no-cpu-baseline --no-kernel-times > /dev/null 2>&1
  echo "lanes=$l: $(python -c "
import json; d=json.loads(open('$O/c4_l$l.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['frac'],4))")"
done
