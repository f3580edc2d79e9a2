#!/bin/bash
# round 3, GPU call 6: do write-through rows / priority by phase help or hurt when two lanes share the GPU (configs 4 and 5)?
cd "$(dirname "$0")/.."
O=gpurun_out/r3f; mkdir -p $O
L=$PWD/opticommpy_amd
for rep in 1 2; do for c in 4 5; do for t in base p0w0 p0w1 p1w0; do
  lib=$L/libssf_hip.so; [ $t = base ] || lib=$L/libssf_hip_$t.so
  SSF_LIB=$lib python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > $O/${t}_c${c}_$rep.json 2> $O/${t}_c${c}_$rep.err
  echo "$t c$c rep $rep: $(python -c "
import json; d=json.loads(open('$O/${t}_c${c}_$rep.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['frac'],4), 'it/step', round(d['config']['iterations_per_step'],2))" 2>&1 | tail -1)"
done; done; done
