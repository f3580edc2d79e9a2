#!/bin/bash
# round 3, GPU call 7: lanes hint (no phase priorities when plans share the GPU), large units per plan against lanes
cd "$(dirname "$0")/.."
O=gpurun_out/r3g; mkdir -p $O
L=$PWD/opticommpy_amd
for rep in 1 2; do for c in 4 5; do for t in base p0w1; do
  lib=$L/libssf_hip.so; [ $t = base ] || lib=$L/libssf_hip_$t.so
  SSF_LIB=$lib python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > $O/${t}_c${c}_$rep.json 2> $O/${t}_c${c}_$rep.err
  echo "$t c$c rep $rep: $(python -c "
import json; d=json.loads(open('$O/${t}_c${c}_$rep.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['frac'],4), 'it/step', round(d['config']['iterations_per_step'],2))" 2>&1 | tail -1)"
done; done; done
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/c2.json 2>&1; python -c "
import json; d=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('c2', round(d['value'],1), round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))"
python tools/bench_units_large.py 20 8 100 > $O/units_large.txt 2>&1; cat $O/units_large.txt
python tools/bench_units_large.py 18 16 100 > $O/units_large18.txt 2>&1; cat $O/units_large18.txt
