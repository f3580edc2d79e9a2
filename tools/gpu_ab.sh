#!/bin/bash
# same-box A/B of two library builds: bash tools/gpu_ab.sh <tagA> <tagB>   ("" = product build)
cd "$(dirname "$0")/.."
O=gpurun_out/ab_$1_$2; mkdir -p $O
L=$PWD/opticommpy_amd
lib() { [ -z "$1" ] || [ "$1" = "base" ] && echo $L/libssf_hip.so || echo $L/libssf_hip_$1.so; }
for rep in 1 2; do for t in "$1" "$2"; do
  for c in 2 3; do
    SSF_LIB=$(lib $t) python bench.py --config $c --steps $([ $c = 2 ] && echo 400 || echo 150) --warmup 20 --no-cpu-baseline > $O/${t}_c${c}_$rep.json 2>&1
    echo "$t config $c rep $rep: $(python -c "
import json; d=json.loads(open('$O/${t}_c${c}_$rep.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))")"
  done
done; done
