#!/bin/bash
# same-box A/B of library builds on the DRIVER'S command (20 timed steps after 5 warm-up steps), interleaved repetitions
#   bash tools/gpu_ab_driver_cmd.sh <out-tag> <reps> <tag> [<tag> ...]      ("base" = the product build)
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O; REPS=$2; shift; shift
L=$PWD/opticommpy_amd
lib() { [ "$1" = "base" ] && echo $L/libssf_hip.so || echo $L/libssf_hip_$1.so; }
for rep in $(seq 1 $REPS); do for t in "$@"; do
  SSF_LIB=$(lib $t) python bench.py --gpus 1 --steps 20 --warmup 5 --no-also > $O/${t}_$rep.json 2> $O/${t}_$rep.err
  echo "$t rep $rep: $(python -c "
import json; d=json.loads(open('$O/${t}_$rep.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2), d['parity'].get('ok'))" 2>&1 | tail -1)"
done; done | tee $O/summary.txt
