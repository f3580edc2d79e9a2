#!/bin/bash
# prologue trimming: prechain (9f091a8) | rowonly (row stage: no stagger test, branch-free unit_view) | final (+ column stage:
# branch-free unit_view, unconditional control-block reads); one box, alternating
cd "$(dirname "$0")/.."
O=gpurun_out/r3ab; mkdir -p $O
val() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['roofline'].get('kernels',{})
    print(d['value'] and round(d['value'],1), 'frac', round(d['roofline']['frac'],4), 'row', round(k.get('row',{}).get('avg_us',0),2), 'col', round(k.get('col',{}).get('avg_us',0),2), 'parity', d.get('parity',{}).get('ok'))
except Exception as e: print('ERR', e)
PY
}
P=$PWD/opticommpy_amd
for rep in 1 2; do
  for v in prechain rowonly final; do
    L=$P/libssf_hip.so; [ $v != final ] && L=$P/libssf_hip_$v.so
    SSF_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/c2_${v}_$rep.json 2> $O/c2_${v}_$rep.err; echo "config 2 $v $rep: $(val $O/c2_${v}_$rep.json)"
  done
done
for v in prechain rowonly final; do
  L=$P/libssf_hip.so; [ $v != final ] && L=$P/libssf_hip_$v.so
  SSF_LIB=$L timeout 300 python bench.py --config 3 --steps 300 --warmup 30 --no-cpu-baseline > $O/c3_$v.json 2> $O/c3_$v.err; echo "config 3 $v: $(val $O/c3_$v.json)"
  for n in 12 14 16 18; do
    SSF_LIB=$L timeout 300 python bench.py --log2n $n --steps 300 --warmup 30 --no-cpu-baseline > $O/n${n}_$v.json 2> $O/n${n}_$v.err; echo "2^$n $v: $(val $O/n${n}_$v.json)"
  done
  SSF_LIB=$L timeout 300 python bench.py --config 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/c1_$v.json 2> $O/c1_$v.err; echo "config 1 $v: $(val $O/c1_$v.json)"
done
