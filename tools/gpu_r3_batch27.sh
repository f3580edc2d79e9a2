#!/bin/bash
# G tile fetched before the control block is read (column stage) against the library before that change, one box
cd "$(dirname "$0")/.."
O=gpurun_out/r3aa; mkdir -p $O
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
  for v in final prechain; do
    L=$P/libssf_hip.so; [ $v != final ] && L=$P/libssf_hip_$v.so
    SSF_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/c2_${v}_$rep.json 2> $O/c2_${v}_$rep.err; echo "config 2 $v $rep: $(val $O/c2_${v}_$rep.json)"
    SSF_LIB=$L timeout 300 python bench.py --config 3 --steps 300 --warmup 30 --no-cpu-baseline > $O/c3_${v}_$rep.json 2> $O/c3_${v}_$rep.err; echo "config 3 $v $rep: $(val $O/c3_${v}_$rep.json)"
    SSF_LIB=$L timeout 300 python bench.py --log2n 16 --steps 300 --warmup 30 --no-cpu-baseline > $O/n16_${v}_$rep.json 2> $O/n16_${v}_$rep.err; echo "2^16 $v $rep: $(val $O/n16_${v}_$rep.json)"
  done
done
python bench.py --steps 50 --warmup 5 > $O/c2_parity.json 2>/dev/null; echo "final with oracle gate: $(val $O/c2_parity.json)"
