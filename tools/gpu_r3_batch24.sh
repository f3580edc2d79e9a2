#!/bin/bash
# issue priority 3 for the LDS exchanges of every transform (variant builds libssf_hip_xprio1 / _xprio2.so) against the default build
cd "$(dirname "$0")/.."
O=gpurun_out/r3x; mkdir -p $O
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
  for v in base xprio1 xprio2; do
    L=$P/libssf_hip.so; [ $v != base ] && L=$P/libssf_hip_$v.so
    SSF_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/c2_${v}_$rep.json 2> $O/c2_${v}_$rep.err; echo "config 2 $v $rep: $(val $O/c2_${v}_$rep.json)"
  done
done
for v in base xprio1 xprio2; do
  L=$P/libssf_hip.so; [ $v != base ] && L=$P/libssf_hip_$v.so
  SSF_LIB=$L timeout 300 python bench.py --config 3 --steps 300 --warmup 30 --no-cpu-baseline > $O/c3_$v.json 2> $O/c3_$v.err; echo "config 3 $v: $(val $O/c3_$v.json)"
  SSF_LIB=$L timeout 300 python bench.py --log2n 16 --steps 300 --warmup 30 --no-cpu-baseline > $O/n16_$v.json 2> $O/n16_$v.err; echo "2^16 $v: $(val $O/n16_$v.json)"
done
SSF_LIB=$P/libssf_hip_xprio1.so timeout 200 python bench.py --steps 50 --warmup 5 > $O/c2_xprio1_parity.json 2>/dev/null; echo "xprio1 with oracle gate: $(val $O/c2_xprio1_parity.json)"
timeout 300 python -m pytest tests/test_round3.py -m gpu -x -q -k chained 2>&1 | tail -3
