#!/bin/bash
# the product build with the chained-launch wrappers compiled out against the library before them (9f091a8), one box
cd "$(dirname "$0")/.."
O=gpurun_out/r3z; mkdir -p $O
val() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['value'] and round(d['value'],1), 'frac', round(d['roofline']['frac'],4))
except Exception as e: print('ERR', e)
PY
}
P=$PWD/opticommpy_amd
for rep in 1 2; do
  for v in final prechain; do
    L=$P/libssf_hip.so; [ $v != final ] && L=$P/libssf_hip_$v.so
    SSF_LIB=$L timeout 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline > $O/c4_${v}_$rep.json 2> $O/c4_${v}_$rep.err; echo "config 4 $v $rep: $(val $O/c4_${v}_$rep.json)"
    SSF_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-times > $O/c2_${v}_$rep.json 2> $O/c2_${v}_$rep.err; echo "config 2 $v $rep: $(val $O/c2_${v}_$rep.json)"
  done
done
