#!/bin/bash
# regression check: the library before the chained-launch wrappers / stagger window (libssf_hip_prechain.so, built from 9f091a8)
# against the final one, configs 4, 5, 2, alternating on one box
cd "$(dirname "$0")/.."
O=gpurun_out/r3y; mkdir -p $O
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
    SSF_LIB=$L timeout 300 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline > $O/c5_${v}_$rep.json 2> $O/c5_${v}_$rep.err; echo "config 5 $v $rep: $(val $O/c5_${v}_$rep.json)"
    SSF_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-times > $O/c2_${v}_$rep.json 2> $O/c2_${v}_$rep.err; echo "config 2 $v $rep: $(val $O/c2_${v}_$rep.json)"
  done
done
