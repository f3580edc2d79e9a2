#!/bin/bash
# (1) driver command with device-resident inputs against the re-upload from host memory, alternating on one box
# (2) config 3: the second resident workgroup of every CU started late (rows / columns), 400 steps
cd "$(dirname "$0")/.."
O=gpurun_out/r3t; mkdir -p $O
val() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['roofline'].get('kernels',{})
    print(round(d['value'],1), 'frac', round(d['roofline']['frac'],4), 'row', round(k.get('row',{}).get('avg_us',0),2), 'col', round(k.get('col',{}).get('avg_us',0),2), 'parity', d.get('parity',{}).get('ok'))
except Exception as e: print('ERR', e)
PY
}
for rep in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv_dev_$rep.json 2> $O/drv_dev_$rep.err; echo "driver cmd, device inputs  $rep: $(val $O/drv_dev_$rep.json)"
  SSF_BENCH_HOST_INPUT=1 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv_host_$rep.json 2> $O/drv_host_$rep.err; echo "driver cmd, host inputs    $rep: $(val $O/drv_host_$rep.json)"
done
C3="python bench.py --config 3 --steps 400 --warmup 30 --no-cpu-baseline"
for e in "X=0" "SSF_ROW_STAGGER=100" "SSF_ROW_STAGGER=200" "SSF_ROW_STAGGER=300" "SSF_ROW_STAGGER=400" "SSF_COL_STAGGER=150" "SSF_COL_STAGGER=300" "SSF_COL_STAGGER=450" "SSF_ROW_STAGGER=250 SSF_COL_STAGGER=300" "X=1"; do
  t=$(echo $e | tr ' =' '__')
  env $e timeout 300 $C3 > $O/c3_$t.json 2> $O/c3_$t.err; echo "config 3 $e: $(val $O/c3_$t.json)"
done
python bench.py --config 3 --steps 20 --warmup 5 > $O/c3_drv.json 2> $O/c3_drv.err; echo "config 3 driver-style: $(val $O/c3_drv.json)"
