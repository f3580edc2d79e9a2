#!/bin/bash
# driver command with the parity gate / probe / per-kernel pass ahead of the timed region (the default now) against the same
# command without those legs (cold GPU at the warm-up), alternating on one box; then configs 3, 1, 4 as the driver would run them
cd "$(dirname "$0")/.."
O=gpurun_out/r3v; mkdir -p $O
val() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['roofline'].get('kernels',{})
    print(d['value'] and round(d['value'],1), 'frac', round(d['roofline']['frac'],4), 'row', round(k.get('row',{}).get('avg_us',0),2), 'col', round(k.get('col',{}).get('avg_us',0),2), 'parity', d.get('parity',{}).get('ok'), 'cpu', d.get('cpu_baseline',{}).get('value'))
except Exception as e: print('ERR', e)
PY
}
for rep in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv_full_$rep.json 2> $O/drv_full_$rep.err; echo "driver cmd (legs first)  $rep: $(val $O/drv_full_$rep.json)"
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-times > $O/drv_cold_$rep.json 2> $O/drv_cold_$rep.err; echo "driver cmd (no legs)     $rep: $(val $O/drv_cold_$rep.json)"
done
python bench.py --config 3 --steps 20 --warmup 5 > $O/c3_drv.json 2> $O/c3_drv.err; echo "config 3 driver-style: $(val $O/c3_drv.json)"
python bench.py --config 1 --steps 100 --warmup 10 > $O/c1.json 2> $O/c1.err; echo "config 1: $(val $O/c1.json)"
python bench.py --config 4 --steps 100 --warmup 10 > $O/c4.json 2> $O/c4.err; echo "config 4: $(val $O/c4.json)"
python bench.py > $O/default.json 2> $O/default.err; echo "no flags: $(val $O/default.json)"
timeout 600 python -m pytest tests/test_round3.py -m gpu -x -q 2>&1 | tail -3
