#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3r; mkdir -p $O
for lg in 14 15 16 17 18; do
  python bench.py --log2n $lg --steps 200 --warmup 20 --no-cpu-baseline > $O/t.json 2>/dev/null
  echo "c128 2^$lg: $(python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))" 2>&1 | tail -1)"
done
python bench.py --config 1 --steps 1000 --warmup 50 > $O/c1.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c1.json').read().strip().splitlines()[-1]); print('config 1:', round(d['value'],1), 'steps/s', 'parity', d['parity']['ok'])"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_round3.py -m gpu -q -x 2>&1 | tail -2
