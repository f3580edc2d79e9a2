#!/bin/bash
# new defaults for fields that do not fill the chip: full GPU suite, sizes sweep, units, configs 1-3 unchanged?
cd "$(dirname "$0")/.."
O=gpurun_out/r3o; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_full.log | tail -2
for lg in 12 13 14 15 16 17 18 19 20; do
  python bench.py --log2n $lg --steps 200 --warmup 20 --no-cpu-baseline > $O/t.json 2>/dev/null
  echo "c128 2^$lg: $(python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))" 2>&1 | tail -1)"
done
for lg in 18 19 20 21; do
  python bench.py --log2n $lg --prec c64 --steps 200 --warmup 20 --no-cpu-baseline > $O/t.json 2>/dev/null
  echo "c64 2^$lg: $(python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))" 2>&1 | tail -1)"
done
python bench.py --config 1 --steps 1000 --warmup 50 > $O/c1.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/c1.json').read().strip().splitlines()[-1]); print('config 1:', round(d['value'],1), 'steps/s', 'parity', d['parity']['ok'])"
python tools/bench_units.py 12 14 16 18 2>&1 | cut -c1-210
python tools/bench_lengths.py 6000 12000 48000 2>&1 | cut -c1-160
