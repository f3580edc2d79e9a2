#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3s; mkdir -p $O
{
for lg in 12 13 14 16 18 19; do
  python bench.py --log2n $lg --steps 200 --warmup 20 --no-cpu-baseline > $O/t.json 2>/dev/null
  echo "c128 2^$lg: $(python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))" 2>&1 | tail -1)"
done
for lg in 14 16 18 19 20; do
  python bench.py --log2n $lg --prec c64 --steps 200 --warmup 20 --no-cpu-baseline > $O/t.json 2>/dev/null
  echo "c64 2^$lg: $(python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))" 2>&1 | tail -1)"
done
python tools/bench_units.py 12 14 16 2>&1 | cut -c1-210
} > $O/small.txt 2>&1
cat $O/small.txt
