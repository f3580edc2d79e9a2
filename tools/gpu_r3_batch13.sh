#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3m; mkdir -p $O
for lg in 17 18 19; do for e in "SSF_X=0" "SSF_ROW_V=8" "SSF_COL_V=8" "SSF_ROW_V=8 SSF_COL_V=8" "SSF_ROW_FPW=1" "SSF_ROW_V=8 SSF_ROW_FPW=1"; do
  env $e python bench.py --log2n $lg --steps 200 --warmup 20 --no-cpu-baseline > $O/t.json 2>/dev/null
  echo "2^$lg $e: $(python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))" 2>&1 | tail -1)"
done; done
