#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3n; mkdir -p $O
sweep() {  # prec sizes...
  local prec=$1; shift
  for lg in "$@"; do for e in "SSF_X=0" "SSF_ROW_FPW=1" "SSF_ROW_V=8 SSF_ROW_FPW=1" "SSF_ROW_V=8 SSF_COL_V=8 SSF_ROW_FPW=1" "SSF_COL_V=8"; do
    env $e python bench.py --log2n $lg --prec $prec --steps 200 --warmup 20 --no-cpu-baseline > $O/t.json 2>/dev/null
    echo "$prec 2^$lg $e: $(python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))" 2>&1 | tail -1)"
  done; done
}
sweep c128 12 13 14 15 16 > $O/sweep.txt 2>&1
sweep c64 18 19 20 21 >> $O/sweep.txt 2>&1
for e in "SSF_X=0" "SSF_ROW_FPW=1" "SSF_ROW_V=8 SSF_ROW_FPW=1" "SSF_ROW_V=8 SSF_COL_V=8 SSF_ROW_FPW=1"; do
  env $e python bench.py --config 1 --steps 1000 --warmup 50 --no-cpu-baseline > $O/t.json 2>/dev/null
  echo "config 1 $e: $(python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); print(round(d['value'],1))" 2>&1 | tail -1)"
done >> $O/sweep.txt 2>&1
cat $O/sweep.txt
