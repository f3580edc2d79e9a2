#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3i; mkdir -p $O
for v in "SSF_X=0" "SSF_ROWS_WTAB=1" "SSF_ROWS_TPR=128" "SSF_ROWS_TPR=128 SSF_ROWS_WTAB=1"; do
  echo "== $v"; env $v python tools/bench_lengths.py 1500 3000 3750 4500 6000 2>&1 | cut -c1-150
done > $O/rows.txt 2>&1
cat $O/rows.txt
timeout 600 python -m pytest tests/test_round2.py -m gpu -q -k "any_length" > $O/pytest_anylen.log 2>&1; echo "any-length tests rc=$?"; tail -1 $O/pytest_anylen.log
