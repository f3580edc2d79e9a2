#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2f; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python tools/bench_matrix.py > $O/matrix.log 2>&1
timeout 600 python tools/bench_lengths.py 97 1500 3000 12000 10007 30030 100003 1000003 48000 960000 > $O/lengths.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
grep -E "passed|failed" $O/pytest.log | tail -2; cat $O/matrix.log $O/lengths.log; tail -4 $O/smoke.log
