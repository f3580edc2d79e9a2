#!/bin/bash
# full GPU test suite (+ optional -k expression); log in gpurun_out/<tag>/pytest.log
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-pt}; mkdir -p $O
shift
timeout 1800 python -m pytest tests -m gpu -q "$@" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | tail -30
