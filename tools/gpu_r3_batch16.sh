#!/bin/bash
cd "$(dirname "$0")/.."
python tools/bench_units.py 12 14 16 18 2>&1 | cut -c1-210
timeout 600 python -m pytest tests/test_round3.py -m gpu -q -k "units" 2>&1 | tail -2
