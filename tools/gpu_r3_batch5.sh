#!/bin/bash
# round 3, GPU call 5: new default build (write-through rows in double precision, priority by phase), the round's new GPU
# tests, independent units per launch, AUTO against both engines at awkward lengths
cd "$(dirname "$0")/.."
O=gpurun_out/r3e; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; echo "driver cmd rc=$? $(cut -c1-120 $O/driver_cmd.json)"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/c2_300.json 2>&1; python -c "
import json; d=json.loads(open('$O/c2_300.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('c2 300 steps', round(d['value'],1), round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))"
python tools/bench_units.py 12 14 16 18 > $O/units.txt 2>&1; cat $O/units.txt
python tools/bench_lengths.py 97 1500 3000 6000 2002 6006 30030 10007 12000 48000 > $O/lengths.txt 2>&1; cat $O/lengths.txt
timeout 1800 python -m pytest tests/test_round3.py tests/test_long_runs.py tests/test_coupled_gpu.py tests/test_round2.py -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; grep -E "16 units|passed|failed|Error|FAILED" $O/pytest_new.log | tail -12
