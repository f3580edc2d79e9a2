#!/bin/bash
# final verification on one box: smoke(), the whole -m gpu suite, the driver's bench command, configs 4 / 5 with the parity leg
cd "$(dirname "$0")/.."
O=gpurun_out/r3_verify; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_full.log | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; echo "driver rc=$? $(cut -c1-110 $O/driver_cmd.json)"
python bench.py --config 4 --steps 100 --warmup 10 > $O/c4.json 2> $O/c4.err
python bench.py --config 5 --steps 100 --warmup 10 > $O/c5.json 2> $O/c5.err
for f in c4 c5; do echo "$f: $(python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print(round(d['value'],1), 'frac', round(d['roofline']['frac'],4), 'parity', d.get('parity',{}).get('ok'))" 2>&1 | tail -1)"; done
