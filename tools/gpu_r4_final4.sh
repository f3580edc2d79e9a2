#!/bin/bash
# round 4, FINAL binary (compact large-angle paths of the single- and double-precision rotations): whole GPU suite, profiles of configs 2 and 3,
# the driver's command three times
cd "$(dirname "$0")/.."
O=gpurun_out/r4v; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$i.json 2> $O/driver_cmd_$i.err; python -c "
import json; d=json.loads(open('$O/driver_cmd_$i.json').read().strip().splitlines()[-1]); print('driver command run $i:', round(d['value'],1), round(d['roofline']['frac'],4), d['parity']['ok'], {k: (round(v['value'],1), round(v['roofline_frac'],3), v['parity']['ok']) for k, v in d['also'].items()})"; done | tee $O/driver_cmd.txt
bash tools/gpu_profiles.sh 2 r4 > $O/profiles_c2.log 2>&1; tail -9 $O/profiles_c2.log
bash tools/gpu_profiles.sh 3 r4 > $O/profiles_c3.log 2>&1; tail -9 $O/profiles_c3.log
