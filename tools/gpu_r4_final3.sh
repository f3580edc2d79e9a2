#!/bin/bash
# round 4, final binary (P / Theta of the packed stage by owner thread): whole GPU suite, config 3 profiles, the driver's command three times,
# a short fuzz with the near-the-bound mode
cd "$(dirname "$0")/.."
O=gpurun_out/r4x; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -9 $O/pytest.log
bash tools/gpu_profiles.sh 3 r4 > $O/profiles_c3.log 2>&1; tail -10 $O/profiles_c3.log
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$i.json 2> $O/driver_cmd_$i.err; python -c "
import json; d=json.loads(open('$O/driver_cmd_$i.json').read().strip().splitlines()[-1]); print('driver command run $i:', round(d['value'],1), round(d['roofline']['frac'],4), d['parity']['ok'], {k: (round(v['value'],1), round(v['roofline_frac'],3), v['parity']['ok']) for k, v in d['also'].items()})"; done | tee $O/driver_cmd.txt
timeout 300 python tests/tools/fuzz_gpu.py 120 91 near > $O/fuzz_91_near.log 2>&1; echo "rc=$?" >> $O/fuzz_91_near.log; grep -v WARNING $O/fuzz_91_near.log | tail -2
