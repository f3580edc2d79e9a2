#!/bin/bash
# round 4, final binary (sparse field store + recovery): profiles of configs 2 and 3, the driver's command three times, fuzzing
# (with the near-the-bound mode that forces recoveries), soak, full-size configurations, the 1000-step span
cd "$(dirname "$0")/.."
O=gpurun_out/r4y; mkdir -p $O
bash tools/gpu_profiles.sh 2 r4 > $O/profiles_c2.log 2>&1; tail -12 $O/profiles_c2.log
bash tools/gpu_profiles.sh 3 r4 > $O/profiles_c3.log 2>&1; tail -12 $O/profiles_c3.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$i.json 2> $O/driver_cmd_$i.err; python -c "
import json; d=json.loads(open('$O/driver_cmd_$i.json').read().strip().splitlines()[-1]); print('driver command run $i:', round(d['value'],1), round(d['roofline']['frac'],4), d['parity']['ok'], {k: (round(v['value'],1), round(v['roofline_frac'],3), v['parity']['ok']) for k, v in d['also'].items()})"; done | tee $O/driver_cmd.txt
timeout 420 python tests/tools/fuzz_gpu.py 200 81 near > $O/fuzz_81_near.log 2>&1; echo "rc=$?" >> $O/fuzz_81_near.log; grep -v WARNING $O/fuzz_81_near.log | tail -3
timeout 300 python tests/tools/fuzz_gpu.py 200 83 > $O/fuzz_83.log 2>&1; echo "rc=$?" >> $O/fuzz_83.log; grep -v WARNING $O/fuzz_83.log | tail -2
timeout 300 python tests/tools/soak_gpu.py 20 > $O/soak.log 2>&1; echo "rc=$?" >> $O/soak.log; tail -2 $O/soak.log
timeout 400 python tests/tools/full_configs.py > $O/full_configs.log 2>&1; echo "rc=$?" >> $O/full_configs.log; grep -E "^C[1-5]|^   " $O/full_configs.log
python bench.py --steps 1000 --warmup 50 --no-also > $O/bench_1000.json 2> $O/bench_1000.err; python -c "
import json; d=json.loads(open('$O/bench_1000.json').read().strip().splitlines()[-1]); print('1000-step span:', round(d['value'],1), round(d['roofline']['frac'],4), d['config']['iterations_per_step'])"
