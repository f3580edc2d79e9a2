#!/bin/bash
# round 5, batch 6: headline numbers and profiles of the final binary
cd "$(dirname "$0")/.."
REPO=$PWD; O=$REPO/gpurun_out/r5_b6; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
timeout 900 python -m pytest tests/test_chain.py tests/test_long_runs.py tests/test_coupled_gpu.py tests/test_emu_tx.py -m gpu -q -s --timeout 600 -k "chain or full_span or rank_order" > $O/pytest_s.log 2>&1
grep -E "chain vs|one span|passed|failed" $O/pytest_s.log | tail -8
for rep in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 $([ $rep = 1 ] || echo --no-also) > $O/driver_cmd_$rep.json 2> $O/driver_cmd_$rep.err
  python -c "
import json; d=json.loads(open('$O/driver_cmd_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('driver cmd rep $rep:', round(d['value'],1), round(r['frac'],4), 'dominant', r['kernels'].get('dominant'))"
done | tee $O/driver_cmd.txt
python bench.py --config 2 --no-also --steps 400 --warmup 30 > $O/c2_400.json 2> $O/c2_400.err
python bench.py --config 2 --no-also --steps 1001 --warmup 30 > $O/c2_1001.json 2> $O/c2_1001.err
python bench.py --config 3 --no-also --steps 200 --warmup 20 --parity fixture_cfg3 > $O/c3_200.json 2> $O/c3_200.err
for f in c2_400 c2_1001 c3_200; do python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', round(d['value'],1), round(r['frac'],4), {k: round(v['avg_us'],2) for k,v in r['kernels'].items() if isinstance(v,dict) and 'avg_us' in v})"; done | tee $O/long_runs.txt
bash tools/gpu_profiles.sh 2 r5 > $O/prof_c2.log 2>&1; tail -12 $O/prof_c2.log
bash tools/gpu_profiles.sh 3 r5 > $O/prof_c3.log 2>&1; tail -12 $O/prof_c3.log
