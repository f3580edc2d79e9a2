#!/bin/bash
# Round-3 final measurements on one box: profiles of configs 2 and 3 (tools/gpu_r3_profiles.sh), phase stamps of the final kernels,
# the other configurations, the self-launching N = 2 line, out-of-cache complex128, two knob checks.
cd "$(dirname "$0")/.."
O=gpurun_out/r3_final; mkdir -p $O
L=$PWD/opticommpy_amd
bash tools/gpu_r3_profiles.sh 2 > $O/prof_c2.log 2>&1; tail -12 $O/prof_c2.log
bash tools/gpu_r3_profiles.sh 3 > $O/prof_c3.log 2>&1; tail -12 $O/prof_c3.log
PHASE_GROUPS=2 PHASE_GROUP_MODE=2 SSF_LIB=$L/libssf_hip_phase.so python tools/phase_timing.py 20 > $O/phase1_c2.txt 2>&1
PHASE_GROUPS=2 PHASE_GROUP_MODE=2 SSF_LIB=$L/libssf_hip_phase2.so python tools/phase_timing.py 20 > $O/phase2_c2.txt 2>&1
PHASE_GROUPS=4 PHASE_GROUP_MODE=2 SSF_LIB=$L/libssf_hip_phase2.so python tools/phase_timing.py 22 c64 > $O/phase2_c3.txt 2>&1
grep -E "group|quarter" $O/phase2_c2.txt
python bench.py --config 1 --steps 1000 --warmup 50 > $O/c1.json 2> $O/c1.err
python bench.py --config 4 --steps 100 --warmup 10 > $O/c4.json 2> $O/c4.err
python bench.py --config 5 --steps 100 --warmup 10 > $O/c5.json 2> $O/c5.err
python bench.py --steps 1000 --warmup 50 > $O/c2_1000.json 2> $O/c2_1000.err
python bench.py --log2n 22 --steps 100 --warmup 10 --cpu-steps 3 > $O/c2_out_of_cache_n22.json 2> $O/c2_out_of_cache_n22.err
SSF_BENCH_DEVICE=0 SSF_BENCH_COMM=gloo python bench.py --gpus 2 --config 4 --steps 50 --warmup 5 > $O/c4_two_ranks_one_gpu_gloo.json 2> $O/c4_two_ranks.err
for f in c1 c4 c5 c2_1000 c2_out_of_cache_n22 c4_two_ranks_one_gpu_gloo; do echo "$f: $(python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['unit'], 'frac', round(d['roofline']['frac'],4), 'n_gpus', d['n_gpus'], 'parity', d.get('parity',{}).get('ok'), 'it/step', round(d['config']['iterations_per_step'],2))" 2>&1 | tail -1)"; done
for e in "SSF_SPLIT_L1=9 SSF_COL_HALF=128 SSF_ROW_FPW=1" "SSF_COL_HALF=64" "SSF_X=1"; do
  env $e python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/knob.json 2>/dev/null
  echo "$e: $(python -c "
import json; d=json.loads(open('$O/knob.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))" 2>&1 | tail -1)"
done
timeout 900 python -m pytest tests/test_round3.py -m gpu -q -k "longest or units or edc" > $O/pytest_late.log 2>&1; echo "pytest late rc=$?"; tail -2 $O/pytest_late.log
du -sh gpurun_out/r3_prof
