#!/bin/bash
# round 3, GPU call 3 (diagnostics): A / B workgroups of a CU, split / workgroup-size knobs, what the trigonometry, the LDS
# exchanges and the butterflies cost by themselves, write-through stores with the stores drained (parity checked).
cd "$(dirname "$0")/.."
O=gpurun_out/r3c; mkdir -p $O
L=$PWD/opticommpy_amd
PHASE_GROUPS=2 PHASE_GROUP_MODE=2 SSF_LIB=$L/libssf_hip_phase.so python tools/phase_timing.py 20 > $O/phase1_c2_halves.txt 2>&1
PHASE_GROUPS=2 PHASE_GROUP_MODE=2 SSF_LIB=$L/libssf_hip_phase2.so python tools/phase_timing.py 20 > $O/phase2_c2_halves.txt 2>&1
grep -A3 "group 0" $O/phase1_c2_halves.txt | head -12; grep "group" $O/phase2_c2_halves.txt
run() {  # tag config steps env...
  local tag=$1 c=$2 steps=$3; shift 3
  env "$@" python bench.py --config $c --steps $steps --warmup 30 --no-cpu-baseline > $O/${tag}_c${c}.json 2> $O/${tag}_c${c}.err
  echo "$tag c$c: $(python - <<PY
import json
try:
    d=json.loads(open('$O/${tag}_c${c}.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']
    print(d['value'] and round(d['value'],1), round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2), 'it/step', round(d['config']['iterations_per_step'],2))
except Exception as e:
    print('FAILED', e)
PY
)"
}
for rep in 1 2; do
  run base$rep 2 300 SSF_X=0
  run l1_9_$rep 2 300 SSF_SPLIT_L1=9
  run l1_9_fpw1_$rep 2 300 SSF_SPLIT_L1=9 SSF_ROW_FPW=1
  run l1_9_v8fpw1_$rep 2 300 SSF_SPLIT_L1=9 SSF_ROW_FPW=1 SSF_ROW_V=8
  run l1_7_$rep 2 300 SSF_SPLIT_L1=7
  run l1_10_fpw1_$rep 2 300 SSF_SPLIT_L1=10 SSF_ROW_FPW=1
  run wtR_$rep 2 300 SSF_LIB=$L/libssf_hip_wtR.so
  run ft1_$rep 2 300 SSF_LIB=$L/libssf_hip_ft1.so
  run ft2_$rep 2 300 SSF_LIB=$L/libssf_hip_ft2.so
done
run abl5 2 100 SSF_LIB=$L/libssf_hip_abl5.so
run abl6 2 100 SSF_LIB=$L/libssf_hip_abl6.so
run base1 3 100 SSF_X=0
run wtR_1 3 100 SSF_LIB=$L/libssf_hip_wtR.so
run wtR_2 3 100 SSF_LIB=$L/libssf_hip_wtR.so
run ft1_1 3 100 SSF_LIB=$L/libssf_hip_ft1.so
# write-through rows: parity against the oracle at full size (config 2) and the complex64 gate (config 3)
for c in 2 3; do
  SSF_LIB=$L/libssf_hip_wtR.so python bench.py --config $c --steps 60 --warmup 10 --no-kernel-times > $O/wtR_parity_c$c.json 2> $O/wtR_parity_c$c.err
  echo "wtR parity c$c rc=$? $(python -c "
import json; d=json.loads(open('$O/wtR_parity_c$c.json').read().strip().splitlines()[-1]); print(d.get('parity'), d['config']['iterations_per_step'])")"
done
SSF_LIB=$L/libssf_hip_wtR.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_long_runs.py -m gpu -x -q -k "not config3" > $O/pytest_wtR.log 2>&1; echo "pytest wtR rc=$?"; tail -2 $O/pytest_wtR.log
