#!/bin/bash
# round 3, GPU call 4: priority by phase, write-through rows at sizes with more workgroups than slots (parity), the round's
# new GPU tests (independent units, long edc filters, complex64 at a notebook length, the full-size config-2 fixture)
cd "$(dirname "$0")/.."
O=gpurun_out/r3d; mkdir -p $O
L=$PWD/opticommpy_amd
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
  run prio1_$rep 2 300 SSF_LIB=$L/libssf_hip_prio1.so
  run prio2_$rep 2 300 SSF_LIB=$L/libssf_hip_prio2.so
  run wtR_$rep 2 300 SSF_LIB=$L/libssf_hip_wtR.so
  run prio1wt_$rep 2 300 SSF_LIB=$L/libssf_hip_prio1wt.so
done
run base1 3 100 SSF_X=0
run prio1_1 3 100 SSF_LIB=$L/libssf_hip_prio1.so
run prio2_1 3 100 SSF_LIB=$L/libssf_hip_prio2.so
# where do write-through rows stop being right?  (complex128 / complex64, one and two generations of workgroups)
for a in "21 c128" "22 c128" "20 c64" "21 c64" "22 c64"; do
  set -- $a
  SSF_LIB=$L/libssf_hip_wtR.so python bench.py --log2n $1 --prec $2 --steps 40 --warmup 5 --no-kernel-times --cpu-steps 4 > $O/wtR_n$1_$2.json 2> $O/wtR_n$1_$2.err
  echo "wtR 2^$1 $2 rc=$? $(python -c "
import json; d=json.loads(open('$O/wtR_n$1_$2.json').read().strip().splitlines()[-1]); print(d.get('parity'), d['config']['iterations_per_step'])" 2>&1 | tail -1)"
done
python bench.py --log2n 22 --prec c64 --steps 40 --warmup 5 --no-kernel-times --cpu-steps 4 > $O/base_n22_c64.json 2> $O/base_n22_c64.err
echo "base 2^22 c64 rc=$? $(python -c "
import json; d=json.loads(open('$O/base_n22_c64.json').read().strip().splitlines()[-1]); print(d.get('parity'), d['config']['iterations_per_step'])" 2>&1 | tail -1)"
timeout 1500 python -m pytest tests/test_round3.py tests/test_long_runs.py tests/test_coupled_gpu.py -m gpu -q -x -s > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; grep -E "16 units|passed|failed|Error" $O/pytest_new.log | tail -8
