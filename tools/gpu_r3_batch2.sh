#!/bin/bash
# round 3, GPU call 2 (diagnostics): where the launches spend their time.  Phase stamps (draining and non-draining), the row
# kernel with its arithmetic / its memory phase removed, write-through stores per store site, 8-value column kernels.
cd "$(dirname "$0")/.."
O=gpurun_out/r3b; mkdir -p $O
L=$PWD/opticommpy_amd
SSF_LIB=$L/libssf_hip_phase.so python tools/phase_timing.py 20 > $O/phase1_c2.txt 2>&1
SSF_LIB=$L/libssf_hip_phase2.so python tools/phase_timing.py 20 > $O/phase2_c2.txt 2>&1
SSF_LIB=$L/libssf_hip_phase.so python tools/phase_timing.py 22 c64 > $O/phase1_c3.txt 2>&1
SSF_LIB=$L/libssf_hip_phase2.so python tools/phase_timing.py 22 c64 > $O/phase2_c3.txt 2>&1
cat $O/phase2_c2.txt
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
  for t in wtR wtC wtRC wtRCF; do run ${t}_$rep 2 300 SSF_LIB=$L/libssf_hip_$t.so; done
  run colv8_$rep 2 300 SSF_COL_V=8
  run bothv8_$rep 2 300 SSF_COL_V=8 SSF_ROW_V=8
done
run abl1 2 100 SSF_LIB=$L/libssf_hip_abl1.so
run abl2 2 100 SSF_LIB=$L/libssf_hip_abl2.so
run abl1v8 2 100 SSF_LIB=$L/libssf_hip_abl1.so SSF_ROW_V=8
run abl2v8 2 100 SSF_LIB=$L/libssf_hip_abl2.so SSF_ROW_V=8
for rep in 1 2; do
  run base$rep 3 100 SSF_X=0
  for t in wtR wtC wtRC; do run ${t}_$rep 3 100 SSF_LIB=$L/libssf_hip_$t.so; done
  run colv8_$rep 3 100 SSF_COL_V=8
done
SSF_COL_V=8 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity_colv8.log 2>&1; echo "pytest parity COL_V=8 rc=$?"; tail -2 $O/pytest_parity_colv8.log
