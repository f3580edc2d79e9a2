#!/bin/bash
# round 3, GPU call 1: new tests, then same-box A/B of the row-kernel family (16 vs 8 values per thread, staggered start)
# and of write-through stores.  Everything lands in gpurun_out/r3a.
cd "$(dirname "$0")/.."
O=gpurun_out/r3a; mkdir -p $O
L=$PWD/opticommpy_amd
( timeout 900 python -m pytest tests/test_round3.py tests/test_coupled_gpu.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" ) 
tail -3 $O/pytest_new.log
( SSF_ROW_V=8 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity_v8.log 2>&1; echo "pytest parity V8 rc=$?" )
tail -3 $O/pytest_parity_v8.log
run() {  # tag config steps env...
  local tag=$1 c=$2 steps=$3; shift 3
  env "$@" python bench.py --config $c --steps $steps --warmup 30 --no-cpu-baseline > $O/${tag}_c${c}.json 2> $O/${tag}_c${c}.err
  echo "$tag c$c: $(python - <<PY
import json
try:
    d=json.loads(open('$O/${tag}_c${c}.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']
    print(round(d['value'],1), round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2), 'copy', round(d['roofline'].get('measured_copy_GBs',0)))
except Exception as e:
    print('FAILED', e)
PY
)"
}
for rep in 1 2; do
  run base$rep 2 300 SSF_X=0
  run v8_$rep 2 300 SSF_ROW_V=8
  for s in 20 40 60 90 120; do run v8s${s}_$rep 2 300 SSF_ROW_V=8 SSF_ROW_STAGGER=$s; done
  run v16s60_$rep 2 300 SSF_ROW_STAGGER=60
  run wt1_$rep 2 300 SSF_LIB=$L/libssf_hip_wt1.so
  run wt5_$rep 2 300 SSF_LIB=$L/libssf_hip_wt5.so
  run wt1v8_$rep 2 300 SSF_LIB=$L/libssf_hip_wt1.so SSF_ROW_V=8 SSF_ROW_STAGGER=60
done
for rep in 1 2; do
  run base$rep 3 100 SSF_X=0
  run v8_$rep 3 100 SSF_ROW_V=8
  run v8s60_$rep 3 100 SSF_ROW_V=8 SSF_ROW_STAGGER=60
  run wt1_$rep 3 100 SSF_LIB=$L/libssf_hip_wt1.so
done
# full-size parity of the 8-value rows against the oracle (bench's own gate) + the driver's command on this binary
SSF_ROW_V=8 python bench.py --steps 100 --warmup 20 > $O/v8_parity_c2.json 2> $O/v8_parity_c2.err; echo "v8 parity rc=$? $(python -c "
import json; d=json.loads(open('$O/v8_parity_c2.json').read().strip().splitlines()[-1]); print(d.get('parity'))")"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; echo "driver cmd rc=$? $(cut -c1-200 $O/driver_cmd.json)"
