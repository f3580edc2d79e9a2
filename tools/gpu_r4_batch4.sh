#!/bin/bash
# round 4, fourth GPU call: A/B of the interleaved P / theta records (column stage, double precision), the 8-value packed kernels again on
# this round's cheaper twiddles, three driver-command runs of the final binary
cd "$(dirname "$0")/.."
O=gpurun_out/r4e; mkdir -p $O
AB_CONFIGS=2 bash tools/gpu_ab.sh r4e/ab base pt
L=$PWD/opticommpy_amd/libssf_hip_exp.so
for v in "SSF_ROW_V=16 SSF_COL_V=16" "SSF_ROW_V=8 SSF_COL_V=16" "SSF_ROW_V=16 SSF_COL_V=8" "SSF_ROW_V=8 SSF_COL_V=8"; do
  env SSF_LIB=$L $v python bench.py --config 3 --steps 120 --warmup 20 --no-also --parity fixture_cfg3 > $O/c3_v.json 2> $O/c3_v.err
  echo "exp lib, config 3, $v: $(python -c "
import json; d=json.loads(open('$O/c3_v.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2), d['parity']['ok'])" 2>&1 | tail -1)"
done | tee $O/c3_values_per_thread.txt
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$i.json 2> $O/driver_cmd_$i.err; python -c "
import json; d=json.loads(open('$O/driver_cmd_$i.json').read().strip().splitlines()[-1]); print('driver command run $i:', round(d['value'],1), round(d['roofline']['frac'],4), {k: (round(v['value'],1), round(v['roofline_frac'],3), v['parity']['ok']) for k, v in d['also'].items()})"; done | tee $O/driver_cmd.txt
