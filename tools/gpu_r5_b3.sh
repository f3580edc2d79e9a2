#!/bin/bash
# round 5, batch 3: the GPU suite on the new sources; driver's command with / without the stage-specialised kernels; per-stage kernel times
cd "$(dirname "$0")/.."
REPO=$PWD; O=$REPO/gpurun_out/r5_b3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|chain vs|config 3, one span" $O/pytest.log | tail -15
export SSF_LIB=$REPO/opticommpy_amd/libssf_hip_exp.so
for rep in 1 2 3; do for sp in 0 1; do
  SSF_COL_SPLIT=$sp python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --cpu-steps 4 > $O/drv_split${sp}_$rep.json 2> $O/drv_split${sp}_$rep.err
  echo "driver cmd split=$sp rep $rep: $(python -c "
import json; d=json.loads(open('$O/drv_split${sp}_$rep.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1), round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2))" 2>&1 | tail -1)"
done; done | tee $O/driver_cmd_ab.txt
unset SSF_LIB
export TMPDIR=/tmp
cd /tmp
for c in 2 3; do
  S=$([ $c = 2 ] && echo 200 || echo 100)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_c$c -o kt -- python $REPO/bench.py --config $c --no-cpu-baseline --no-kernel-times --no-also --parity none --steps $S --warmup 10 > $O/kt_c$c.log 2>&1
  python $REPO/tools/rocpd_stats.py "$(find $O/kt_c$c -name '*.db' | head -1)" > $O/kernel_stats_c$c.txt 2>&1
  find $O/kt_c$c -name '*.db' -delete
  head -12 $O/kernel_stats_c$c.txt | cut -c1-170
done
