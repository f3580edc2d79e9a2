#!/bin/bash
# same-box A/B of an environment knob of the experiment library (libssf_hip_exp.so):
#   bash tools/gpu_ab_env.sh <out-tag> <VAR> <value> [<value> ...]      AB_CONFIGS="2 3" AB_REPS=3
# interleaved repetitions; configs 2 (300 steps) and 3 (120 steps)
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O; VAR=$2; shift 2
export SSF_LIB=$PWD/opticommpy_amd/libssf_hip_exp.so
for rep in $(seq 1 ${AB_REPS:-3}); do for v in "$@"; do
  for c in ${AB_CONFIGS:-2 3}; do
    env $VAR=$v python bench.py --config $c --steps $([ $c = 2 ] && echo 300 || echo 120) --warmup 20 --no-also \
        $([ $c = 3 ] && echo "--parity fixture_cfg3" || echo "--cpu-steps 4") > $O/${VAR}${v}_c${c}_$rep.json 2> $O/${VAR}${v}_c${c}_$rep.err
    echo "$VAR=$v config $c rep $rep: $(python -c "
import json; d=json.loads(open('$O/${VAR}${v}_c${c}_$rep.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1) if d['value'] else None, round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2), 'parity', {x: (('%.2e' % y) if isinstance(y, float) else y) for x, y in d['parity'].items() if x in ('rel_l2_vs_oracle', 'rel_l2_vs_reference_c128', 'ok')})" 2>&1 | tail -1)"
  done
done; done | tee $O/summary.txt
