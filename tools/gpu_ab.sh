#!/bin/bash
# same-box A/B of library builds: bash tools/gpu_ab.sh <out-tag> <tag> [<tag> ...]   ("base" = the product build; others =
# libssf_hip_<tag>.so from `make variant TAG=<tag> VFLAGS=...`); configs 2 and 3, two interleaved repetitions
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O; shift
L=$PWD/opticommpy_amd
lib() { [ "$1" = "base" ] && echo $L/libssf_hip.so || echo $L/libssf_hip_$1.so; }
for rep in 1 2; do for t in "$@"; do
  [ -f "$(lib $t)" ] || { echo "$t: no such library"; continue; }
  for c in ${AB_CONFIGS:-2 3}; do
    SSF_LIB=$(lib $t) python bench.py --config $c --steps $([ $c = 2 ] && echo 300 || echo 120) --warmup 20 --no-also \
        $([ $c = 3 ] && echo "--parity fixture_cfg3" || echo "--cpu-steps 4") > $O/${t}_c${c}_$rep.json 2> $O/${t}_c${c}_$rep.err
    echo "$t config $c rep $rep: $(python -c "
import json; d=json.loads(open('$O/${t}_c${c}_$rep.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['value'],1) if d['value'] else None, round(d['roofline']['frac'],4), 'row', round(k['row']['avg_us'],2), 'col', round(k['col']['avg_us'],2), 'parity', {x: (('%.2e' % y) if isinstance(y, float) else y) for x, y in d['parity'].items() if x in ('rel_l2_vs_oracle', 'rel_l2_vs_reference_c128', 'ok')})" 2>&1 | tail -1)"
  done
done; done | tee $O/summary.txt
