#!/bin/bash
# GPU validation of a source tree: smoke(), the GPU suite, the driver's command three times.   gpurun --timeout 3000 -- "bash tools/gpu_validate.sh <tag>"
cd "$(dirname "$0")/.."
REPO=$PWD; O=$REPO/gpurun_out/${1:-r5_final}; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
timeout 2000 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | tail -10
for rep in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 $([ $rep = 1 ] || echo --no-also) > $O/driver_cmd_$rep.json 2> $O/driver_cmd_$rep.err
  python -c "
import json; d=json.loads(open('$O/driver_cmd_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('driver cmd rep $rep:', round(d['value'],1), round(r['frac'],4), 'dominant', r['kernels'].get('dominant'))"
done | tee $O/driver_cmd.txt
