#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "ssfm or golden or property or edge or api_contract or dbp_and" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
B="python bench.py --config 1 --no-kernel-times --steps 2000 --warmup 100"
for lg in 12 14 16 18 20; do
  for p in 0 256; do
    SSF_PERSIST=$p timeout 120 $B --log2n $lg > $O/c1_${lg}_p$p.json 2> $O/c1_${lg}_p$p.err
    echo "log2n=$lg persist=$p: $(python -c "
import json
try:
    d=json.loads(open('$O/c1_${lg}_p$p.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d.get('parity'))
except Exception as e: print('ERR', e, open('$O/c1_${lg}_p$p.err').read()[-300:])
")"
  done
done
