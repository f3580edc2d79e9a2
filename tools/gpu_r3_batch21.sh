#!/bin/bash
# chained launches (SSF_CHAIN=1: launches alternate between two streams, every workgroup waits inside the kernel for the previous
# launch) against the plain launch sequence: config 2 (300 steps + driver command), config 3, small sizes; parity through bench's oracle gate
cd "$(dirname "$0")/.."
O=gpurun_out/r3u; mkdir -p $O
val() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['roofline'].get('kernels',{})
    print(d['value'] and round(d['value'],1), 'frac', round(d['roofline']['frac'],4), 'parity', d.get('parity',{}).get('ok'), d.get('parity',{}).get('rel_l2_vs_oracle'))
except Exception as e: print('ERR', e)
PY
}
run() { # tag, env, args
  env $2 timeout 300 python bench.py $3 > $O/$1.json 2> $O/$1.err; echo "$1 [$2] $3: $(val $O/$1.json) $(tail -c 300 $O/$1.err | tr '\n' ' ')"
}
run c2_plain_a X=0 "--steps 300 --warmup 30 --no-kernel-times"
run c2_chain_a SSF_CHAIN=1 "--steps 300 --warmup 30 --no-kernel-times"
run c2_plain_b X=0 "--steps 300 --warmup 30 --no-kernel-times"
run c2_chain_b SSF_CHAIN=1 "--steps 300 --warmup 30 --no-kernel-times"
run drv_plain X=0 "--gpus 1 --steps 20 --warmup 5 --no-kernel-times --no-cpu-baseline"
run drv_chain SSF_CHAIN=1 "--gpus 1 --steps 20 --warmup 5 --no-kernel-times --no-cpu-baseline"
run drv_plain2 X=0 "--gpus 1 --steps 20 --warmup 5 --no-kernel-times --no-cpu-baseline"
run drv_chain2 SSF_CHAIN=1 "--gpus 1 --steps 20 --warmup 5 --no-kernel-times --no-cpu-baseline"
run c3_plain X=0 "--config 3 --steps 300 --warmup 30 --no-kernel-times"
run c3_chain SSF_CHAIN=1 "--config 3 --steps 300 --warmup 30 --no-kernel-times"
for n in 14 16 18 22; do
  run n${n}_plain X=0 "--log2n $n --steps 300 --warmup 30 --no-kernel-times --no-cpu-baseline"
  run n${n}_chain SSF_CHAIN=1 "--log2n $n --steps 300 --warmup 30 --no-kernel-times --no-cpu-baseline"
done
SSF_CHAIN=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_long_runs.py -m gpu -x -q 2>&1 | tail -3
