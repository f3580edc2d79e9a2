#!/bin/bash
# control flow of bench.py with several ranks, all on GPU 0, over the gloo stand-in (RCCL refuses two ranks on one device)
cd "$(dirname "$0")/.."
O=gpurun_out/r2i; mkdir -p $O
export SSF_BENCH_COMM=gloo SSF_BENCH_DEVICE=0
for n in 2 4; do
  for c in 2 4 5; do
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n * 10 + c)) bench.py --gpus $n --steps 20 --warmup 5 --config $c > $O/sim_n${n}_c$c.out 2> $O/sim_n${n}_c$c.err
    echo "n=$n config=$c rc=$? : $(tail -1 $O/sim_n${n}_c$c.out | cut -c1-260)"
    tail -1 $O/sim_n${n}_c$c.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   value', d['value'], 'n_gpus', d['n_gpus'], 'scaling', d['scaling'], 'units', d['config']['units_total'], d['config']['units_per_gpu'], 'checksums', len(d['unit_checksums']), 'parity', d.get('parity',{}).get('ok'), 'comm', d['comm'][:40])" 2>&1 | tail -1
  done
done
