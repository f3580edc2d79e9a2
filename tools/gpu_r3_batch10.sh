#!/bin/bash
# round 3, GPU call 10: the Manakov span as one persistent launch (agent-scope barrier / one-XCD barrier) against the launch sequence
cd "$(dirname "$0")/.."
O=gpurun_out/r3j; mkdir -p $O
export SSF_COL_HALF=128                       # 256-thread column workgroups at every size (what the persistent kernel needs)
{
echo "== launch sequence (SSF_COL_HALF=128)"; timeout 300 python tools/bench_persist_mk.py 12 14 16 18 20
for w in 32 64 256 512; do echo "== persistent, agent-scope barrier, $w workers"; SSF_PERSIST_MK=$w timeout 300 python tools/bench_persist_mk.py 12 14 16 18 20; done
for w in 16 32 64; do echo "== persistent, one XCD, $w workers"; SSF_PERSIST_MK=$w SSF_PERSIST_XCD=1 timeout 300 python tools/bench_persist_mk.py 12 14 16 18; done
unset SSF_COL_HALF
echo "== launch sequence, default geometry"; timeout 300 python tools/bench_persist_mk.py 12 14 16 18 20
} > $O/persist.txt 2>&1
cat $O/persist.txt
