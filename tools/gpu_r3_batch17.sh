#!/bin/bash
cd "$(dirname "$0")/.."
REPO=$PWD; O=$REPO/gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt16 -o kt -- python $REPO/bench.py --log2n 16 --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-times > $O/kt16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktc1 -o kt -- python $REPO/bench.py --config 1 --steps 1000 --warmup 10 --no-cpu-baseline --no-kernel-times > $O/ktc1.log 2>&1
UNITS=16 STEPS=100 timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktu -o kt -- python $REPO/tools/bench_units.py 14 > $O/ktu.log 2>&1
cd $REPO
for d in kt16 ktc1 ktu; do python tools/rocpd_stats.py $(find $O/$d -name '*.db' | head -1) > $O/${d}_stats.txt 2>&1; head -8 $O/${d}_stats.txt | cut -c1-170; done
find $O -name '*.db' -exec gzip -f {} \;
