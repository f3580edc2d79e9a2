#!/bin/bash
# HBM traffic of the kernels a given length runs on (tools/profile_length.py N as the workload: three calls of 200 fixed steps, 3 iterations
# each), FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, corrected as tools/traffic_from_pmc.py does for the BASELINE configs.
#   gpurun --timeout 900 -- 'bash tools/gpu_traffic_length.sh 2000000 r6_traffic_2e6'
cd "$(dirname "$0")/.."
REPO=$PWD; N=${1:-2000000}; O=$REPO/gpurun_out/${2:-traffic_$N}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o pf -- python $REPO/tools/profile_length.py $N > $O/pf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o pw -- python $REPO/tools/profile_length.py $N > $O/pw.log 2>&1
cd $REPO
PF=$(find $O/pf -name '*.db' | head -1); PW=$(find $O/pw -name '*.db' | head -1)
ALG=$((512 * N))
python tools/traffic_from_pmc.py "$PF" "$PW" 600 fused > $O/traffic.txt 2>&1
echo "algorithmic bytes per step at N = $N, 3 iterations: $ALG" >> $O/traffic.txt
find $O -name '*.db' -delete
cat $O/traffic.txt | cut -c1-200
