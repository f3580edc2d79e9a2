#!/bin/bash
# same-box A/B of two library builds on the receiver-side calls (DeviceArray in / out): call times (two interleaved repetitions)
# and rocprofv3 kernel-trace summaries per case group.   bash tools/gpu_rx_ab.sh <out-tag> <tag> [<tag> ...]
# ("base" = the product build, others = opticommpy_amd/libssf_hip_<tag>.so)
cd "$(dirname "$0")/.."
REPO=$PWD
O=$REPO/gpurun_out/$1; mkdir -p $O; shift
L=$REPO/opticommpy_amd
lib() { [ "$1" = "base" ] && echo $L/libssf_hip.so || echo $L/libssf_hip_$1.so; }
export TMPDIR=/tmp
CASES=${AB_CASES:-pdm_notebook,pdm_impaired,pdm_defaults,firFilter255,decimate16to2,edc800km}
for rep in 1 2; do for t in "$@"; do
  SSF_LIB=$(lib $t) timeout 300 python tools/bench_rx_device.py 20 22 --reps 20 --cases $CASES > $O/${t}_calls_$rep.txt 2> $O/${t}_calls_$rep.err
  echo "== $t rep $rep"; cat $O/${t}_calls_$rep.txt
done; done | tee $O/summary.txt
cd /tmp
for t in "$@"; do for lg in 20 22; do for grp in ${AB_GROUPS:-firFilter255 pdm_notebook,pdm_defaults edc800km}; do
  g=${grp%%,*}
  SSF_LIB=$(lib $t) timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_${t}_${lg}_$g -o kt -- python $REPO/tools/bench_rx_device.py $lg --reps 10 --cases $grp > $O/kt_${t}_${lg}_$g.log 2>&1
  DB=$(find $O/kt_${t}_${lg}_$g -name '*.db' | head -1)
  echo "== kernels: $t 2^$lg $grp" >> $O/summary.txt
  python $REPO/tools/rocpd_stats.py "$DB" 2>&1 | grep -v "^TOTAL" | cut -c1-160 >> $O/summary.txt
  rm -rf $O/kt_${t}_${lg}_$g
done; done; done
tail -60 $O/summary.txt
