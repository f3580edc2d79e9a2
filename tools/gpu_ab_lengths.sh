#!/bin/bash
# same-box A/B of library builds at notebook lengths (mixed-radix stages): bash tools/gpu_ab_lengths.sh <out-tag> "<N ...>" <tag> [<tag> ...]
# ("base" = the product build; others = libssf_hip_<tag>.so from `make variant TAG=<tag> VFLAGS=...`); two interleaved repetitions
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O; NS=$2; shift; shift
L=$PWD/opticommpy_amd
lib() { [ "$1" = "base" ] && echo $L/libssf_hip.so || echo $L/libssf_hip_$1.so; }
for rep in 1 2; do for n in $NS; do for t in "$@"; do
  echo "== $t rep $rep: $(SSF_LIB=$(lib $t) python tools/profile_length.py $n 2>&1 | tr '\n' ' ' | sed 's/ launches,//g; s/TB\/s algorithmic = //g')"
done; done; done | tee $O/summary.txt
