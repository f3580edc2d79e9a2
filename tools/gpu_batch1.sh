#!/bin/bash
# round-2 GPU batch 1: parity of the refactored / packed kernels, memory-policy A/B, config-3 shape, drift, profiles
cd "$(dirname "$0")/.."
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1   # page the image in (bench.py at N=1 does not import torch)
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
L=opticommpy_amd
B="python bench.py --no-cpu-baseline --no-kernel-times --warmup 20"
for v in "" _mp1 _mp3 _mp7 _mp15 ""; do
  SSF_LIB=$PWD/$L/libssf_hip$v.so timeout 120 $B --steps 400 > $O/c2$v.json 2> $O/c2$v.err
done
SSF_LIB=$PWD/$L/libssf_hip.so timeout 120 python bench.py --no-cpu-baseline --warmup 20 --steps 200 > $O/c2_kt.json 2>&1
# config-3 shape: 2^22 complex64
C3="$B --log2n 22 --prec c64 --steps 200"
timeout 200 $C3 > $O/c3_packed.json 2> $O/c3_packed.err
SSF_C64_PACKED=0 timeout 200 $C3 > $O/c3_unpacked.json 2> $O/c3_unpacked.err
SSF_LIB=$PWD/$L/libssf_hip_hilo0.so timeout 200 $C3 > $O/c3_packed_hilo0.json 2> $O/c3_hilo0.err
SSF_SPLIT_L1=10 timeout 200 $C3 > $O/c3_packed_l1_10.json 2> $O/c3_l1_10.err
SSF_LIB=$PWD/$L/libssf_hip_mp3.so timeout 200 $C3 > $O/c3_packed_mp3.json 2> $O/c3_mp3.err
SSF_LIB=$PWD/$L/libssf_hip_mp15.so timeout 200 $C3 > $O/c3_packed_mp15.json 2> $O/c3_mp15.err
timeout 120 $B --log2n 20 --prec c64 --steps 400 > $O/c64_20_packed.json 2>&1
SSF_C64_PACKED=0 timeout 120 $B --log2n 20 --prec c64 --steps 400 > $O/c64_20_unpacked.json 2>&1
timeout 120 $B --log2n 21 --prec c128 --steps 200 > $O/c128_21.json 2>&1
timeout 120 $B --log2n 22 --prec c128 --steps 200 > $O/c128_22.json 2>&1
# drift over config 3's step count at 2^18, and one span at 2^22
timeout 300 python tests/tools/c64_drift.py 18 800 > $O/drift_18_packed.log 2>&1
SSF_C64_PACKED=0 timeout 300 python tests/tools/c64_drift.py 18 800 > $O/drift_18_unpacked.log 2>&1
timeout 300 python tests/tools/c64_drift.py 22 80 > $O/drift_22_packed.log 2>&1
# kernel traces
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/kt_c3 -o kt -- python $OLDPWD/bench.py --no-cpu-baseline --no-kernel-times --log2n 22 --prec c64 --steps 100 --warmup 10 > $OLDPWD/$O/kt_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/kt_c2 -o kt -- python $OLDPWD/bench.py --no-cpu-baseline --no-kernel-times --steps 200 --warmup 20 > $OLDPWD/$O/kt_c2.log 2>&1
cd $OLDPWD
for d in kt_c3 kt_c2; do python tools/rocpd_stats.py $(find $O/$d -name "*.db" | head -1) > $O/$d.stats.txt 2>&1; done
find $O -name "*.db" -size +20M -delete
tail -3 $O/pytest.log; for f in $O/c2*.json $O/c3*.json $O/c64*.json $O/c128*.json; do echo "$f: $(python - <<PY
import json,sys
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); print(round(d["value"],1), d["unit"], "frac", round(d["roofline"]["frac"],4), "it/step", d["config"]["iterations_per_step"])
except Exception as e: print("ERR", e)
PY
)"; done
cat $O/drift_*.log | grep -v "^complex"
