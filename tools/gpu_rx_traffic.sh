#!/bin/bash
# HBM-side traffic of the receiver-side kernels (two separate PMC passes, FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md's HBM section
# prescribes; tools/traffic_from_pmc.py has the calibration) per launch, next to the algorithmic bytes.   bash tools/gpu_rx_traffic.sh <tag> [log2n]
cd "$(dirname "$0")/.."
REPO=$PWD; TAG=${1:-r5}; LG=${2:-22}
O=$REPO/gpurun_out/${TAG}_rxpmc; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
CASES=firFilter255,edc800km,pdm_notebook,pdm_defaults,photodiode,decimate16to2
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o pf -- python $REPO/tools/bench_rx_device.py $LG --reps 4 --cases $CASES > $O/pf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o pw -- python $REPO/tools/bench_rx_device.py $LG --reps 4 --cases $CASES > $O/pw.log 2>&1
python - "$(find $O/pf -name '*.db' | head -1)" "$(find $O/pw -name '*.db' | head -1)" $LG > $O/traffic.txt <<'PY'
import sqlite3, sys
def per_kernel(path, counter):
    db = sqlite3.connect(path)
    return {n: (c, v) for n, c, v in db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,))}
f, w, lg = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
print("# 2^%d samples x 2 columns, complex128; per launch: reads = FETCH_SIZE x 2 KiB, writes = WRITE_SIZE KiB (MI355X_MICROARCH.md, HBM section)" % lg)
print("%-78s %6s %12s %12s %12s" % ("kernel", "calls", "read MiB", "written MiB", "total MiB"))
for name in sorted(set(f) | set(w)):
    if "copyBuffer" in name or "fillBuffer" in name:
        continue
    c = f.get(name, w.get(name))[0]
    rd = 2.0 * f.get(name, (1, 0))[1] / c / 1024.0
    wr = w.get(name, (1, 0))[1] / w.get(name, (1, 0))[0] / 1024.0
    print("%-78s %6d %12.1f %12.1f %12.1f" % (name.replace("ssf::(anonymous namespace)::", "").replace("void ", "")[:78], c, rd, wr, rd + wr))
PY
rm -rf $O/pf $O/pw
cat $O/traffic.txt
