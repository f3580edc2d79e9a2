#!/bin/bash
# Round 6 in one call on a GPU box: the GPU suite, the driver's command three times, the default bench line (1001-step span + "also" legs),
# kernel-trace + traffic profiles of configs 2 / 3 (tools/profile_round.sh), the rocprofv3 summary of the driver's own command, of the
# reference notebook's benchmark (bench.py --notebook: adaptive step, 2e5 / 8e5 / 2e6 samples) and of the receiver chain, the receiver /
# transmitter call times.   gpurun --timeout 3000 -- 'bash tools/gpu_round6.sh r6'   -> gpurun_out/<tag>_*
cd "$(dirname "$0")/.."
REPO=$PWD; TAG=${1:-r6}; O=$REPO/gpurun_out/${TAG}_final; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$i.json 2> $O/driver_cmd_$i.err; python - <<PY
import json; d=json.loads(open("$O/driver_cmd_$i.json").read().strip().splitlines()[-1]); print("driver cmd $i:", round(d["value"],1), round(d["roofline"]["frac"],4), {k: round(v["avg_us"],2) for k,v in d["roofline"]["kernels"].items() if isinstance(v,dict) and "avg_us" in v})
PY
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json; d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); print("default:", round(d["value"],1), round(d["roofline"]["frac"],4), d["cpu_baseline"]["value"]); a=d["also"]
for k,v in a.items():
    print(" ", k, {x: (round(y,4) if isinstance(y,float) else y) for x,y in v.items() if x in ("value","roofline_frac","ms_per_chain","ms_per_chain_four_calls","error")})
PY
bash tools/profile_round.sh ${TAG}_prof 2 200 > $O/profile_c2.log 2>&1; tail -8 $O/profile_c2.log | cut -c1-200
bash tools/profile_round.sh ${TAG}_prof 3 100 > $O/profile_c3.log 2>&1; tail -8 $O/profile_c3.log | cut -c1-200
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktd -o kt -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-also > $O/ktd.log 2>&1
python $REPO/tools/rocpd_stats.py "$(find $O/ktd -name '*.db' | head -1)" > $O/driver_cmd_kernel_stats.txt 2>&1; rm -rf $O/ktd
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ktn -o kt -- python $REPO/bench.py --notebook > $O/ktn.log 2>&1
python $REPO/tools/rocpd_stats.py "$(find $O/ktn -name '*.db' | head -1)" > $O/notebook_kernel_stats.txt 2>&1; rm -rf $O/ktn
for m in four one; do timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktc_$m -o kt -- python $REPO/tools/bench_rx_chain.py --reps 20 --mode $m > $O/ktc_$m.log 2>&1; python $REPO/tools/rocpd_stats.py "$(find $O/ktc_$m -name '*.db' | head -1)" > $O/rx_chain_kernel_stats_$m.txt 2>&1; rm -rf $O/ktc_$m; done
cd $REPO
python tools/bench_rx_chain.py --reps 50 > $O/rx_chain_calls.txt 2>&1; cat $O/rx_chain_calls.txt
python tools/bench_rx_device.py 20 22 --reps 20 > $O/rx_tx_calls.txt 2> $O/rx_tx_calls.err; cat $O/rx_tx_calls.txt
python tools/bench_lengths.py 48000 240000 960000 1440000 2000000 > $O/lengths.txt 2>&1; cat $O/lengths.txt
head -12 $O/driver_cmd_kernel_stats.txt | cut -c1-170; head -14 $O/notebook_kernel_stats.txt | cut -c1-170
