#!/bin/bash
# Final check of a round on a GPU box: the GPU suite, the driver's command three times (with its "also" legs), the default bench line, smoke().
#   gpurun --timeout 2400 -- 'bash tools/gpu_final_check.sh r6_final2'
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-final}; mkdir -p $O
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
summ() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); a = d.get("also", {})
rx = a.get("rx_chain_2^20", {}); nb = a.get("reference_notebook", {}).get("lengths", {})
print(sys.argv[1], round(d["value"], 1), round(d["roofline"]["frac"], 4), "rx chain", rx.get("ms_per_chain"), rx.get("ms_per_chain_four_calls"),
      "notebook", {k: round(v["wall_s_device_resident"], 4) for k, v in nb.items()}, "errors", [k for k, v in a.items() if "error" in v])
PY
}
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$i.json 2> $O/driver_cmd_$i.err; summ $O/driver_cmd_$i.json; done
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; summ $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
