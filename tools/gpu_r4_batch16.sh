#!/bin/bash
# round 4, batch 16: one rolled make_linop per row kernel + compact single-precision sin(d/2) fallback (base) against the build before (cis); configs 2 and 3
cd "$(dirname "$0")/.."
O=gpurun_out/r4q; mkdir -p $O
bash tools/gpu_ab.sh r4q/ab cis base cis base
