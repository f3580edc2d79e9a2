#!/bin/bash
# round 4, batch 13: column tiles of the packed stage dealt so that tiles w and w + 32 of an XCD are column neighbours (pair) against
# contiguous runs per XCD (base); config 3, four interleaved repetitions
cd "$(dirname "$0")/.."
O=gpurun_out/r4m; mkdir -p $O
AB_CONFIGS="3" bash tools/gpu_ab.sh r4m/ab base pair base pair
