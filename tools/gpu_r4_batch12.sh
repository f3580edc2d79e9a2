#!/bin/bash
# round 4, batch 12: E_hd of the packed column stage by owner thread (base) against by sample (ehdold); config 3, four interleaved repetitions
cd "$(dirname "$0")/.."
O=gpurun_out/r4l; mkdir -p $O
AB_CONFIGS="3" bash tools/gpu_ab.sh r4l/ab ehdold base ehdold base
