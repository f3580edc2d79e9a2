#!/bin/bash
# round 4, batch 15: compact large-angle path of the single-precision cis (base) against sincosf's own (cisold); config 3
cd "$(dirname "$0")/.."
O=gpurun_out/r4p; mkdir -p $O
AB_CONFIGS="3" bash tools/gpu_ab.sh r4p/ab cisold base cisold base
