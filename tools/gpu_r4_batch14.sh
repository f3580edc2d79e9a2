#!/bin/bash
# round 4, batch 14: the 64-byte-segment loads of the packed column stage (G, E_hd) with cache-policy bits: sc0, sc0 sc1 against plain (base)
cd "$(dirname "$0")/.."
O=gpurun_out/r4o; mkdir -p $O
AB_CONFIGS="3" bash tools/gpu_ab.sh r4o/ab base sc0 sc01
