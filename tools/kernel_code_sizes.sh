#!/bin/bash
# Code size (bytes) of every kernel of the fused engine's two translation units, largest first; works without a GPU (~4 min).
# The instruction cache is 64 KB per two CUs and every step alternates a row and a column kernel: see DESIGN.md 3.7 / 4.45.
#   bash tools/kernel_code_sizes.sh [extra hipcc flags]        e.g.  bash tools/kernel_code_sizes.sh | grep -E "k_col_pk<10>|k_col<double, 8, 3>|, 256, 2, 12>"
cd "$(dirname "$0")/../opticommpy_amd/csrc"
T=$(mktemp -d)
for u in f64 f32; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-pass-failed -mllvm -amdgpu-use-amdgpu-trackers=1 --cuda-device-only "$@" -c engine_fused_$u.hip -o $T/$u.o 2>/dev/null
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/$u.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$u.co 2>/dev/null
  /opt/rocm/lib/llvm/bin/llvm-readelf -sW $T/$u.co 2>/dev/null | awk '$4=="FUNC" {print $3, $8}' | sort -rn | c++filt |
    sed 's/ssf::(anonymous namespace):://; s/(ssf::fused::.*//; s/void //' | awk '{s=$1; $1=""; print s, $0}' | uniq
done
rm -rf $T
