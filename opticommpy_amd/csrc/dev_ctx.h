// Execution context handed to the kernel bodies on the GPU (the CPU tests hand them the fiber
// emulator's context instead).  HIP only.
#pragma once
#include <hip/hip_runtime.h>

namespace ssf {

struct DevCtxCore {
    int tid, bid, nthreads, nblocks;
    char *lds;
    static constexpr bool kWaveOps = true;
    __device__ __forceinline__ void sync() { __syncthreads(); }
    // keeps the instruction scheduler from moving memory operations across this point
    __device__ __forceinline__ void issue_fence() { __builtin_amdgcn_sched_barrier(0); }
    // issue priority of this wave against the other waves of its SIMD (0 .. 3); the hardware default is 0
    template <int P> __device__ __forceinline__ void setprio() { __builtin_amdgcn_s_setprio(P); }
    // all-lanes butterfly over the 64-lane wave (every lane returns the same value)
    __device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    // the value the neighbouring lane (lane ^ 1) passes: two DPP moves (quad_perm [1, 0, 3, 2]), no LDS, no barrier
    __device__ __forceinline__ double xchg(double v) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ double wave_max(double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        return v;
    }
};

}  // namespace ssf
