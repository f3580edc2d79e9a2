// Micro-benchmark: can ONE wave per SIMD hide its LDS exchanges behind its own arithmetic?  (a row launch's
// arithmetic phase takes VALU time + LDS time with two workgroups per CU relying on the hardware to interleave them.)
// Shapes of the row kernel's transform phase: 256 threads, 16 complex128 values per thread and row, per pass ~NF double-precision
// FMAs per value-pair, then a 64 KiB exchange (16 ds_write_b128, barrier, 16 ds_read_b128 with a transposing index).
//   mode 0: one row per workgroup, two workgroups per CU (grid 512)                 -- the product kernel's structure
//   mode 1: two rows per workgroup, one workgroup per CU (grid 256), software-pipelined: row B's arithmetic between row A's
//           stores and the barrier, row A's next pass while row B's stores drain
//   mode 2: mode 0 without the exchanges (VALU only)    mode 3: mode 0 without the arithmetic (LDS only)
//   mode 4: mode 1's work in mode 1's geometry, NOT interleaved (A then B)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/lds_overlap tools/exp/lds_overlap.hip ; run on a GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

struct c2 { double re, im; };
#define V 16
#ifndef NF
#define NF 12
#endif

__device__ __forceinline__ void arith(c2 *v, double a, double b) {
#pragma unroll
    for (int r = 0; r < NF; ++r) {
#pragma unroll
        for (int i = 0; i < V; i += 2) {                       // butterfly-like: 8 independent pairs, 8 FMAs each
            const c2 x = v[i], y = v[i + 1];
            v[i].re = fma(y.re, a, x.re) - y.im * b;
            v[i].im = fma(y.im, a, x.im) + y.re * b;
            v[i + 1].re = fma(-y.re, a, x.re) + y.im * b;
            v[i + 1].im = fma(-y.im, a, x.im) - y.re * b;
        }
    }
}
__device__ __forceinline__ void put(const c2 *v, c2 *lds, int t) {
#pragma unroll
    for (int i = 0; i < V; ++i) lds[i * 256 + t] = v[i];
}
__device__ __forceinline__ void get(c2 *v, const c2 *lds, int t) {
    const int base = (t >> 4) * 256 + (t & 15);               // transposing read: 16-element stride
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = lds[base + 16 * i];
}

template <int MODE> __global__ void __launch_bounds__(256, MODE == 1 || MODE == 4 ? 1 : 2) k(c2 *out, int rounds, double a, double b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c2 *lA = (c2 *)smem, *lB = lA + 4096;
    const int t = threadIdx.x;
    c2 A[V], B[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        A[i] = {1.0 + t * 1e-3 + i, 0.5 - i * 1e-2};
        B[i] = {0.3 + t * 1e-3 - i, 0.25 + i * 1e-2};
    }
    if (MODE == 0 || MODE == 2 || MODE == 3) {
        for (int r = 0; r < rounds; ++r) {
            if (MODE != 3) arith(A, a, b);
            if (MODE != 2) {
                put(A, lA, t);
                __syncthreads();
                get(A, lA, t);
            }
        }
    } else if (MODE == 4) {
        for (int r = 0; r < rounds; ++r) {
            arith(A, a, b);
            put(A, lA, t);
            __syncthreads();
            get(A, lA, t);
            arith(B, a, b);
            put(B, lB, t);
            __syncthreads();
            get(B, lB, t);
        }
    } else {
        arith(A, a, b);
        for (int r = 0; r < rounds; ++r) {
            put(A, lA, t);                                     // A's stores drain ...
            __builtin_amdgcn_sched_barrier(0);
            arith(B, a, b);                                    // ... behind B's arithmetic
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            get(A, lA, t);
            put(B, lB, t);
            __builtin_amdgcn_sched_barrier(0);
            arith(A, a, b);                                    // A's next pass while B's stores drain
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            get(B, lB, t);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < V; ++i) s += A[i].re + A[i].im + ((MODE == 1 || MODE == 4) ? B[i].re + B[i].im : 0.0);
    if (s == 12345.678) out[blockIdx.x * 256 + t] = {s, s};
}

template <int MODE> double run(c2 *out, int grid, size_t lds, int rounds) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, 256, lds>>>(out, rounds, 0.999, 0.01);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE><<<grid, 256, lds>>>(out, rounds, 0.999, 0.01);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0 * 1e3;                                     // us per launch
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    c2 *out;
    hipMalloc(&out, sizeof(c2) * 512 * 256);
    const double t0 = run<0>(out, 512, 65536, rounds), t2 = run<2>(out, 512, 65536, rounds), t3 = run<3>(out, 512, 65536, rounds);
    const double t1 = run<1>(out, 256, 131072, rounds), t4 = run<4>(out, 256, 131072, rounds);
    printf("NF=%d, %d rounds (one round = one pass of arithmetic + one 64 KiB exchange per row; 512 rows in every mode)\n", NF, rounds);
    printf("us per round and CU (two rows):\n");
    printf("  mode 0  1 row / WG, 2 WG / CU (product structure)     %.3f\n", t0 / rounds);
    printf("  mode 2  ... arithmetic only                            %.3f\n", t2 / rounds);
    printf("  mode 3  ... exchanges only                             %.3f\n", t3 / rounds);
    printf("  mode 1  2 rows / WG, 1 WG / CU, software-pipelined     %.3f\n", t1 / rounds);
    printf("  mode 4  2 rows / WG, 1 WG / CU, not interleaved        %.3f\n", t4 / rounds);
    return 0;
}
