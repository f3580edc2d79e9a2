// Micro-benchmark: what does a kernel's argument fetch cost at launch start?  512 workgroups x 256 threads copy 16 x 16 B per
// thread (the row stage's memory shape, 32 MiB in + 32 MiB out), back-to-back dependent launches.
//   big    : the pointers sit at the END of a 448-byte by-value struct (the product kernels' argument block), first use = after
//            a scalar-load round trip
//   direct : the pointers are plain kernel arguments
//   preload: ... compiled with -mllvm -amdgpu-kernarg-preload-count=8 (the first arguments arrive in SGPRs with the wave)
// Build: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=8 -DPRELOAD] -o tools/exp/kernarg_latency[_pre] tools/exp/kernarg_latency.hip
#include <hip/hip_runtime.h>

#include <cstdio>

struct Big {
    double pad[52];
    const double2 *src;
    double2 *dst;
    int n;
};
__global__ void __launch_bounds__(256) k_big(const Big a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = a.src[i + (size_t)q * a.n];
#pragma unroll
    for (int q = 0; q < 16; ++q) a.dst[i + (size_t)q * a.n] = v[q];
}
__global__ void __launch_bounds__(256) k_direct(const double2 *src, double2 *dst, int n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = src[i + (size_t)q * n];
#pragma unroll
    for (int q = 0; q < 16; ++q) dst[i + (size_t)q * n] = v[q];
}

int main() {
    const int n = 512 * 256;
    double2 *a, *b;
    hipMalloc(&a, sizeof(double2) * n * 16);
    hipMalloc(&b, sizeof(double2) * n * 16);
    hipMemset(a, 0, sizeof(double2) * n * 16);
    Big big{};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        float ms1 = 0, ms2 = 0;
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            for (int i = 0; i < 200; ++i) {
                big.src = (i & 1) ? b : a;
                big.dst = (i & 1) ? a : b;
                big.n = n;
                k_big<<<512, 256>>>(big);
            }
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms1, e0, e1);
            hipEventRecord(e0);
            for (int i = 0; i < 200; ++i) k_direct<<<512, 256>>>((i & 1) ? b : a, (i & 1) ? a : b, n);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms2, e0, e1);
        }
#ifdef PRELOAD
        printf("preload build: 448-byte struct %.3f us / launch   direct (preloaded) arguments %.3f us / launch\n", ms1 * 5, ms2 * 5);
#else
        printf("plain build:   448-byte struct %.3f us / launch   direct arguments %.3f us / launch\n", ms1 * 5, ms2 * 5);
#endif
    }
    return 0;
}
