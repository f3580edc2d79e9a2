// valu_rates.hip -- issue cost (cycles per wave-instruction) of the vector instructions the complex64 kernels are made of:
// packed FP32, FP64, the FP64 <-> FP32 conversions of the hi + lo split, plain FP32 and moves.  One or two waves per SIMD,
// eight independent chains per wave, s_memtime around 64 x 256 instructions.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/valu_rates.hip -o tools/exp/valu_rates && tools/exp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ void __launch_bounds__(1024) k(unsigned long long *out, int iters) {
    double d[8];
    float f[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) {
        d[i] = 1.0 + 1e-9 * (threadIdx.x + i);
        f[i] = 1.0f + 1e-4f * (threadIdx.x + i);
        p[i] = f2{f[i], f[i] * 0.5f};
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define ONE(i)                                                                                              \
    if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));          \
    if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));              \
    if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));              \
    if (OP == 3) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));             \
    if (OP == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));                 \
    if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));                 \
    if (OP == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));                           \
    if (OP == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));                           \
    if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));             \
    if (OP == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));                     \
    if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));       \
    if (OP == 11) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]) : "vcc"); \
    if (OP == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));             \
    if (OP == 13) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
            REP8(ONE)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + f[i] + p[i].x + p[i].y;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 12345.678) out[0] = 0;
}

template <int OP> void run(const char *name, unsigned long long *dout) {
    for (int threads : {256, 512, 1024, 2048}) {
        const int iters = 256;
        const int blocks = threads > 1024 ? 512 : 256, tpb = threads > 1024 ? 1024 : threads;       // 2048: two 1024-thread blocks per CU
        k<OP><<<blocks, tpb>>>(dout, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        k<OP><<<blocks, tpb>>>(dout, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto x : h) avg += (double)x;
        avg /= 256.0;
        // s_memtime counts at 100 MHz on gfx950?  report raw ticks per instruction as well as the ratio to v_fma_f32
        // per SIMD: (threads / 256) waves, each issuing iters * 64 instructions, in `ms` of wall time (launch overhead ~ 5 us included)
        printf("%-16s %d waves/SIMD: %.3f ticks per instruction per wave;  wall %.1f us -> %.2f ns per wave-instruction per SIMD\n", name,
               threads / 256, avg / (iters * 64.0), ms * 1e3, ms * 1e6 / ((double)(threads / 256) * iters * 64.0));
    }
}

int main() {
    unsigned long long *dout;
    hipMalloc(&dout, 512 * 8);
    run<8>("v_fma_f32", dout);
    run<0>("v_pk_fma_f32", dout);
    run<1>("v_pk_add_f32", dout);
    run<2>("v_pk_mul_f32", dout);
    run<3>("v_fma_f64", dout);
    run<13>("v_fmac_f64", dout);
    run<4>("v_add_f64", dout);
    run<5>("v_mul_f64", dout);
    run<6>("v_cvt_f32_f64", dout);
    run<7>("v_cvt_f64_f32", dout);
    run<9>("v_mov_b32", dout);
    run<10>("v_cndmask_b32", dout);
    run<11>("v_add_co_u32", dout);
    run<12>("v_mul_lo_u32", dout);
    return 0;
}
