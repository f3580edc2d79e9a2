// Memory-system probes for MI355X (diagnostics; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/membw.hip -o tools/exp/membw
// 1. streaming copy / read-only kernels (grid-stride, 16 B per lane per access) over working sets
//    from 32 MiB to 2 GiB: what the memory system delivers when nothing else is in the way.
// 2. "burst" kernels shaped like the fused row kernel: every workgroup issues 16 x 16 B loads per
//    thread at once, then stores; one launch reads 32 MiB and writes 32 MiB, with 1, 2 or 4
//    workgroups resident per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_read(const double2 *__restrict__ a, double *out, size_t n) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 1.2345e300) out[0] = s;
}
// burst: workgroup w handles elements [w*256*E, (w+1)*256*E), thread t loads t + 256*q
template <int E> __global__ void __launch_bounds__(256) k_burst(const double2 *__restrict__ a, double2 *__restrict__ b, int spin) {
    const double2 *g = a + (size_t)blockIdx.x * 256 * E;
    double2 *o = b + (size_t)blockIdx.x * 256 * E;
    double2 v[E];
#pragma unroll
    for (int q = 0; q < E; ++q) v[q] = g[threadIdx.x + 256 * q];
    for (int r = 0; r < spin; ++r) {           // dependent FMA chain per element: ALU phase
#pragma unroll
        for (int q = 0; q < E; ++q) {
            v[q].x = v[q].x * 1.0000001 + v[(q + 1) % E].y * 1e-9;
            v[q].y = v[q].y * 0.9999999 + v[(q + 1) % E].x * 1e-9;
        }
    }
#pragma unroll
    for (int q = 0; q < E; ++q) o[threadIdx.x + 256 * q] = v[q];
}

// column-stage shape: a workgroup of 256 threads owns 8 adjacent columns (128 B) of a (256, pitch) array of
// double2 and reads / writes all 256 rows of them: 16 accesses per thread at a stride of `pitch` elements.
__global__ void __launch_bounds__(256) k_colshape(const double2 *__restrict__ a, double2 *__restrict__ b, int pitch) {
    const int c = threadIdx.x & 7, r0 = threadIdx.x >> 3;          // 8 columns x 32 rows per pass
    const size_t col = (size_t)blockIdx.x * 8 + c;
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = a[(size_t)(r0 + 32 * q) * pitch + col];
#pragma unroll
    for (int q = 0; q < 8; ++q) b[(size_t)(r0 + 32 * q) * pitch + col] = v[q];
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t maxb = (size_t)2 << 30;
    double2 *a, *b;
    double *out;
    CK(hipMalloc(&a, maxb));
    CK(hipMalloc(&b, maxb));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, maxb));
    CK(hipMemset(b, 0, maxb));
    printf("# streaming kernels: bytes moved / time\n");
    for (size_t mib : {16, 32, 64, 128, 256, 512, 1024, 2048}) {
        const size_t bytes = mib << 20, n = bytes / 16;
        for (int grid : {256 * 4, 256 * 8, 256 * 16}) {
            const int it = 20;
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
            CK(hipEventRecord(e0));
            for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            const double tc = time_ms(e0, e1) / it * 1e-3;
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, out, n);
            CK(hipEventRecord(e0));
            for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, out, n);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            const double tr = time_ms(e0, e1) / it * 1e-3;
            printf("size %5zu MiB grid %5d: copy %7.2f us = %6.0f GB/s (r+w)   read %7.2f us = %6.0f GB/s\n", mib, grid,
                   tc * 1e6, 2.0 * bytes / tc / 1e9, tr * 1e6, bytes / tr / 1e9);
        }
    }
    printf("# column-shaped access: 512 workgroups x (8 columns x 256 rows), 16 MiB in + 16 MiB out, row pitch varied\n");
    for (int pad : {0, 8, 16, 64, 136, 520}) {
        const int pitch = 4096 + pad, it = 50;
        for (int w = 0; w < 4; ++w) hipLaunchKernelGGL(k_colshape, dim3(512), dim3(256), 0, 0, w & 1 ? b : a, w & 1 ? a : b, pitch);
        CK(hipEventRecord(e0));
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k_colshape, dim3(512), dim3(256), 0, 0, i & 1 ? b : a, i & 1 ? a : b, pitch);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        const double t = time_ms(e0, e1) / it * 1e-3;
        printf("pitch 4096 + %3d elements: %7.2f us  %6.0f GB/s (r+w)\n", pad, t * 1e6, 2.0 * (16 << 20) / t / 1e9);
    }
    printf("# burst kernels (32 MiB in, 32 MiB out per launch, ping-pong a<->b like the row/column stages)\n");
    for (int spin : {0, 100, 200, 400}) {
        const int it = 50;
        // E = 16: 512 workgroups (LDS-free: occupancy limited by registers only)
        auto run = [&](auto kern, int nwg, size_t lds, const char *name) {
            for (int w = 0; w < 4; ++w) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, w & 1 ? b : a, w & 1 ? a : b, spin);
            (void)hipEventRecord(e0);
            for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, i & 1 ? b : a, i & 1 ? a : b, spin);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            const double t = time_ms(e0, e1) / it * 1e-3;
            printf("spin %3d  %-34s %7.2f us  %6.0f GB/s (r+w)\n", spin, name, t * 1e6, 2.0 * (32 << 20) / t / 1e9);
        };
        run(k_burst<16>, 512, 0, "16 el/thread, 512 wg, lds 0");
        run(k_burst<16>, 512, 70 * 1024, "16 el/thread, 512 wg, lds 70K (2/CU)");
        run(k_burst<16>, 512, 100 * 1024, "16 el/thread, 512 wg, lds 100K (1/CU)");
        run(k_burst<8>, 1024, 0, "8 el/thread, 1024 wg, lds 0");
        run(k_burst<8>, 1024, 36 * 1024, "8 el/thread, 1024 wg, lds 36K (4/CU)");
        run(k_burst<4>, 2048, 0, "4 el/thread, 2048 wg, lds 0");
    }
    return 0;
}
