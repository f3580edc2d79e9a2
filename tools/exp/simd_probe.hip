// Which SIMD do the waves of a workgroup land on, and do the two workgroups that share a CU put their wave k on the same SIMD?
// (mixed-radix rows: a pass with a large radix has butterflies for the first two waves of a row only -- if those sit on the same two
// SIMDs in both workgroups of a CU, the pass is issue-bound on half of the CU.)  Part 1 records HW_ID of every wave; part 2 times
// a launch in which only two waves of every workgroup do FP64 work: waves {0,1} everywhere / {0,1} or {2,3} by (bid / 256) & 1 /
// {0,1} or {2,3} by the workgroup's TG_ID bit 0 / all four waves with half the work each.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>

struct Rec { unsigned hw_id, xcc_id; };

__global__ void __launch_bounds__(256, 2) probe(Rec *out) {
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double acc = (double)threadIdx.x;
    for (int i = 0; i < 4000; ++i) acc = acc * 1.0000001 + 0.5;
    lds[threadIdx.x] = (char)acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = Rec{hw + (unsigned)(lds[1] & 0), xcc};
}

// mode 0: waves 0,1 work; 1: by (bid / 256) & 1; 2: by TG_ID & 1; 3: all four waves, half the iterations each; 4: by SIMD id (waves on SIMD 0,1 for even TG, 2,3 for odd)
__global__ void __launch_bounds__(256, 2) work(double *out, int mode, int iters) {
    extern __shared__ char lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int wave = threadIdx.x / 64;
    const int tg = (hw >> 16) & 15, simd = (hw >> 4) & 3;
    int par = 0, n = iters;
    bool act;
    if (mode == 1) par = (blockIdx.x / 256) & 1;
    if (mode == 2 || mode == 4) par = tg & 1;
    if (mode == 3) { act = true; n = iters / 2; }
    else if (mode == 4) act = (simd >> 1) == par;
    else act = (wave >> 1) == par;
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    if (act)
        for (int i = 0; i < n; ++i) {
            a0 = a0 * 1.0000001 + 0.5; a1 = a1 * 1.0000001 + 0.5; a2 = a2 * 1.0000001 + 0.5; a3 = a3 * 1.0000001 + 0.5;
        }
    lds[threadIdx.x] = (char)(a0 + a1 + a2 + a3);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = a0 + lds[5];
}

int main() {
    const int nwg = 512, ldsb = 69632;
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * nwg * 4);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipFuncSetAttribute((const void *)work, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    std::vector<Rec> r(nwg * 4);
    for (int rep = 0; rep < 2; ++rep) { probe<<<nwg, 256, ldsb>>>(d); hipDeviceSynchronize(); }
    hipMemcpy(r.data(), d, sizeof(Rec) * nwg * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    int wave_eq_simd = 0, same_simd_pairs = 0, pairs = 0, tg_differs = 0, bid256 = 0, cus2 = 0;
    for (int b = 0; b < nwg; ++b) {
        const unsigned hw = r[b * 4].hw_id, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = r[b * 4].xcc_id & 15;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        for (int w = 0; w < 4; ++w) wave_eq_simd += (int)((r[b * 4 + w].hw_id >> 4) & 3) == w;
    }
    int n = 0;
    for (auto &kv : by_cu) {
        if (n++ < 12) {
            printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
            for (int b : kv.second) {
                printf("  bid %3d tg %u simd of waves 0-3:", b, (r[b * 4].hw_id >> 16) & 15);
                for (int w = 0; w < 4; ++w) printf(" %u", (r[b * 4 + w].hw_id >> 4) & 3);
            }
            printf("\n");
        }
        if (kv.second.size() == 2) {
            ++cus2;
            const int a = kv.second[0], b = kv.second[1];
            bid256 += std::abs(a - b) == 256;
            tg_differs += (((r[a * 4].hw_id >> 16) & 1) != ((r[b * 4].hw_id >> 16) & 1));
            for (int w = 0; w < 4; ++w) { ++pairs; same_simd_pairs += ((r[a * 4 + w].hw_id >> 4) & 3) == ((r[b * 4 + w].hw_id >> 4) & 3); }
        }
    }
    printf("# CUs used: %zu (with two workgroups: %d); wave k on SIMD k: %d of %d waves; co-resident workgroups: wave k of both on the same SIMD in %d of %d cases,\n"
           "#   bids 256 apart in %d of %d CUs, TG_ID bit 0 differs in %d of %d CUs\n", by_cu.size(), cus2, wave_eq_simd, nwg * 4, same_simd_pairs, pairs, bid256, cus2, tg_differs, cus2);
    double *o;
    hipMalloc(&o, sizeof(double) * nwg);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 3; ++rep) work<<<nwg, 256, ldsb>>>(o, mode, 20000);
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; ++rep) work<<<nwg, 256, ldsb>>>(o, mode, 20000);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %.1f us per launch\n", mode, ms * 100.0);
    }
    return 0;
}
