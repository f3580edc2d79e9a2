// Where does the hardware put the workgroups of a launch shaped like the packed column stage (1024 workgroups of 256 threads,
// ~70 KiB of LDS each: two per CU, two rounds)?  Records XCC / SE / SH / CU of every workgroup and the time it started, so that the
// column-tile map can be checked against the dispatch order (neighbouring tiles share 128-byte lines: do they share an L1 / a CU?).
//   hipcc --offload-arch=gfx950 -O2 tools/exp/dispatch_order.hip -o tools/exp/dispatch_order && tools/exp/dispatch_order [nwg] [lds_bytes] [spin]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>

struct Rec { unsigned hw_id, xcc_id; unsigned long long t0, t1; };

__global__ void __launch_bounds__(256, 2) probe(Rec *out, int spin) {
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = wall_clock64();
    float acc = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0000001f + 0.5f;          // hold the slot for a while
    lds[threadIdx.x] = (char)acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{hw, xcc, t0, wall_clock64() + (unsigned long long)(lds[1] & 0)};
}

int main(int argc, char **argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 1024, ldsb = argc > 2 ? atoi(argv[2]) : 71168, spin = argc > 3 ? atoi(argv[3]) : 20000;
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * nwg);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    std::vector<Rec> r(nwg);
    for (int rep = 0; rep < 3; ++rep) {
        probe<<<nwg, 256, ldsb>>>(d, spin);
        hipDeviceSynchronize();
    }
    hipMemcpy(r.data(), d, sizeof(Rec) * nwg, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (auto &x : r) tmin = std::min(tmin, x.t0);
    printf("# bid xcc se sh cu  t0 t1 (100 MHz ticks from the first start)   [hw_id fields: cu 11:8, sh 12, se 15:13]\n");
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < nwg; ++b) {
        const unsigned hw = r[b].hw_id, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = r[b].xcc_id & 15;
        if (b < 160 || b % 8 == 0) printf("%4d %2u %u %u %2u  %6llu %6llu\n", b, xcc, se, sh, cu, r[b].t0 - tmin, r[b].t1 - tmin);
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    printf("# distinct (xcc, se, sh, cu): %zu\n", by_cu.size());
    printf("# workgroups per CU, in bid order (w = bid / 8 in brackets), first 40 CUs:\n");
    int n = 0;
    for (auto &kv : by_cu) {
        if (n++ >= 40) break;
        printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
        for (int b : kv.second) printf(" %d[%d]", b, b / 8);
        printf("\n");
    }
    // how often do w and w + 1 (same XCD) share a CU?  w and w + d for other d?
    std::vector<unsigned> cu_of(nwg);
    for (auto &kv : by_cu) for (int b : kv.second) cu_of[b] = kv.first;
    for (int d : {1, 2, 3, 4, 8, 16, 28, 30, 32, 36, 64}) {
        int same = 0, tot = 0;
        for (int b = 0; b + 8 * d < nwg; ++b) { ++tot; same += cu_of[b] == cu_of[b + 8 * d]; }
        printf("# w and w + %2d on the same CU: %d of %d\n", d, same, tot);
    }
    {   // pairs (2k, 2k+1) in w -- the tiles that share lines under the product's map
        int same = 0, tot = 0;
        for (int b = 0; b + 8 < nwg; ++b) if (((b / 8) & 1) == 0) { ++tot; same += cu_of[b] == cu_of[b + 8]; }
        printf("# line-sharing pairs (w even, w + 1) on the same CU: %d of %d\n", same, tot);
    }
    return 0;
}
