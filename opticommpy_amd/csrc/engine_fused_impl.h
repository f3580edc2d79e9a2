// engine_fused_impl.h -- HIP backend of the fused engine: __global__ wrappers around the kernel bodies of fused_kernels.h, the
// Backend that FusedCore (fused_engine.h) drives, the engine / convolution / overlap-save classes.  Everything is a template
// or lives in an anonymous namespace: the file is included by one translation unit per precision (engine_fused_f64.hip,
// engine_fused_f32.hip), which are compiled in parallel, and SSF_FUSED_PRECISION_TAG names that unit's entry points.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "dev_ctx.h"
#include "fused_engine.h"
#include "ssf_internal.h"

namespace ssf {
namespace {

using namespace fused;

// Phase timing (diagnostic builds only, make PHASE=1): mark(i) drains the memory counters and
// stores the 100 MHz wall clock for the lead thread of every workgroup; the product build
// compiles mark() to nothing.
#ifdef SSF_PHASE_TIMING
__device__ unsigned long long g_marks[4][4096][8];
#endif

struct DevCtx : DevCtxCore {
#ifdef SSF_PHASE_TIMING
    int kind;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // stamps stay in registers; flush(set) stores the complete record of this launch, so a
    // launch of another kind (early return, no forward transform) never mixes into it
    __device__ __forceinline__ void mark(int i) {
        __builtin_amdgcn_sched_barrier(0);
#if SSF_PHASE_TIMING != 2                                          // (2: stamps only, phases overlap as in the product build)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        ts[i] = wall_clock64();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void flush(int set) {
        if (tid == 0 && bid < 4096)
            for (int i = 0; i < 8; ++i) g_marks[kind + 2 * set][bid][i] = ts[i];
    }
#else
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int) {}
#endif
};

#ifdef SSF_PHASE_TIMING
#define SSF_CTX_KIND(k) , k
#else
#define SSF_CTX_KIND(k)
#endif
#define SSF_DEV_CTX(k)                                                        \
    extern __shared__ __attribute__((aligned(16))) char ssf_smem[];           \
    DevCtx ctx{{(int)threadIdx.x, (int)blockIdx.x, (int)blockDim.x, (int)gridDim.x, ssf_smem} SSF_CTX_KIND(k)}

// OCC = minimum waves per SIMD the register allocator must leave room for: 1 = up to 512
// registers per lane and no spills (one 256-thread workgroup per CU), 2 = 256 registers
// (two workgroups per CU).  LG = compile-time log2 of the transform length (0 = runtime).
// experiment (appendix #42): SSF_WPE = 3 squeezes the 16-value kernels into the 168 registers three waves per SIMD would need (the
// LDS tiles still allow two: this measures what the squeeze alone costs)
#ifndef SSF_WPE
#define SSF_WPE 0
#endif
#if SSF_WPE
#define SSF_WPE_ATTR __attribute__((amdgpu_waves_per_eu(SSF_WPE, SSF_WPE)))
#else
#define SSF_WPE_ATTR
#endif
template <typename T, int MAXT, int OCC, int LG> __global__ void __launch_bounds__(MAXT, OCC) SSF_WPE_ATTR k_row(const RowArgs<T> a) {
    SSF_DEV_CTX(0);
    row_body<T, LG>(ctx, unit_view(a, (int)blockIdx.y));
}
// eight values per thread: at most 128 registers, four waves per SIMD (two 512-thread workgroups, or one of 1024, per CU)
template <typename T, int MAXT, int LG> __global__ void __launch_bounds__(MAXT, 4) k_row8(const RowArgs<T> a) {
    SSF_DEV_CTX(0);
    row_body<T, LG, 8>(ctx, unit_view(a, (int)blockIdx.y));
}
// Manakov column kernels run 512 threads (x half | y half); the single-row modes run 256
// CI > 0: the workgroup's CI columns interleaved in LDS (fused_kernels.h: lds_put) -- the launch geometry must give exactly CI
// SG: the stage groups the instantiation carries (fused_kernels.h: stage_group)
template <typename T, int LG, int MODE, int CI = 0, int SG = SG_ALL>
__global__ void __launch_bounds__(MODE == CM_MK ? 512 : 256) k_col(const ColArgs<T> a) {
    SSF_DEV_CTX(1);
    col_body<T, LG, MODE, false, 16, CI, SG>(ctx, unit_view(a, (int)blockIdx.y));
}
// row lengths with factors 3 / 5 (mixed_fft.h): mixed-radix row stage, column stage with ragged last tiles
template <typename T, int MAXT> __global__ void __launch_bounds__(MAXT, MAXT <= 256 ? 2 : 1) k_row_mixed(const RowArgs<T> a) {
    SSF_DEV_CTX(0);
    row_mixed_body<T>(ctx, unit_view(a, (int)blockIdx.y), a.plan);
}
template <typename T, int LG, int MODE>
__global__ void __launch_bounds__(MODE == CM_MK ? 512 : 256) k_col_ragged(const ColArgs<T> a) {
    SSF_DEV_CTX(1);
    col_body<T, LG, MODE, true>(ctx, unit_view(a, (int)blockIdx.y));
}
// column lengths with factors 3 / 5 (fused_kernels.h: col_mixed_body): the tile in LDS, mixed-radix passes
template <typename T, int MODE> __global__ void __launch_bounds__(256, 2) k_col_mixed(const ColArgs<T> a) {
    SSF_DEV_CTX(1);
    col_mixed_body<T, MODE>(ctx, unit_view(a, (int)blockIdx.y), a.plan1);
}
// eight values per thread (at most 128 registers, four waves per SIMD): 512-thread workgroups, two per CU
template <typename T, int LG, int MODE> __global__ void __launch_bounds__(512, 4) k_col8(const ColArgs<T> a) {
    SSF_DEV_CTX(1);
    col_body<T, LG, MODE, false, 8>(ctx, unit_view(a, (int)blockIdx.y));
}
template <int LG> __global__ void __launch_bounds__(512, 4) k_col_pk8(const ColArgs<pf2> a) {
    SSF_DEV_CTX(1);
    col_pk_body<LG, 8>(ctx, unit_view(a, (int)blockIdx.y));
}
// complex64 Manakov: packed polarisation pairs (fused_kernels.h: col_pk_body); up to 512 threads (8 columns of 1024)
template <int LG, int CI = 0, int SG = SG_ALL> __global__ void __launch_bounds__(512) SSF_WPE_ATTR k_col_pk(const ColArgs<pf2> a) {
    SSF_DEV_CTX(1);
    col_pk_body<LG, 16, CI, SG>(ctx, unit_view(a, (int)blockIdx.y));
}
__global__ void __launch_bounds__(256) k_repack(const RepackArgs a) {
    SSF_DEV_CTX(1);
    repack_body(ctx, a);
}
template <typename T> __global__ void __launch_bounds__(256) k_amp(const AmpArgs<T> a) {
    SSF_DEV_CTX(1);
    amp_body<T>(ctx, a);
}

template <typename T, int MAXT> __global__ void __launch_bounds__(MAXT) k_ols(const OlsArgs<T> a) {
    SSF_DEV_CTX(1);
    ols_body<T, 0, 1>(ctx, a);
}

#if SSF_EXPERIMENTS
#include "fused_experiments.h"      // persistent span kernels (measured slower at every size: DESIGN.md appendix)
#endif

// ---- coupled batch across ranks (ssf_set_coupling_comm; reference optic/models/channels.py:394, 517-519) --------------------------------
// The column stage leaves per-workgroup partials (sums of lim_0 / lim_i, maxima of phi); the row stage reduces them in a fixed
// order.  When the pairs of ONE coupled call are spread over several ranks, the partials are first reduced per rank
// (k_couple_local: the row stage's own order, one workgroup), all-gathered (40 bytes per rank, on the plan's stream) and reduced
// over the ranks in rank order (k_couple_finish): every rank ends up with the same five doubles and the row stage reads those
// (npart = 1) -- same decisions, same step sizes, same iteration counts on every rank, no host in the loop.
struct CoupleArgs {
    const double *pnum0, *pden0, *pnum, *pden, *pmax;
    int npart;
    double *out;              // [5]: sum pnum0, sum pden0, sum pnum, sum pden, max pmax
};
__global__ void __launch_bounds__(256) k_couple_local(const CoupleArgs a) {
    __shared__ double sh[5][256];
    double s[5] = {0, 0, 0, 0, -INFINITY};
    for (int i = (int)threadIdx.x; i < a.npart; i += 256) {
        s[0] += a.pnum0[i];
        s[1] += a.pden0[i];
        s[2] += a.pnum[i];
        s[3] += a.pden[i];
        s[4] = a.pmax[i] > s[4] ? a.pmax[i] : s[4];
    }
    for (int q = 0; q < 5; ++q) sh[q][threadIdx.x] = s[q];
    __syncthreads();
    if (threadIdx.x < 5) {
        const int q = (int)threadIdx.x;
        double r = sh[q][0];
        for (int i = 1; i < 256; ++i) r = q == 4 ? (sh[q][i] > r ? sh[q][i] : r) : r + sh[q][i];
        a.out[q] = r;
    }
}
__global__ void k_couple_finish(const double *gathered, int nranks, double *out) {
    const int q = (int)threadIdx.x;
    if (q >= 5) return;
    double r = gathered[q];
    for (int i = 1; i < nranks; ++i) {
        const double x = gathered[5 * i + q];
        r = q == 4 ? (x > r ? x : r) : r + x;
    }
    out[q] = r;
}

template <typename T> using RowFn = void (*)(const RowArgs<T>);
template <typename T> using ColFn = void (*)(const ColArgs<T>);

// kernel selection: specialised lengths for the sizes that matter, generic otherwise
template <typename T> RowFn<T> pick_row(int lg2, int block, int occ, int vpt = 16) {
    if (vpt == 8) {
        if (block <= 512) {
            switch (lg2) {
            case 6: return k_row8<T, 512, 6>;          // (the short rows of fields that do not fill the chip: 2^12 ... 2^17 samples;
            case 7: return k_row8<T, 512, 7>;          //  compile-time lengths are worth 25 - 30 % of a launch there)
            case 8: return k_row8<T, 512, 8>;
            case 9: return k_row8<T, 512, 9>;
            case 10: return k_row8<T, 512, 10>;
            case 11: return k_row8<T, 512, 11>;
            case 12: return k_row8<T, 512, 12>;
            default: return k_row8<T, 512, 0>;
            }
        }
        return lg2 == 13 ? k_row8<T, 1024, 13> : k_row8<T, 1024, 0>;
    }
    if (block <= 256) {
        if (occ == 2) {
            switch (lg2) {
            case 10: return k_row<T, 256, 2, 10>;
            case 11: return k_row<T, 256, 2, 11>;
            case 12: return k_row<T, 256, 2, 12>;
            default: return k_row<T, 256, 2, 0>;
            }
        }
        switch (lg2) {
        case 10: return k_row<T, 256, 1, 10>;
        case 11: return k_row<T, 256, 1, 11>;
        case 12: return k_row<T, 256, 1, 12>;
        default: return k_row<T, 256, 1, 0>;
        }
    }
    if (block <= 512) return lg2 == 13 ? k_row<T, 512, 1, 13> : k_row<T, 512, 1, 0>;
    return lg2 == 14 ? k_row<T, 1024, 1, 14> : k_row<T, 1024, 1, 0>;
}
template <typename T, int LG> ColFn<T> pick_col_mode(int mode) {
    switch (mode) {
    case CM_NLSE_FIRST: return k_col<T, LG, CM_NLSE_FIRST>;
    case CM_NLSE_STEP: return k_col<T, LG, CM_NLSE_STEP>;
    case CM_NLSE_LAST: return k_col<T, LG, CM_NLSE_LAST>;
    case CM_MK: return k_col<T, LG, CM_MK>;
    case CM_PLAIN_FWD: return k_col<T, LG, CM_PLAIN_FWD>;
    default: return k_col<T, LG, CM_PLAIN_INV>;
    }
}
template <typename T> ColFn<T> pick_col_mixed(int mode) {
    switch (mode) {
    case CM_NLSE_FIRST: return k_col_mixed<T, CM_NLSE_FIRST>;
    case CM_NLSE_STEP: return k_col_mixed<T, CM_NLSE_STEP>;
    case CM_NLSE_LAST: return k_col_mixed<T, CM_NLSE_LAST>;
    case CM_MK: return k_col_mixed<T, CM_MK>;
    case CM_PLAIN_FWD: return k_col_mixed<T, CM_PLAIN_FWD>;
    default: return k_col_mixed<T, CM_PLAIN_INV>;
    }
}
template <typename T> ColFn<T> pick_col_ragged(int lg1, int mode) {
    if (mode == CM_MK) {                      // the Manakov stage is worth its specialised lengths
        switch (lg1) {
        case 7: return k_col_ragged<T, 7, CM_MK>;
        case 8: return k_col_ragged<T, 8, CM_MK>;
        case 9: return k_col_ragged<T, 9, CM_MK>;
        default: return k_col_ragged<T, 0, CM_MK>;
        }
    }
    switch (mode) {
    case CM_NLSE_FIRST: return k_col_ragged<T, 0, CM_NLSE_FIRST>;
    case CM_NLSE_STEP: return k_col_ragged<T, 0, CM_NLSE_STEP>;
    case CM_NLSE_LAST: return k_col_ragged<T, 0, CM_NLSE_LAST>;
    case CM_PLAIN_FWD: return k_col_ragged<T, 0, CM_PLAIN_FWD>;
    default: return k_col_ragged<T, 0, CM_PLAIN_INV>;
    }
}
template <typename T, int LG> ColFn<T> pick_col8_mode(int mode) {
    switch (mode) {
    case CM_NLSE_FIRST: return k_col8<T, LG, CM_NLSE_FIRST>;
    case CM_NLSE_STEP: return k_col8<T, LG, CM_NLSE_STEP>;
    case CM_NLSE_LAST: return k_col8<T, LG, CM_NLSE_LAST>;
    case CM_MK: return k_col8<T, LG, CM_MK>;
    case CM_PLAIN_FWD: return k_col8<T, LG, CM_PLAIN_FWD>;
    default: return k_col8<T, LG, CM_PLAIN_INV>;
    }
}
template <typename T> ColFn<T> pick_col8(int lg1, int mode) {
    switch (lg1) {
    case 6: return pick_col8_mode<T, 6>(mode);
    case 8: return pick_col8_mode<T, 8>(mode);
    default: return pick_col8_mode<T, 0>(mode);
    }
}
// Manakov column stage with the columns of a workgroup interleaved in LDS: the geometries the chip-filling fields get
// (FusedCore::col_geometry: 8 columns of 256 / 512 per polarisation row in double precision)
#if SSF_EXPERIMENTS
template <typename T> ColFn<T> pick_col_il(int lg1, int cols) {
    if constexpr (std::is_same<T, double>::value) {
        if (cols == 8 && lg1 == 8) return k_col<T, 8, CM_MK, 8>;
        if (cols == 8 && lg1 == 9) return k_col<T, 9, CM_MK, 8>;
    }
    return nullptr;
}
#else
template <typename T> ColFn<T> pick_col_il(int, int) { return nullptr; }
#endif
// stage-specialised Manakov column kernels (fused_kernels.h: stage_group; FusedCore::run_span enqueues them along the predicted
// stage sequence) for the geometries the chip-filling fields get (FusedCore::col_geometry): complex128 columns of 256 / 512 with 8
// per workgroup and polarisation row (2^20 ... 2^22 samples), packed complex64 columns of 512 / 1024 (2^21 ... 2^23)
template <typename T, int LG> ColFn<T> pick_col_sg_d(int sg) {
    switch (sg) {
    case SG_H | SG_RARE: return k_col<T, LG, CM_MK, 0, SG_H | SG_RARE>;
    case SG_ADV: return k_col<T, LG, CM_MK, 0, SG_ADV>;
    case SG_FIN: return k_col<T, LG, CM_MK, 0, SG_FIN>;
    default: return nullptr;
    }
}
template <int LG> ColFn<pf2> pick_col_sg_pk(int sg) {
    switch (sg) {
    case SG_H | SG_RARE: return k_col_pk<LG, 0, SG_H | SG_RARE>;
    case SG_ADV: return k_col_pk<LG, 0, SG_ADV>;
    case SG_FIN: return k_col_pk<LG, 0, SG_FIN>;
    default: return nullptr;
    }
}
template <typename T> ColFn<T> pick_col_sg(int lg1, int cols, int sg) {
    if constexpr (std::is_same<T, double>::value) {
        if (lg1 == 8 && cols == 8) return pick_col_sg_d<T, 8>(sg);
        if (lg1 == 9 && cols == 8) return pick_col_sg_d<T, 9>(sg);
    } else if constexpr (std::is_same<T, pf2>::value) {
        if (lg1 == 10 && cols == 4) return pick_col_sg_pk<10>(sg);
        if (lg1 == 9 && cols == 8) return pick_col_sg_pk<9>(sg);
    }
    return nullptr;
}
template <typename T> ColFn<T> pick_col(int lg1, int mode) {
    switch (lg1) {
    case 7: return pick_col_mode<T, 7>(mode);
    case 8: return pick_col_mode<T, 8>(mode);
    case 9: return pick_col_mode<T, 9>(mode);
    default: return pick_col_mode<T, 0>(mode);
    }
}

struct HipBackend {
    ssf_plan *pl;
    hipError_t first_err = hipSuccess;
    std::string where;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    int row_occ = 2;
    bool col_il = false;     // interleaved LDS columns in the Manakov column stage (experiment builds: SSF_COL_IL=1; measured: no gain)
    // optional per-launch event timing (ssf_set_profiling)
    bool profiling = false;
    struct Stamp { hipEvent_t a, b; int cat; };
    std::vector<Stamp> stamps;
    std::vector<hipEvent_t> pool;
    ssf_kernel_times kt{};
    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        chk(hipEventCreate(&e), "hipEventCreate");
        return e;
    }
    void stamp_begin(int cat) {
        if (!profiling) return;
        Stamp s{get_event(), get_event(), cat};
        chk(hipEventRecord(s.a, pl->stream), "hipEventRecord");
        stamps.push_back(s);
    }
    void stamp_end() {
        if (!profiling) return;
        chk(hipEventRecord(stamps.back().b, pl->stream), "hipEventRecord");
    }
    void collect() {        // call after a stream synchronise
        // A launch whose events are more than kOutlier x the median of its class apart did not run that long: the stream was held
        // up (a clock transition, another process' work, the profiler itself) and the events saw it -- 48 ms in ONE of 1 600
        // launches turned an 18 us average into 47 us in a round-5 run.  Such launches are counted in `outliers` and left out.
        constexpr float kOutlier = 8.0f;
        std::vector<float> ms_of(stamps.size(), -1.0f);
        std::vector<float> by_cat[8];
        for (size_t i = 0; i < stamps.size(); ++i) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, stamps[i].a, stamps[i].b) == hipSuccess) {
                ms_of[i] = ms;
                by_cat[stamps[i].cat & 7].push_back(ms);
            }
            pool.push_back(stamps[i].a);
            pool.push_back(stamps[i].b);
        }
        float med[8];
        for (int c = 0; c < 8; ++c) {
            med[c] = 0;
            if (!by_cat[c].empty()) {
                std::nth_element(by_cat[c].begin(), by_cat[c].begin() + by_cat[c].size() / 2, by_cat[c].end());
                med[c] = by_cat[c][by_cat[c].size() / 2];
            }
        }
        for (size_t i = 0; i < stamps.size(); ++i) {
            const float ms = ms_of[i];
            if (ms < 0) continue;
            const int cat = stamps[i].cat;
            if (by_cat[cat & 7].size() >= 8 && ms > kOutlier * med[cat & 7]) {
                ++kt.outliers;
                continue;
            }
            // categories: 0 row, 1 Manakov column stage (general kernel), 3 other, 4 / 5 / 6 the H / ADV / FIN kernels
            const bool col = cat == 1 || cat >= 4;
            double *t = cat == 0 ? &kt.row_ms : col ? &kt.col_ms : &kt.other_ms;
            int64_t *n = cat == 0 ? &kt.row_n : col ? &kt.col_n : &kt.other_n;
            *t += ms;
            *n += 1;
            if (cat >= 4) {
                double *ts = cat == 4 ? &kt.col_h_ms : cat == 5 ? &kt.col_adv_ms : &kt.col_fin_ms;
                int64_t *ns = cat == 4 ? &kt.col_h_n : cat == 5 ? &kt.col_adv_n : &kt.col_fin_n;
                *ts += ms;
                *ns += 1;
            }
        }
        stamps.clear();
    }
    explicit HipBackend(ssf_plan *p) : pl(p) {
        if (const char *s = tune_env("SSF_FUSED_ROW_OCC")) row_occ = atoi(s) == 1 ? 1 : 2;
        if (const char *s = tune_env("SSF_COL_IL")) col_il = atoi(s) != 0;
        chk(hipEventCreate(&ev0), "hipEventCreate");
        chk(hipEventCreate(&ev1), "hipEventCreate");
    }
    ~HipBackend() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        for (auto &s : stamps) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
    void chk(hipError_t e, const char *what) {
        if (e != hipSuccess && first_err == hipSuccess) {
            first_err = e;
            where = what;
        }
    }
    bool ok() const { return first_err == hipSuccess; }
    std::string last_error() const {
        return first_err == hipSuccess ? std::string() : where + ": " + hipGetErrorString(first_err);
    }
    void *alloc(size_t n) {
        void *p = nullptr;
        chk(hipMalloc(&p, n ? n : 16), "hipMalloc");
        return p;
    }
    void free(void *p) { (void)hipFree(p); }
    void h2d(void *d, const void *h, size_t n) {
        chk(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, pl->stream), "hipMemcpyAsync H2D");
        chk(hipStreamSynchronize(pl->stream), "hipStreamSynchronize");
    }
    void d2h(void *h, const void *d, size_t n) {
        chk(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, pl->stream), "hipMemcpyAsync D2H");
        chk(hipStreamSynchronize(pl->stream), "hipStreamSynchronize");
    }
    void h2d_big(void *d, const void *h, size_t n) { chk(pl->stager.h2d(d, h, n, pl->stream), "staged H2D"); }
    void d2h_big(void *h, const void *d, size_t n) { chk(pl->stager.d2h(h, d, n, pl->stream), "staged D2H"); }
    template <typename C> void aos_to_soa(C *soa, const C *aos, long long N, int nrows) {
        k_aos_to_soa<C><<<1024, 256, 0, pl->stream>>>(aos, soa, N, nrows);
        chk(hipStreamSynchronize(pl->stream), "aos_to_soa");
    }
    template <typename C> void soa_to_aos(C *aos, const C *soa, long long N, int nrows) {
        k_soa_to_aos<C><<<1024, 256, 0, pl->stream>>>(soa, aos, N, nrows);
        chk(hipGetLastError(), "soa_to_aos");
    }
    void d2d(void *d, const void *s, size_t n) {
        chk(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, pl->stream), "hipMemcpyAsync D2D");
    }
    void memset(void *d, int v, size_t n) { chk(hipMemsetAsync(d, v, n, pl->stream), "hipMemsetAsync"); }
    void sync() { chk(hipStreamSynchronize(pl->stream), "hipStreamSynchronize"); }
    void time_begin() { chk(hipEventRecord(ev0, pl->stream), "hipEventRecord"); }
    double time_end() {
        chk(hipEventRecord(ev1, pl->stream), "hipEventRecord");
        chk(hipStreamSynchronize(pl->stream), "hipStreamSynchronize");
        float ms = 0;
        chk(hipEventElapsedTime(&ms, ev0, ev1), "hipEventElapsedTime");
        collect();
        return ms;
    }
    size_t row_lds_max = 0, col_lds_max = 0;
    void prepare(size_t row_lds, size_t col_lds) {
        row_lds_max = row_lds;
        col_lds_max = col_lds;
    }
    template <typename T> void launch_row(const RowArgs<T> &a, int grid, int block, size_t lds, int units = 1) {
        RowFn<T> f;
        if constexpr (std::is_same<T, pf2>::value) f = pick_row<T>(a.log2N2, block, row_occ, a.vpt);     // (no mixed-radix rows there)
        else
            f = !a.mixed ? pick_row<T>(a.log2N2, block, row_occ, a.vpt)
                : block <= 256 ? (RowFn<T>)k_row_mixed<T, 256> : block <= 512 ? (RowFn<T>)k_row_mixed<T, 512> : (RowFn<T>)k_row_mixed<T, 1024>;
        arm((const void *)f);
        stamp_begin(0);
        f<<<dim3((unsigned)grid, (unsigned)units), block, lds, pl->stream>>>(a);
        stamp_end();
        chk(hipGetLastError(), "launch k_row");
    }
    template <typename T> void launch_col(const ColArgs<T> &a, int grid, int block, size_t lds, int units = 1) {
        ColFn<T> f;
        const int cols = a.N1mix ? a.mix_cols : (block / a.npol) / ((1 << a.log2N1) / (a.vpt == 8 ? 8 : 16));      // columns per workgroup and polarisation row
        if constexpr (std::is_same<T, pf2>::value) {
            if (a.vpt == 8) {
                switch (a.log2N1) {
                case 6: f = k_col_pk8<6>; break;
                case 8: f = k_col_pk8<8>; break;
                case 10: f = k_col_pk8<10>; break;
                default: f = k_col_pk8<0>; break;
                }
            } else
            switch (a.log2N1) {
            case 7: f = k_col_pk<7>; break;
            case 8: f = k_col_pk<8>; break;
#if SSF_EXPERIMENTS
            case 9: f = cols == 8 && col_il ? k_col_pk<9, 8> : k_col_pk<9>; break;
            case 10: f = cols == 4 && col_il ? k_col_pk<10, 4> : k_col_pk<10>; break;
#else
            case 9: f = k_col_pk<9>; break;
            case 10: f = k_col_pk<10>; break;
#endif
            default: f = k_col_pk<0>; break;
            }
        } else {
            f = a.N1mix ? pick_col_mixed<T>(a.mode)
                : a.N2  ? pick_col_ragged<T>(a.log2N1, a.mode) : a.vpt == 8 ? pick_col8<T>(a.log2N1, a.mode) : pick_col<T>(a.log2N1, a.mode);
            if (!a.N2 && a.vpt != 8 && a.mode == CM_MK && col_il)
                if (ColFn<T> fi = pick_col_il<T>(a.log2N1, cols)) f = fi;
        }
        int cat = a.mode == CM_MK ? 1 : 3;
        if (a.mode == CM_MK && a.sg && a.sg != SG_ALL && !a.N2 && a.vpt != 8)
            if (ColFn<T> fs = pick_col_sg<T>(a.log2N1, cols, a.sg)) {
                f = fs;
                cat = (a.sg & SG_H) ? 4 : a.sg == SG_ADV ? 5 : 6;
            }
        arm((const void *)f);
        stamp_begin(cat);
        f<<<dim3((unsigned)grid, (unsigned)units), block, lds, pl->stream>>>(a);
        stamp_end();
        chk(hipGetLastError(), "launch k_col");
    }
    // are there stage-specialised column kernels for this geometry?  (run_span must not enqueue a pattern the launcher would
    // silently serve with the general kernel: that one advances the state at EVERY launch, which is fine, but the pattern's
    // chunk sizes assume idle launches)
    template <typename T> bool can_split_cols(const ColArgs<T> &a, int block) const {
        if (a.N1mix) return false;
        const int cols = (block / a.npol) / ((1 << a.log2N1) / (a.vpt == 8 ? 8 : 16));
        return !a.N2 && a.vpt != 8 && pick_col_sg<T>(a.log2N1, cols, SG_FIN) != nullptr;
    }
    // raise the dynamic-LDS cap of a kernel the first time THIS backend (= this plan, hence this
    // device) launches it; the attribute is per device, so the record must not be shared
    std::vector<const void *> armed;
    void arm(const void *f) {
        for (const void *g : armed)
            if (g == f) return;
        chk(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        armed.push_back(f);
    }
#if SSF_EXPERIMENTS
    // persistent span kernel: grid <= CUs (every workgroup resident), 256 threads, any supported power-of-two split.
    // OFF by default: measured slower than one launch per stage at every size (MI355X, ssfm, steps/s, launches vs
    // persistent: 2^12 41.1k / 31.6k, 2^14 43.6k / 28.7k, 2^16 51.2k / 49.0k, 2^18 49.6k / 25.6k, 2^20 33.1k / 7.6k;
    // gpurun_out/r2e) -- an agent-scope release + acquire around a grid barrier costs more than the 1.5-2 us of a kernel
    // boundary (the micro-architecture guide's price list says the same: 4-7 us per barrier), and a stage's ~10 us is one
    // wave's ~2000 dependent FP64 instructions, not launch overhead.  SSF_PERSIST=<largest grid> turns it on.
    static constexpr bool kCanPersist = true;
    int persist_limit() {
        if (const char *e = tune_env("SSF_PERSIST")) return atoi(e);
        return 0;
    }
    template <typename T> int launch_nlse_span(const SpanNlseArgs<T> &a, int grid, size_t lds) {
        void (*f)(const SpanNlseArgs<T>);
        if constexpr (std::is_same<T, pf2>::value) return SSF_ERR_UNSUPPORTED;
        else {
            if (a.row.log2N2 == 8 && a.col.log2N1 == 8) f = k_nlse_span<T, 8, 8>;       // (config 1's 256 x 256 split)
            else f = k_nlse_span<T, 0, 0>;
            arm((const void *)f);
            chk(hipMemsetAsync(a.bar, 0, 3 * sizeof(unsigned), pl->stream), "hipMemsetAsync(barrier)");
            stamp_begin(3);
            f<<<grid, 256, lds, pl->stream>>>(a);
            stamp_end();
            chk(hipGetLastError(), "launch k_nlse_span");
            return ok() ? SSF_OK : SSF_ERR_HIP;
        }
    }
    // persistent Manakov span kernel (experiment, off by default): workers = SSF_PERSIST_MK, SSF_PERSIST_XCD=1: one XCD only
    int persist_mk_workers() {
        if (const char *e = tune_env("SSF_PERSIST_MK")) return atoi(e);
        return 0;
    }
    bool persist_mk_xcd() {
        const char *e = tune_env("SSF_PERSIST_XCD");
        return e && atoi(e) != 0;
    }
    template <typename T> int launch_mk_span(const SpanMkArgs<T> &a, int grid, size_t lds) {
        if constexpr (!std::is_same<T, double>::value) return SSF_ERR_UNSUPPORTED;     // (an experiment: double precision only)
        else {
            void (*f)(const SpanMkArgs<T>) = k_mk_span<T, 0, 0>;       // (length-specialised bodies spill 1.8 KiB per lane here)
            arm((const void *)f);
            chk(hipMemsetAsync(a.bar, 0, 8 * sizeof(unsigned), pl->stream), "hipMemsetAsync(barrier)");
            stamp_begin(3);
            f<<<grid, 256, lds, pl->stream>>>(a);
            stamp_end();
            chk(hipGetLastError(), "launch k_mk_span");
            return ok() ? SSF_OK : SSF_ERR_HIP;
        }
    }
#else
    static constexpr bool kCanPersist = false;
#endif
    // partial sums / maxima of this rank -> the same five values on every rank of `comm`, on the plan's stream (see k_couple_local)
    static constexpr bool kCanCouple = true;
    int couple(void *comm, const double *part, size_t stride, int npart, double *work) {
        ssf_comm *c = (ssf_comm *)comm;
        const int nr = comm_nranks(c);
        CoupleArgs a{part + 3 * stride, part + 4 * stride, part + stride, part + 2 * stride, part, npart, work};   // (col_args' layout)
        k_couple_local<<<1, 256, 0, pl->stream>>>(a);
        chk(hipGetLastError(), "launch k_couple_local");
        int rc = comm_allgather_on(c, work, work + 8, 5 * sizeof(double), pl->stream);
        if (rc) {
            if (first_err == hipSuccess) {
                first_err = hipErrorUnknown;
                where = std::string("coupled batch: ") + comm_error(c);
            }
            return rc;
        }
        k_couple_finish<<<1, 64, 0, pl->stream>>>(work + 8, nr, work);
        chk(hipGetLastError(), "launch k_couple_finish");
        return SSF_OK;
    }
    bool sink_active() const { return pl->sink.active(); }
    template <typename C> void sink_capture(const C *soa, long long N, int nrows) {
        chk(pl->sink.capture(soa, N, nrows, pl->stream), "snapshot sink");
    }
    void launch_repack(const RepackArgs &a, int grid, int block) {
        k_repack<<<grid, block, 0, pl->stream>>>(a);
        chk(hipGetLastError(), "launch k_repack");
    }
    template <typename T> void launch_amp(const AmpArgs<T> &a, int grid, int block) {
        k_amp<T><<<grid, block, 0, pl->stream>>>(a);
        chk(hipGetLastError(), "launch k_amp");
    }
};

template <typename T> class FusedEngine final : public Engine {
    ssf_plan *pl;
    HipBackend be;
    FusedCore<T, HipBackend> core;

  public:
    explicit FusedEngine(ssf_plan *p) : pl(p), be(p), core(be, p->N, p->nrows, p->precision, nullptr, p->units) {}
    int id() const override { return SSF_ENGINE_FUSED; }
    int ret(int rc) {
        if (rc != SSF_OK) pl->err = core.err.empty() ? be.last_error() : core.err;
        return rc;
    }
    int init() { return ret(core.init()); }
    int upload(const void *field, bool aos) override {
        be.kt = ssf_kernel_times{};
        return ret(core.upload(field, aos));
    }
    int download(void *field, int which, bool aos) override { return ret(core.download(field, which, aos)); }
    int n_snapshots() const override { return (int)core.snaps.size(); }
    int execute(const ssf_params &p, int s0, int s1, const void *noise, ssf_stats *st, ssf_trace *tr) override {
        return ret(core.execute(p, s0, s1, noise, st, tr));
    }
    int linear_channel(double Fs, double Fc, double alpha, double D, double L) override {
        return ret(core.linear_channel(Fs, Fc, alpha, D, L));
    }
    int set_profiling(int on) override {
        be.profiling = on != 0;
        return SSF_OK;
    }
    int kernel_times(ssf_kernel_times *out) override {
        *out = be.kt;
        return SSF_OK;
    }
    void reset_times() { be.kt = ssf_kernel_times{}; }
    int set_lanes(int n) override {
        core.lanes_hint = n;
        if (core.pk) core.pk->lanes_hint = n;
        return SSF_OK;
    }
    int unit_stats(int u, ssf_stats *out) override { return core.unit_stats(u, out) ? SSF_OK : SSF_ERR_BAD_ARG; }
    int set_coupling_comm(ssf_comm *comm) override {
        if (comm && (core.N2mix || core.units > 1)) return SSF_ERR_UNSUPPORTED;
        int rc = core.set_couple(comm, comm ? comm_nranks(comm) : 0);
        if (core.pk && rc == SSF_OK) rc = core.pk->set_couple(comm, comm ? comm_nranks(comm) : 0);
        return rc;
    }
};

template <typename T> class FusedConvImpl final : public FusedConv {
    using Cc = cx<T>;
    ssf_plan *pl;
    HipBackend be;
    FusedCore<T, HipBackend> core;
    Cc *hk[2] = {nullptr, nullptr};
    int64_t M;
    std::string err_;

  public:
    FusedConvImpl(ssf_plan *p, int64_t M_, int nrows) : pl(p), be(p), core(be, M_, nrows, p->precision), M(M_) {}
    ~FusedConvImpl() override {
        for (Cc *h : hk)
            if (h) (void)hipFree(h);
    }
    int init() {
        if (core.N2mix) {
            err_ = "convolution length must be a power of two";
            return SSF_ERR_BAD_ARG;
        }
        int rc = core.init();
        if (rc) err_ = core.err;
        return rc;
    }
    std::string error() const override { return err_.empty() ? be.last_error() : err_; }
    void *work() override { return core.T0; }
    int set_kernel(int which, const void *b_host) override {
        if (which < 0 || which > 1) return SSF_ERR_BAD_ARG;
        if (!hk[which] && hipMalloc(&hk[which], sizeof(Cc) * (size_t)M) != hipSuccess) {
            err_ = "out of memory (convolution kernel)";
            return SSF_ERR_OOM;
        }
        be.memset(core.T0, 0, core.field_bytes);                       // the kernel goes into row 0, the other rows are idle
        be.h2d(core.T0, b_host, sizeof(Cc) * (size_t)M);
        core.launch_col_plain(CM_PLAIN_FWD, core.T0, (T)0);
        core.launch_row_conv(nullptr, 1);                              // spectrum in the row kernel's own order -> G
        be.d2d(hk[which], core.G, sizeof(Cc) * (size_t)M);
        be.sync();
        return be.ok() ? SSF_OK : SSF_ERR_HIP;
    }
    int run(int which) override {
        core.launch_col_plain(CM_PLAIN_FWD, core.T0, (T)0);
        core.launch_row_conv(hk[which], 0);
        core.launch_col_plain(CM_PLAIN_INV, core.T0, (T)0);
        return be.ok() ? SSF_OK : SSF_ERR_HIP;
    }
};

template <typename T> class FusedRowsImpl final : public FusedRows {
    ssf_plan *pl;
    HipBackend be;
    int64_t N;
    int nrows, tpr = 128, rows_wg = 1;
    MixPlan plan{};
    cx<double> *wtab = nullptr;
    LinOp *lin_d = nullptr;
    double key[5] = {0, 0, 0, 0, 0};             // parameters of the operator that is on the device
    std::string err_;

  public:
    FusedRowsImpl(ssf_plan *p, int64_t N_, int nrows_) : pl(p), be(p), N(N_), nrows(nrows_) {}
    ~FusedRowsImpl() override {
        if (wtab) (void)hipFree(wtab);
        if (lin_d) (void)hipFree(lin_d);
    }
    static bool supports(int64_t n) {
        if (n < 16 || n > kMixMaxRow) return false;
        int64_t r = n;
        for (int q : {2, 3, 5})
            while (r % q == 0) r /= q;
        MixPlan mp;
        return r == 1 && mix_make_plan((int)n, &mp);
    }
    bool use_wtab = false;
    int init() {
        // These launches are latency chains of one workgroup per row on an otherwise idle GPU (a field has 2 K rows): as many
        // threads per row as a pass has butterflies (one round per pass), one row per workgroup, and the pass twiddles from
        // sincospi instead of a table in global memory (a dependent L2 round trip per pass costs more than 60 instructions
        // here; the big mixed-radix rows are throughput-bound and keep the table).  SSF_ROWS_TPR / SSF_ROWS_WTAB: A/B knobs.
        tpr = N <= 2048 ? 256 : 512;
        if (const char *e = tune_env("SSF_ROWS_TPR")) tpr = std::max(64, std::min(1024, atoi(e)));
        while (16 * tpr < N) tpr *= 2;
        rows_wg = nrows > 512 ? std::max(1, 256 / tpr) : 1;
        while (nrows % rows_wg) rows_wg >>= 1;
        if (const char *e = tune_env("SSF_ROWS_WTAB")) use_wtab = atoi(e) != 0;
        if (!mix_make_plan((int)N, &plan, tpr)) return SSF_ERR_UNSUPPORTED;
        std::vector<cx<double>> w((size_t)N);
        for (int64_t q = 0; q < N; ++q) {
            const double a = -2.0 * 3.14159265358979323846 * (double)q / (double)N;
            w[(size_t)q].re = std::cos(a);
            w[(size_t)q].im = std::sin(a);
        }
        if (hipMalloc(&wtab, sizeof(cx<double>) * (size_t)N) != hipSuccess || hipMalloc(&lin_d, sizeof(LinOp)) != hipSuccess) {
            err_ = "out of memory (row transform tables)";
            return SSF_ERR_OOM;
        }
        be.h2d(wtab, w.data(), sizeof(cx<double>) * (size_t)N);
        return be.ok() ? SSF_OK : SSF_ERR_HIP;
    }
    std::string error() const override { return err_.empty() ? be.last_error() : err_; }
    int lin(const void *in, void *out, double hzh, double lin_a, double lin_b, double w_scale, double scale) override {
        const double k5[5] = {hzh, lin_a, lin_b, w_scale, scale};
        if (std::memcmp(k5, key, sizeof(key)) != 0) {            // a new step size / span: the operator block changes
            const double w2 = (w_scale / (double)N) * (w_scale / (double)N);
            LinOp lo = make_linop(hzh, lin_a, lin_b, w2, scale, 4);
            be.h2d(lin_d, &lo, sizeof(LinOp));
            std::memcpy(key, k5, sizeof(key));
        }
        RowArgs<T> a{};
        a.G = (cx<T> *)out;
        a.src = in == out ? nullptr : (const cx<T> *)in;
        a.log2N1 = 0;
        a.log2N2 = 0;
        a.nfft = nrows;
        a.use_ctrl = 0;
        a.lin = lin_d;
        a.N2 = (int)N;
        a.N = N;
        a.mixed = 1;
        a.plan = plan;
        a.wtab = use_wtab ? wtab : nullptr;
        a.rows_per_wg = rows_wg;
        be.launch_row(a, nrows / rows_wg, tpr * rows_wg, 4096 + (size_t)rows_wg * (size_t)N * sizeof(cx<T>));
        return be.ok() ? SSF_OK : SSF_ERR_HIP;
    }
};

template <typename T>
int overlap_save_t(int64_t sigLen, int nrows, int lg, int K, const void *Hfft, const void *in, void *out, std::string *err) {
    using Cc = cx<T>;
    const int nfft = 1 << lg, d = nfft - K + 1, discard = K - 1, D = (K - 1) / 2;
    const long long numBlocks = (sigLen + K - 1 + d - 1) / d;                       // core.py:1023-1025
    const size_t sig_bytes = sizeof(Cc) * (size_t)sigLen * (size_t)nrows;
    hipStream_t st = nullptr;
    Cc *din = nullptr, *dout = nullptr, *dH = nullptr;
    auto fail_ = [&](const char *what, hipError_t e) {
        *err = std::string(what) + ": " + hipGetErrorString(e);
        if (din) (void)hipFree(din);
        if (dout) (void)hipFree(dout);
        if (dH) (void)hipFree(dH);
        if (st) (void)hipStreamDestroy(st);
        return e == hipErrorOutOfMemory ? SSF_ERR_OOM : SSF_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return fail_("hipStreamCreate", e);
    if ((e = hipMalloc(&din, sig_bytes)) != hipSuccess) return fail_("hipMalloc", e);
    if ((e = hipMalloc(&dout, sig_bytes)) != hipSuccess) return fail_("hipMalloc", e);
    if ((e = hipMalloc(&dH, sizeof(Cc) * (size_t)nfft)) != hipSuccess) return fail_("hipMalloc", e);
    std::vector<Cc> Hs((size_t)nfft);                                                // fold the 1/NFFT of the ifft into H
    for (int i = 0; i < nfft; ++i) {
        Hs[(size_t)i].re = ((const Cc *)Hfft)[i].re / (T)nfft;
        Hs[(size_t)i].im = ((const Cc *)Hfft)[i].im / (T)nfft;
    }
    ols_permute_filter(Hs.data(), lg);                                               // (the order ols_body_x reads)
    Stager stg;
    (void)stg.init();
    if ((e = stg.h2d(din, in, sig_bytes, st)) != hipSuccess) return fail_("upload", e);
    if ((e = hipMemcpyAsync(dH, Hs.data(), sizeof(Cc) * (size_t)nfft, hipMemcpyHostToDevice, st)) != hipSuccess) return fail_("upload H", e);
    OlsArgs<T> a{};
    a.in = din;
    a.out = dout;
    a.H = dH;
    a.sigLen = sigLen;
    a.njobs = numBlocks * nrows;
    a.nrows = nrows;
    a.log2nfft = lg;
    a.d = d;
    a.discard = discard;
    a.D = D;
    ols_defaults(a);
    const int tpf = nfft / 16;
    const int block = tpf >= 256 ? tpf : 256, fpw = block / tpf;
    const long long grid = (a.njobs + fpw - 1) / fpw;
    const size_t lds = (size_t)fpw * lds_slots_per_fft(nfft) * sizeof(Cc);
    if (block <= 256) {
        (void)hipFuncSetAttribute((const void *)k_ols<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        k_ols<T, 256><<<(unsigned)grid, block, lds, st>>>(a);
    } else {
        (void)hipFuncSetAttribute((const void *)k_ols<T, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        k_ols<T, 1024><<<(unsigned)grid, block, lds, st>>>(a);
    }
    if ((e = hipGetLastError()) != hipSuccess) return fail_("launch k_ols", e);
    if ((e = stg.d2h(out, dout, sig_bytes, st)) != hipSuccess) return fail_("download", e);
    (void)hipFree(din);
    (void)hipFree(dout);
    (void)hipFree(dH);
    (void)hipStreamDestroy(st);
    return SSF_OK;
}

// The device-side reduction of a coupled batch by itself (ssf_couple_reduce_selftest): `parts` holds, per rank, the five arrays of
// per-workgroup partials in col_args' order (pmax, pnum, pden, pnum0, pden0), npart values each -- what the column stage of that
// rank would have left.  Every rank's block goes through k_couple_local into the slot the all-gather would put it in, then
// k_couple_finish reduces the slots in rank order: out5 = (sum pnum0, sum pden0, sum pnum, sum pden, max pmax), the five values
// the row stage of EVERY rank reads.  No communicator involved: this is the arithmetic that must give identical bits on all ranks.
inline int couple_reduce_selftest_impl(int nranks, int npart, const double *parts, double *out5, std::string *err) {
    if (nranks < 1 || npart < 1 || !parts || !out5) return SSF_ERR_BAD_ARG;
    double *d = nullptr, *work = nullptr;
    const size_t n = (size_t)nranks * 5 * (size_t)npart;
    hipError_t e = hipMalloc(&d, sizeof(double) * n);
    if (e == hipSuccess) e = hipMalloc(&work, sizeof(double) * (size_t)(8 + 5 * nranks));
    if (e == hipSuccess) e = hipMemcpy(d, parts, sizeof(double) * n, hipMemcpyHostToDevice);
    for (int r = 0; r < nranks && e == hipSuccess; ++r) {
        const double *b = d + (size_t)r * 5 * npart;
        CoupleArgs a{b + 3 * (size_t)npart, b + 4 * (size_t)npart, b + (size_t)npart, b + 2 * (size_t)npart, b, npart, work + 8 + 5 * r};
        k_couple_local<<<1, 256>>>(a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        k_couple_finish<<<1, 64>>>(work + 8, nranks, work);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out5, work, sizeof(double) * 5, hipMemcpyDeviceToHost);
    if (d) (void)hipFree(d);
    if (work) (void)hipFree(work);
    if (e != hipSuccess) {
        *err = std::string("ssf_couple_reduce_selftest: ") + hipGetErrorString(e);
        return SSF_ERR_HIP;
    }
    return SSF_OK;
}

// entry points of this translation unit (one per precision; engine_fused.hip dispatches)
template <typename T> Engine *make_fused_engine_t(ssf_plan *plan) {
    auto *x = new FusedEngine<T>(plan);
    if (x->init() != SSF_OK) {
        delete x;
        return nullptr;
    }
    return x;
}
template <typename T> FusedRows *make_fused_rows_t(ssf_plan *plan, int64_t N, int nrows) {
    if (!FusedRowsImpl<T>::supports(N)) return nullptr;
    auto *x = new FusedRowsImpl<T>(plan, N, nrows);
    if (x->init() != SSF_OK) {
        delete x;
        return nullptr;
    }
    return x;
}
template <typename T> FusedConv *make_fused_conv_t(ssf_plan *plan, int64_t M, int nrows) {
    auto *x = new FusedConvImpl<T>(plan, M, nrows);
    if (x->init() != SSF_OK) {
        plan->err = x->error();
        delete x;
        return nullptr;
    }
    return x;
}

}  // namespace
}  // namespace ssf
