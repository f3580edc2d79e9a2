// engine_fused.hip -- HIP backend of the fused radix-2^n engine: __global__ wrappers around
// the kernel bodies of fused_kernels.h plus the Backend that FusedCore (fused_engine.h)
// drives.  All launches go to the plan's stream; nothing here synchronises inside a step.
#include <cstdlib>

#include "fused_engine.h"
#include "ssf_internal.h"

namespace ssf {
namespace {

using namespace fused;

struct DevCtx {
    int tid, bid, nthreads, nblocks;
    char *lds;
    __device__ __forceinline__ void sync() { __syncthreads(); }
};

#define SSF_DEV_CTX()                                                         \
    extern __shared__ __attribute__((aligned(16))) char ssf_smem[];           \
    DevCtx ctx{(int)threadIdx.x, (int)blockIdx.x, (int)blockDim.x, (int)gridDim.x, ssf_smem}

// OCC = minimum waves per SIMD the register allocator must leave room for: 1 = up to 512
// registers per lane and no spills (one 256-thread workgroup per CU), 2 = 256 registers
// (two workgroups per CU, some spills in the fp64 kernels).  Picked per plan by $SSF_FUSED_OCC.
template <typename T, int MAXT, int OCC> __global__ void __launch_bounds__(MAXT, OCC) k_row(const RowArgs<T> a) {
    SSF_DEV_CTX();
    row_body<T>(ctx, a);
}
template <typename T, int OCC> __global__ void __launch_bounds__(256, OCC) k_col(const ColArgs<T> a) {
    SSF_DEV_CTX();
    col_body<T>(ctx, a);
}
template <typename T> __global__ void __launch_bounds__(256) k_amp(const AmpArgs<T> a) {
    SSF_DEV_CTX();
    amp_body<T>(ctx, a);
}

struct HipBackend {
    ssf_plan *pl;
    hipError_t first_err = hipSuccess;
    std::string where;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    int row_occ = 2, col_occ = 1;
    explicit HipBackend(ssf_plan *p) : pl(p) {
        if (const char *s = getenv("SSF_FUSED_ROW_OCC")) row_occ = atoi(s) == 1 ? 1 : 2;
        if (const char *s = getenv("SSF_FUSED_COL_OCC")) col_occ = atoi(s) == 2 ? 2 : 1;
        chk(hipEventCreate(&ev0), "hipEventCreate");
        chk(hipEventCreate(&ev1), "hipEventCreate");
    }
    ~HipBackend() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
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
        return ms;
    }
    template <typename F> void set_lds(F f, size_t bytes) {
        chk(hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    void prepare(size_t row_lds, size_t col_lds) {
        set_lds(k_row<double, 256, 1>, row_lds);
        set_lds(k_row<double, 256, 2>, row_lds);
        set_lds(k_row<double, 512, 1>, row_lds);
        set_lds(k_row<double, 1024, 1>, row_lds);
        set_lds(k_row<float, 256, 1>, row_lds);
        set_lds(k_row<float, 256, 2>, row_lds);
        set_lds(k_row<float, 512, 1>, row_lds);
        set_lds(k_row<float, 1024, 1>, row_lds);
        set_lds(k_col<double, 1>, col_lds);
        set_lds(k_col<double, 2>, col_lds);
        set_lds(k_col<float, 1>, col_lds);
        set_lds(k_col<float, 2>, col_lds);
    }
    template <typename T> void launch_row(const RowArgs<T> &a, int grid, int block, size_t lds) {
        if (block <= 256 && row_occ == 2)
            k_row<T, 256, 2><<<grid, block, lds, pl->stream>>>(a);
        else if (block <= 256)
            k_row<T, 256, 1><<<grid, block, lds, pl->stream>>>(a);
        else if (block <= 512)
            k_row<T, 512, 1><<<grid, block, lds, pl->stream>>>(a);
        else
            k_row<T, 1024, 1><<<grid, block, lds, pl->stream>>>(a);
        chk(hipGetLastError(), "launch k_row");
    }
    template <typename T> void launch_col(const ColArgs<T> &a, int grid, int block, size_t lds) {
        if (col_occ == 2)
            k_col<T, 2><<<grid, block, lds, pl->stream>>>(a);
        else
            k_col<T, 1><<<grid, block, lds, pl->stream>>>(a);
        chk(hipGetLastError(), "launch k_col");
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
    explicit FusedEngine(ssf_plan *p) : pl(p), be(p), core(be, p->N, p->nrows, p->precision) {}
    int id() const override { return SSF_ENGINE_FUSED; }
    int ret(int rc) {
        if (rc != SSF_OK) pl->err = core.err.empty() ? be.last_error() : core.err;
        return rc;
    }
    int init() { return ret(core.init()); }
    int upload(const void *soa) override { return ret(core.upload(soa)); }
    int download(void *soa) override { return ret(core.download(soa)); }
    int download_snapshots(void *soa) override { return ret(core.download_snapshots(soa)); }
    int execute(const ssf_params &p, int s0, int s1, const void *noise, ssf_stats *st, ssf_trace *tr) override {
        return ret(core.execute(p, s0, s1, noise, st, tr));
    }
    int linear_channel(double Fs, double Fc, double alpha, double D, double L) override {
        return ret(core.linear_channel(Fs, Fc, alpha, D, L));
    }
};

}  // namespace

bool fused_supports(int64_t N, int nrows, int precision) {
    if (N < 256 || (N & (N - 1)) || nrows < 1) return false;
    int l = 0;
    while ((1ll << l) < N) ++l;
    fused::Split s;
    return fused::choose_split(l, precision, &s);
}

Engine *make_fused_engine(ssf_plan *plan) {
    int rc;
    Engine *e;
    if (plan->precision == SSF_C128) {
        auto *x = new FusedEngine<double>(plan);
        rc = x->init();
        e = x;
    } else {
        auto *x = new FusedEngine<float>(plan);
        rc = x->init();
        e = x;
    }
    if (rc != SSF_OK) {
        delete e;
        return nullptr;
    }
    return e;
}

}  // namespace ssf
