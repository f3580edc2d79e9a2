// engine_rx.hip -- HIP backend of the receiver front-end pipeline: __global__ wrappers around the
// bodies of rx_kernels.h / ols_body and the Backend that RxCore (rx_pipeline.h) drives.  One
// stream per call; stages are enqueued back to back and nothing is read back between them (the
// decimator's sampling-phase decision is the one exception: 8 x SpSin doubles).
#include <cstring>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "dev_ctx.h"
#include "rx_pipeline.h"
#include "ssf_copy.h"
#include "ssf_internal.h"

namespace ssf {
namespace {

using namespace rx;

struct RxCtx : DevCtxCore {
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int) {}
};
#define SSF_RX_CTX()                                                          \
    extern __shared__ __attribute__((aligned(16))) char ssf_smem[];           \
    RxCtx ctx{{(int)threadIdx.x, (int)blockIdx.x, (int)blockDim.x, (int)gridDim.x, ssf_smem}}

template <int LG, int C> __global__ void __launch_bounds__(fused::ols_threads(LG, C)) k_rx_ols(const fused::OlsArgs<double> a) {
    SSF_RX_CTX();
    fused::ols_body<double, LG, C>(ctx, a);
}
// the coherent receivers' filters with the stages around them in their loads / stores (rx_kernels.h: rx_ols_body)
template <int LG, int C, int PRE, int NOISE> __global__ void __launch_bounds__(fused::ols_threads(LG, C)) k_rx_ols_fused(const RxOlsArgs a) {
    SSF_RX_CTX();
    rx_ols_body<LG, C, PRE, NOISE>(ctx, a);
}
__global__ void __launch_bounds__(256) k_rx_det(const DetKernelArgs a) {
    SSF_RX_CTX();
    det_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_axpy(const AxpyArgs a) {
    SSF_RX_CTX();
    axpy_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_rx_iqf(const IqfArgs a) {
    SSF_RX_CTX();
    iqf_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_rx_front(const FrontArgs a) {
    SSF_RX_CTX();
    front_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_optics(const OpticsArgs a) {
    SSF_RX_CTX();
    optics_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_nlin_phase(const NlinPhaseArgs a) {
    SSF_RX_CTX();
    nlin_phase_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_conv_sums(const ConvSumsArgs a) {
    SSF_RX_CTX();
    conv_sums_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_tx_absmax(const AbsMaxArgs a) {
    SSF_RX_CTX();
    absmax_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_tx_phase_noise(const PnArgs a) {
    SSF_RX_CTX();
    pn_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_tx_iqm(const IqmArgs a) {
    SSF_RX_CTX();
    iqm_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_tx_shift_add(const ShiftAddArgs a) {
    SSF_RX_CTX();
    shift_add_body(ctx, a);
}
template <int LG, int C, int MODE> __global__ void __launch_bounds__(fused::ols_threads(LG, C)) k_rx_chain_ols(const ChainOlsArgs a) {
    SSF_RX_CTX();
    chain_ols_body<LG, C, MODE>(ctx, a);
}
__global__ void __launch_bounds__(1024) k_rx_chain_finish(const ChainFinishArgs a) {
    SSF_RX_CTX();
    chain_finish_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_rx_dec_stats(const DecStatsArgs a) {
    SSF_RX_CTX();
    dec_stats_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_rx_dec_sum(const DecSumArgs a) {
    SSF_RX_CTX();
    dec_sum_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_rx_dec_finish(const DecFinishArgs a) {
    SSF_RX_CTX();
    dec_finish_body(ctx, a);
}
__global__ void __launch_bounds__(256) k_rx_dec_gather(const DecGatherArgs a) {
    SSF_RX_CTX();
    dec_gather_body(ctx, a);
}

struct HipRxBackend {
    hipStream_t st = nullptr;
    hipError_t first_err = hipSuccess;
    std::string where;
    Stager stg;
    void chk(hipError_t e, const char *what) {
        if (e != hipSuccess && first_err == hipSuccess) {
            first_err = e;
            where = what;
        }
    }
    int open(int device) {
        chk(hipSetDevice(device), "hipSetDevice");
        chk(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
        if (ok()) chk(stg.init(), "pinned staging buffers");
        return ok() ? SSF_OK : SSF_ERR_HIP;
    }
    ~HipRxBackend() {
        if (st) (void)hipStreamDestroy(st);
    }
    bool ok() const { return first_err == hipSuccess; }
    bool oom() const { return first_err == hipErrorOutOfMemory; }
    std::string last_error() const { return where + ": " + hipGetErrorString(first_err); }
    void *alloc(size_t n) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            chk(e, "hipMalloc");
            return nullptr;
        }
        return p;
    }
    void free(void *p) { (void)hipFree(p); }
    // small copies go on the SAME stream as their consumers (a non-blocking stream has no implicit ordering with
    // the null stream a plain hipMemcpy uses) and are waited for: the source may be a temporary
    void h2d(void *d, const void *h, size_t n) {
        chk(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st), "hipMemcpyAsync H2D");
        chk(hipStreamSynchronize(st), "hipStreamSynchronize");
    }
    void d2h(void *h, const void *d, size_t n) {
        chk(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st), "hipMemcpyAsync D2H");
        chk(hipStreamSynchronize(st), "hipStreamSynchronize");
    }
    void h2d_big(void *d, const void *h, size_t n) { chk(stg.h2d(d, h, n, st), "staged upload"); }
    void d2h_big(void *h, const void *d, size_t n) { chk(stg.d2h(h, d, n, st), "staged download"); }
    void sync() { chk(hipStreamSynchronize(st), "hipStreamSynchronize"); }
    static unsigned ew_grid(long long n) {
        const long long g = (n + 255) / 256;
        return (unsigned)(g < 1 ? 1 : g > 4096 ? 4096 : g);
    }
    // one instantiation per transform size and columns-per-group (fused_kernels.h: ols_launch / ols_dispatch); a kernel's
    // dynamic-LDS limit is raised the first time this device launches it
    std::set<const void *> armed;
    template <class K> void arm(K kernel) {
        if (!armed.insert((const void *)kernel).second) return;
        chk(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute");
    }
    void launch_ols(const fused::OlsArgs<double> &a) {
        const fused::OlsLaunch o = fused::ols_launch(a.log2nfft, a.nrows, a.njobs);
        fused::ols_dispatch(o, [&](auto lg, auto cc) {
            constexpr int LG = decltype(lg)::value, C = decltype(cc)::value;
            arm(k_rx_ols<LG, C>);
            k_rx_ols<LG, C><<<(unsigned)o.grid, o.threads, o.lds_bytes, st>>>(a);
        });
        chk(hipGetLastError(), "launch k_rx_ols");
    }
    void launch_rx_ols(const RxOlsArgs &a) {
        const fused::OlsLaunch o = fused::ols_launch(a.o.log2nfft, a.o.nrows, a.o.njobs);
        const bool found = rx_ols_dispatch(a, o, [&](auto lg, auto cc, auto pre, auto noise) {
            constexpr int LG = decltype(lg)::value, C = decltype(cc)::value, PRE = decltype(pre)::value, NOISE = decltype(noise)::value;
            arm(k_rx_ols_fused<LG, C, PRE, NOISE>);
            k_rx_ols_fused<LG, C, PRE, NOISE><<<(unsigned)o.grid, o.threads, o.lds_bytes, st>>>(a);
        });
        if (!found) chk(hipErrorInvalidConfiguration, "k_rx_ols_fused: no kernel for this stage / transform size");
        chk(hipGetLastError(), "launch k_rx_ols_fused");
    }
    void launch_det(const DetKernelArgs &a) {
        k_rx_det<<<ew_grid(a.det.N * a.det.nm), 256, 0, st>>>(a);
        chk(hipGetLastError(), "launch k_rx_det");
    }
    void launch_axpy(const AxpyArgs &a) {
        k_axpy<<<ew_grid(a.n), 256, 0, st>>>(a);
        chk(hipGetLastError(), "launch k_axpy");
    }
    void launch_iqf(const IqfArgs &a) {
        k_rx_iqf<<<ew_grid(a.N * a.nm), 256, 0, st>>>(a);
        chk(hipGetLastError(), "launch k_rx_iqf");
    }
    bool is_resident(const void *p) const { return on_device(p); }
    void launch_front(const FrontArgs &a) {
        k_rx_front<<<ew_grid(a.f.N), 256, 0, st>>>(a);
        chk(hipGetLastError(), "launch k_rx_front");
    }
    void memset(void *d, int v, size_t n) { chk(hipMemsetAsync(d, v, n, st), "hipMemsetAsync"); }
    void launch_optics(const OpticsArgs &a) {
        k_optics<<<ew_grid(a.n), 256, 64, st>>>(a);
        chk(hipGetLastError(), "launch k_optics");
    }
    void launch_nlin_phase(const NlinPhaseArgs &a) {
        const long long nb = (a.n + 255) / 256;
        k_nlin_phase<<<(unsigned)std::min<long long>(nb, 16384), 256, 64, st>>>(a);
        chk(hipGetLastError(), "launch k_nlin_phase");
    }
    void launch_conv_sums(const ConvSumsArgs &a, int nblocks) {
        k_conv_sums<<<(unsigned)nblocks, 256, 4096, st>>>(a);
        chk(hipGetLastError(), "launch k_conv_sums");
    }
    void launch_absmax(const AbsMaxArgs &a, int nblocks) {
        k_tx_absmax<<<(unsigned)nblocks, 256, 4096, st>>>(a);
        chk(hipGetLastError(), "launch k_tx_absmax");
    }
    void launch_pn(const PnArgs &a, int nchunks) {
        k_tx_phase_noise<<<(unsigned)nchunks, 256, 256 * sizeof(double), st>>>(a);
        chk(hipGetLastError(), "launch k_tx_phase_noise");
    }
    void launch_iqm(const IqmArgs &a, int nblocks) {
        k_tx_iqm<<<(unsigned)nblocks, 256, 4096, st>>>(a);
        chk(hipGetLastError(), "launch k_tx_iqm");
    }
    void launch_shift_add(const ShiftAddArgs &a) {
        k_tx_shift_add<<<ew_grid(a.N), 256, 0, st>>>(a);
        chk(hipGetLastError(), "launch k_tx_shift_add");
    }
    void launch_chain_ols(const ChainOlsArgs &a, int mode) {
        const fused::OlsLaunch o = fused::ols_launch(a.o.log2nfft, a.o.nrows, a.o.njobs);
        const bool found = chain_ols_dispatch(o, [&](auto lg, auto cc) {
            constexpr int LG = decltype(lg)::value, C = decltype(cc)::value;
            if (mode == CH_STATS) {
                arm(k_rx_chain_ols<LG, C, CH_STATS>);
                k_rx_chain_ols<LG, C, CH_STATS><<<(unsigned)o.grid, o.threads, o.lds_bytes, st>>>(a);
            } else {
                arm(k_rx_chain_ols<LG, C, CH_GATHER>);
                k_rx_chain_ols<LG, C, CH_GATHER><<<(unsigned)o.grid, o.threads, o.lds_bytes, st>>>(a);
            }
        });
        if (!found) chk(hipErrorInvalidConfiguration, "k_rx_chain_ols: no kernel for this transform size");
        chk(hipGetLastError(), "launch k_rx_chain_ols");
    }
    void launch_chain_finish(const ChainFinishArgs &a) {
        // (one workgroup of 1024 threads: the partials of a class are added in 1024 / nclass interleaved chains -- a latency chain of a
        //  few memory round trips instead of a few dozen with 256 threads: 9.3 -> 3 us at 342 x 32 partials)
        k_rx_chain_finish<<<1, 1024, sizeof(double) * (3 * 1024 + 256), st>>>(a);
        chk(hipGetLastError(), "launch k_rx_chain_finish");
    }
    void launch_dec_stats(const DecStatsArgs &a, int nblocks, int nthreads) {
        k_rx_dec_stats<<<(unsigned)nblocks, nthreads, 3 * sizeof(double) * (size_t)nthreads, st>>>(a);
        chk(hipGetLastError(), "launch k_rx_dec_stats");
    }
    void launch_dec_sum(const DecSumArgs &a, int nblocks, int nthreads) {
        k_rx_dec_sum<<<(unsigned)nblocks, nthreads, 2 * sizeof(double) * (size_t)nthreads, st>>>(a);
        chk(hipGetLastError(), "launch k_rx_dec_sum");
    }
    void launch_dec_finish(const DecFinishArgs &a) {
        k_rx_dec_finish<<<1, 256, sizeof(double) * 3 * 256, st>>>(a);
        chk(hipGetLastError(), "launch k_rx_dec_finish");
    }
    void launch_dec_gather(const DecGatherArgs &a) {
        k_rx_dec_gather<<<ew_grid(a.Nout * a.ncols), 256, 0, st>>>(a);
        chk(hipGetLastError(), "launch k_rx_dec_gather");
    }
};

// One backend per (host thread, device), kept between calls: the stream, the pinned staging buffers
// and a small pool of device blocks (hipMalloc / hipHostMalloc per call cost more than the kernels of
// a 2^20-sample receiver).  The pool keeps at most kPoolBytes; larger working sets are freed on return.
struct Pooled : HipRxBackend {
    struct Block {
        void *p;
        size_t n;
        bool used;
    };
    std::vector<Block> blocks;
    static constexpr size_t kPoolBytes = (size_t)2 << 30;
    // device images of filters, kept between calls (RxCore::cached_filter); dropped all at once between two calls when there are
    // too many (never inside a call: its launches may still read them)
    struct Filter {
        std::string key;
        void *p;
    };
    std::vector<Filter> filters;
    static constexpr size_t kMaxFilters = 128;
    void *filter_lookup(const void *key, size_t n) {
        for (auto &f : filters)
            if (f.key.size() == n && std::memcmp(f.key.data(), key, n) == 0) return f.p;
        return nullptr;
    }
    void *filter_store(const void *key, size_t n, const void *host, size_t bytes) {
        const hipError_t before = first_err;        // (an earlier error of this call stays recorded whatever happens here)
        void *p = HipRxBackend::alloc(bytes);
        if (!p) {
            first_err = before;                     // (no room for a cache entry is not an error: the caller uploads per call)
            return nullptr;
        }
        h2d(p, host, bytes);
        filters.push_back(Filter{std::string((const char *)key, n), p});
        return p;
    }
    void drop_filters() {
        for (auto &f : filters) (void)hipFree(f.p);
        filters.clear();
    }
    void *alloc(size_t n) {
        Block *best = nullptr;
        for (auto &b : blocks)
            if (!b.used && b.n >= n && b.n <= 2 * n + (1u << 20) && (!best || b.n < best->n)) best = &b;
        if (best) {
            best->used = true;
            return best->p;
        }
        const hipError_t before = first_err;
        void *p = HipRxBackend::alloc(n);
        if (!p) {                                   // out of memory: drop the idle blocks and try once more
            trim(0);
            first_err = before;
            p = HipRxBackend::alloc(n);
        }
        if (p) blocks.push_back(Block{p, n, true});
        return p;
    }
    void free(void *p) {
        for (auto &b : blocks)
            if (b.p == p) b.used = false;
    }
    void trim(size_t keep) {
        size_t total = 0;
        for (auto &b : blocks) total += b.n;
        for (size_t i = 0; i < blocks.size() && total > keep;) {
            if (!blocks[i].used) {
                total -= blocks[i].n;
                (void)hipFree(blocks[i].p);
                blocks.erase(blocks.begin() + (long)i);
            } else ++i;
        }
    }
    ~Pooled() {
        trim(0);
        drop_filters();
    }
};

Pooled *backend_for(int device, std::string *err) {
    thread_local std::vector<std::pair<int, Pooled *>> cache;
    for (auto &c : cache)
        if (c.first == device) {
            (void)hipSetDevice(device);
            c.second->first_err = hipSuccess;
            return c.second;
        }
    Pooled *be = new Pooled();
    if (be->open(device) != SSF_OK) {
        *err = be->last_error();
        delete be;
        return nullptr;
    }
    cache.emplace_back(device, be);
    return be;
}

template <class F> int with_core(int device, std::string *err, F &&f) {
    Pooled *be = backend_for(device, err);
    if (!be) return SSF_ERR_HIP;
    if (be->filters.size() > Pooled::kMaxFilters) be->drop_filters();
    int rc;
    {
        RxCore<Pooled> core(*be);
        rc = f(core);
        if (rc != SSF_OK) *err = core.err;
    }
    if (rc == SSF_ERR_HIP && be->oom()) rc = SSF_ERR_OOM;
    be->trim(Pooled::kPoolBytes);
    return rc;
}

}  // namespace

int rx_run(int device, int mode, int64_t N, int nmodes, const ssf_rx_params *p, const void *in0, const void *lo,
           const double *un, void *out, std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.run(mode, N, nmodes, *p, in0, lo, un, out); });
}
int rx_fir(int device, int64_t sigLen, int ncols, int ntaps, const void *taps, const void *in, void *out, std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.fir(sigLen, ncols, ntaps, taps, in, out); });
}
int rx_fir_long(int device, int64_t inLen, int64_t outLen, int ncols, int64_t ntaps, const void *taps, int64_t shift, const void *in,
                void *out, std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.fir_long(inLen, outLen, ncols, ntaps, taps, shift, in, out); });
}
int rx_axpy(int device, int64_t n, double alpha, const void *x, void *y, std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.axpy(n, alpha, x, y); });
}
int rx_overlap_save(int device, int64_t sigLen, int ncols, int nfft, int K, const void *Hfft, const void *in, void *out,
                    std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.overlap_save(sigLen, ncols, nfft, K, Hfft, in, out); });
}
int rx_delay(int device, int64_t N, double delay, double Fs, const void *in, void *out, std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.delay(N, delay, Fs, in, out); });
}
int tx_wdm(int device, const ssf_tx_params *p, const void *symbols, const double *taps, const double *phi, const double *amp,
           const double *deltaF, void *out, double *power_out, std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.wdm_tx(*p, symbols, taps, phi, amp, deltaF, out, power_out); });
}
int rx_chain(int device, int64_t N, const ssf_rx_params *p, const void *Es, const void *Elo, const void *taps, int ntaps, int SpSin,
             int decFactor, const void *edcH, int edcK, int edc_nfft, void *out, int32_t *sampDelay, std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) {
        return c.chain(N, *p, Es, Elo, taps, ntaps, SpSin, decFactor, edcH, edcK, edc_nfft, out, sampDelay);
    });
}
int rx_optics(int device, int op, int64_t n, int ncols, double p0, double p1, unsigned long long seed, unsigned row0, const void *a,
              const void *b, void *o0, void *o1, std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.optics(op, n, ncols, p0, p1, seed, row0, a, b, o0, o1); });
}
int mk_nlin_phase(int device, int64_t n, double gamma, const void *Ex, const void *Ey, const double *Pch, double *phi,
                  std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.nlin_phase(n, gamma, Ex, Ey, Pch, phi); });
}
int mk_convergence(int device, int64_t n, const void *xfd, const void *yfd, const void *xc, const void *yc, double *lim,
                   std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.convergence(n, xfd, yfd, xc, yc, lim); });
}
int rx_decimate(int device, int64_t N, int ncols, int SpSin, int decFactor, const void *in, void *out, int32_t *sampDelay,
                std::string *err) {
    return with_core(device, err, [&](RxCore<Pooled> &c) { return c.decimate(N, ncols, SpSin, decFactor, in, out, sampDelay); });
}

}  // namespace ssf
