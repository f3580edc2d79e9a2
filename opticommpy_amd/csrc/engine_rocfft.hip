// engine_rocfft.hip -- general-length engine: batched length-N transforms + fused elementwise HIP
// kernels.  Handles every N (the reference accepts any length; typical notebooks use
// N = SpS * Nsymbols, not a power of two).  Host drives the data-dependent control flow
// (one 16-byte read-back per fixed-point iteration), so this engine is the generality /
// cross-check path; the roofline engine is engine_fused.hip.
//
// Two transform providers:
//   * Bluestein on the fused kernels (product path for the lengths the fused pipeline does not take natively: a prime
//     factor above 5, too few factors of two, ...): X[k] = w[k] sum_n (x[n] w[n]) conj(w)[k - n], w[n] = exp(-j pi n^2 / N),
//     i.e. one circular convolution of length M = 2^m >= 2N - 1 with a fixed kernel = FusedConv (engine_fused.hip):
//     chirp-multiply + zero-pad, column FWD, row FFT . B . IFFT, column INV, chirp-multiply: five launches per transform;
//   * rocFFT (SSF_ENGINE_ROCFFT on request): the independent on-GPU cross-check of all hand-written transforms.
//
// Reference semantics followed: optic/models/channels.py:215-238 (ssfm),
// :380-456 (manakovSSF), optic/dsp/equalization.py:1087-1160 (manakovDBP).
#include <rocfft/rocfft.h>

#include <mutex>

#include "dev_ctx.h"
#include "fused_kernels.h"
#include "rx_pipeline.h"
#include "ssf_internal.h"
#include "ssf_rng.h"

namespace ssf {
namespace {

template <typename T> struct Cx;
template <> struct Cx<float> { using type = float2; };
template <> struct Cx<double> { using type = double2; };

template <typename T> __device__ __forceinline__ void sincos_t(T x, T *s, T *c);
template <> __device__ __forceinline__ void sincos_t<float>(float x, float *s, float *c) { sincosf(x, s, c); }
template <> __device__ __forceinline__ void sincos_t<double>(double x, double *s, double *c) { sincos(x, s, c); }

constexpr int kBlock = 256;
constexpr int kMaxPartials = 2048;

__host__ inline int grid_for(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return (int)(g > kMaxPartials ? kMaxPartials : (g < 1 ? 1 : g));
}

// lin[i] = scale * exp(a*hzh) * cis(b * w_i^2 * hzh),  w_i = w_scale * fftfreq(N)[i]
template <typename T>
__global__ void k_make_lin(typename Cx<T>::type *lin, int64_t N, double w_scale, double a, double b,
                           double hzh, double scale) {
    const double mag = exp(a * hzh) * scale;
    const int64_t npos = (N + 1) / 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t kk = i < npos ? i : i - N;
        const double w = w_scale * ((double)kk / (double)N);
        double s, c;
        sincos(b * (w * w) * hzh, &s, &c);
        typename Cx<T>::type v;
        v.x = (T)(mag * c);
        v.y = (T)(mag * s);
        lin[i] = v;
    }
}

template <typename T>
__global__ void k_mul_lin(typename Cx<T>::type *F, const typename Cx<T>::type *lin, int64_t N, int nrows) {
    const int64_t total = N * nrows;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const auto l = lin[i % N];
        auto f = F[i];
        typename Cx<T>::type o;
        o.x = f.x * l.x - f.y * l.y;
        o.y = f.x * l.y + f.y * l.x;
        F[i] = o;
    }
}

// scalar NLSE nonlinear step: E *= exp(j * g_hz * |E|^2)      (channels.py:225)
template <typename T>
__global__ void k_nl_nlse(typename Cx<T>::type *E, int64_t total, T scale, T g_hz) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        auto e = E[i];
        e.x *= scale;
        e.y *= scale;
        T s, c;
        sincos_t<T>(g_hz * (e.x * e.x + e.y * e.y), &s, &c);
        typename Cx<T>::type o;
        o.x = e.x * c - e.y * s;
        o.y = e.x * s + e.y * c;
        E[i] = o;
    }
}

__device__ __forceinline__ double block_reduce_sum(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_reduce_max(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    double r = -INFINITY;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r = fmax(r, sh[i]);
    __syncthreads();
    return r;
}

// Pch = |Ex|^2 + |Ey|^2 at the step start, and the block max of
// phi = c8g * (Pch + |Ex|^2 + |Ey|^2) / 2 (E_conv == E at every step start).
template <typename T>
__global__ void k_power(const typename Cx<T>::type *E, T *P, int K, int64_t N, T c8g, double *pmax) {
    __shared__ double sh[kBlock / 64];
    const int64_t total = (int64_t)K * N;
    double m = -INFINITY;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i / N, n = i - k * N;
        const auto x = E[(2 * k) * N + n];
        const auto y = E[(2 * k + 1) * N + n];
        const T ax = x.x * x.x + x.y * x.y, ay = y.x * y.x + y.y * y.y;
        const T p = ax + ay;
        P[i] = p;
        const T phi = c8g * (p + ax + ay) / (T)2;
        m = fmax(m, (double)phi);
    }
    m = block_reduce_max(m, sh);
    if (threadIdx.x == 0) pmax[blockIdx.x] = m;
}

// E_fd = E_hd * exp(j * shz * phi), phi = c8g (Pch + |Ecx|^2 + |Ecy|^2)/2   (channels.py:414-417, 493)
template <typename T>
__global__ void k_rot(const typename Cx<T>::type *Ehd, const T *P, const typename Cx<T>::type *Ec,
                      typename Cx<T>::type *Efd, int K, int64_t N, T c8g, T shz) {
    const int64_t total = (int64_t)K * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i / N, n = i - k * N;
        const int64_t ix = (2 * k) * N + n, iy = ix + N;
        const auto cx = Ec[ix], cy = Ec[iy];
        const T phi = c8g * (P[i] + (cx.x * cx.x + cx.y * cx.y) + (cy.x * cy.x + cy.y * cy.y)) / (T)2;
        T s, c;
        sincos_t<T>(shz * phi, &s, &c);
        const auto hx = Ehd[ix], hy = Ehd[iy];
        typename Cx<T>::type ox, oy;
        ox.x = hx.x * c - hx.y * s;
        ox.y = hx.x * s + hx.y * c;
        oy.x = hy.x * c - hy.y * s;
        oy.y = hy.x * s + hy.y * c;
        Efd[ix] = ox;
        Efd[iy] = oy;
    }
}

// partial sums of |E_fd - E_conv|^2 and |E_conv|^2 over all rows (channels.py:517-519)
template <typename T>
__global__ void k_conv(const typename Cx<T>::type *Efd, const typename Cx<T>::type *Ec, int64_t total,
                       double *pnum, double *pden) {
    __shared__ double sh[kBlock / 64];
    double num = 0, den = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const auto f = Efd[i], c = Ec[i];
        const double dx = (double)f.x - (double)c.x, dy = (double)f.y - (double)c.y;
        num += dx * dx + dy * dy;
        den += (double)c.x * c.x + (double)c.y * c.y;
    }
    num = block_reduce_sum(num, sh);
    den = block_reduce_sum(den, sh);
    if (threadIdx.x == 0) {
        pnum[blockIdx.x] = num;
        pden[blockIdx.x] = den;
    }
}

// out[0] = sum(a[0..n)), out[1] = sum(b[0..n)), out[2] = max(c[0..n))  (any pointer may be null)
__global__ void k_finish(const double *a, const double *b, const double *c, int n, double *out) {
    __shared__ double sh[kBlock / 64];
    double sa = 0, sb = 0, mc = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (a) sa += a[i];
        if (b) sb += b[i];
        if (c) mc = fmax(mc, c[i]);
    }
    sa = block_reduce_sum(sa, sh);
    sb = block_reduce_sum(sb, sh);
    mc = block_reduce_max(mc, sh);
    if (threadIdx.x == 0) {
        out[0] = sa;
        out[1] = sb;
        out[2] = mc;
    }
}

// span epilogue: E = E*gain (+ noise)      (channels.py:443-451, devices.py:726)
template <typename T>
__global__ void k_amp(typename Cx<T>::type *E, int64_t total, T gain, const typename Cx<T>::type *noise,
                      double sigma = 0.0, unsigned long long seed = 0, unsigned span = 0, int64_t N = 1, unsigned row0 = 0) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        auto e = E[i];
        e.x *= gain;
        e.y *= gain;
        if (noise) {
            e.x += noise[i].x;
            e.y += noise[i].y;
        }
        if (sigma > 0) {       // device-generated ASE (same generator as the fused engine)
            double re, im;
            gauss_pair((unsigned long long)(i % N), row0 + (unsigned)(i / N), span, seed, sigma, re, im);
            e.x += (T)re;
            e.y += (T)im;
        }
        E[i] = e;
    }
}

// Bluestein: wk[r][n] = in[r][n] * ch[n] (conj(ch) for the inverse) for n < N, zero up to M
template <typename T>
__global__ void k_blue_pre(const typename Cx<T>::type *in, typename Cx<T>::type *wk, const double2 *ch, int64_t N, int64_t M,
                           int nrows, int inverse) {
    const int64_t total = M * nrows;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / M, n = i - r * M;
        typename Cx<T>::type o;
        o.x = 0;
        o.y = 0;
        if (n < N) {
            const auto e = in[r * N + n];
            const double c = ch[n].x, sn = inverse ? -ch[n].y : ch[n].y;
            o.x = (T)((double)e.x * c - (double)e.y * sn);
            o.y = (T)((double)e.x * sn + (double)e.y * c);
        }
        wk[i] = o;
    }
}
// out[r][k] = wk[r][k] * ch[k] (conj for the inverse), k < N
template <typename T>
__global__ void k_blue_post(const typename Cx<T>::type *wk, typename Cx<T>::type *out, const double2 *ch, int64_t N, int64_t M,
                            int nrows, int inverse) {
    const int64_t total = N * nrows;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / N, k = i - r * N;
        const auto e = wk[r * M + k];
        const double c = ch[k].x, sn = inverse ? -ch[k].y : ch[k].y;
        typename Cx<T>::type o;
        o.x = (T)((double)e.x * c - (double)e.y * sn);
        o.y = (T)((double)e.x * sn + (double)e.y * c);
        out[i] = o;
    }
}

// Bluestein for M <= 8192 (N <= 4096): the whole transform in ONE launch -- chirp-multiply and zero-pad on load, M-point
// forward transform in LDS, x B (natural-order spectrum of the chirp kernel, 1 / M folded in), inverse transform, chirp-multiply
// on store: the overlap-save kernel's structure (fused_kernels.h: ols_body) with one block per row.  The three-launch
// FusedConv path costs five launches per transform, and at these sizes every launch is a latency chain.
template <typename T> struct BlueArgs {
    const fused::cx<T> *in;
    fused::cx<T> *out;
    const double2 *ch;
    const fused::cx<T> *Bhat;       // M values
    long long N, njobs;
    int log2M, inverse;
};
template <typename T, int MAXT> __global__ void __launch_bounds__(MAXT) k_blue(const BlueArgs<T> a) {
    using namespace fused;
    extern __shared__ __attribute__((aligned(16))) char blue_smem[];
    DevCtxCore ctx{(int)threadIdx.x, (int)blockIdx.x, (int)blockDim.x, (int)gridDim.x, blue_smem};
    const PassPlan p = make_plan(a.log2M);
    const int fpw = ctx.nthreads / p.tpf, f = ctx.tid / p.tpf, b = ctx.tid % p.tpf;
    const long long job = (long long)ctx.bid * fpw + f;
    const bool live = job < a.njobs;                       // idle threads still take part in the barriers
    cx<T> *l = (cx<T> *)ctx.lds + (size_t)f * lds_slots_per_fft(p.L);
    cx<T> v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const long long n = b + p.tpf * q;
        v[q] = mk<T>((T)0, (T)0);
        if (live && n < a.N) {
            const cx<T> e = a.in[job * a.N + n];
            const double c = a.ch[n].x, sn = a.inverse ? -a.ch[n].y : a.ch[n].y;
            v[q] = mk<T>((T)((double)e.re * c - (double)e.im * sn), (T)((double)e.re * sn + (double)e.im * c));
        }
    }
    fft_dif<-1>(ctx, p, b, v, l);
    const int last = p.npass - 1;
#pragma unroll
    for (int idx = 0; idx < 16; ++idx) v[idx] = v[idx] * a.Bhat[rev_pos(p, reg_pos(p, last, b, idx))];
    fft_dit<+1>(ctx, p, b, v, l);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const long long k = b + p.tpf * q;
        if (live && k < a.N) {
            const double c = a.ch[k].x, sn = a.inverse ? -a.ch[k].y : a.ch[k].y;
            a.out[job * a.N + k] = mk<T>((T)((double)v[q].re * c - (double)v[q].im * sn), (T)((double)v[q].re * sn + (double)v[q].im * c));
        }
    }
}

std::once_flag g_rocfft_once;

template <typename T> class RocfftEngine final : public Engine {
    using C = typename Cx<T>::type;
    ssf_plan *pl;
    int64_t N;
    int nrows;
    size_t row_bytes, field_bytes;
    C *bufA = nullptr, *bufB = nullptr, *Ehd = nullptr, *F = nullptr, *lin = nullptr, *noise_d = nullptr;
    T *P = nullptr;
    double *part = nullptr;   // 3 * kMaxPartials partials + 4 results
    double *res_h = nullptr;  // pinned host, 4 doubles
    rocfft_plan fwd = nullptr, inv = nullptr;     // (the two plans double as the direction tags of fft())
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
    const bool blue;                              // transforms by Bluestein on the fused kernels instead of rocFFT
    int64_t M = 0;
    FusedConv *conv = nullptr;                    // M > 8192: three fused launches per convolution
    FusedRows *rows = nullptr;                    // N = 2^a 3^b 5^c <= 8192: FFT . H . IFFT of a row in ONE launch (no Bluestein)
    fused::cx<T> *Bhat[2] = {nullptr, nullptr};   // M <= 8192: spectra of the two chirp kernels for the one-launch transform
    double2 *chirp = nullptr;                     // exp(-j pi n^2 / N), n < N
    std::vector<C *> snaps;
    C *E = nullptr;           // current field (points at bufA or bufB)
    double lin_hz = NAN, lin_scale = NAN, lin_a = NAN, lin_b = NAN, lin_w = NAN;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ssf_reduce_fn reduce = nullptr;               // coupled batch across plans (ssf_set_coupling)
    void *reduce_ctx = nullptr;

    int coupled(double *values, int n, int op) {
        if (!reduce) return SSF_OK;
        const int rc = reduce(reduce_ctx, values, n, op);
        return rc ? fail(pl, SSF_ERR_COMM, "coupling reducer failed") : SSF_OK;
    }

  public:
    int set_coupling(ssf_reduce_fn f, void *ctx) override {
        reduce = f;
        reduce_ctx = ctx;
        return SSF_OK;
    }
    RocfftEngine(ssf_plan *p, bool bluestein) : pl(p), N(p->N), nrows(p->nrows), blue(bluestein) {
        row_bytes = sizeof(C) * (size_t)N;
        field_bytes = row_bytes * (size_t)nrows;
    }
    // (with Bluestein every transform runs on the fused kernels: reported as the fused engine)
    int id() const override { return blue ? SSF_ENGINE_FUSED : SSF_ENGINE_ROCFFT; }
    int pipeline() const override { return !blue ? SSF_PIPE_ROCFFT : rows ? SSF_PIPE_ROWS : SSF_PIPE_BLUESTEIN; }

    static int64_t bluestein_length(int64_t n) {
        int64_t m = 256;
        while (m < 2 * n - 1) m *= 2;
        return m;
    }
    int init_bluestein() {
        M = bluestein_length(N);
        const bool one_launch = M <= 8192;
        if (one_launch) {                            // the one-launch transform needs more than the default dynamic LDS: once per plan
            SSF_HIP(pl, hipFuncSetAttribute((const void *)k_blue<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            SSF_HIP(pl, hipFuncSetAttribute((const void *)k_blue<T, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        if (!one_launch) {
            conv = make_fused_conv(pl, M, nrows);
            if (!conv) return pl->err.find("out of memory") != std::string::npos ? SSF_ERR_OOM : SSF_ERR_UNSUPPORTED;
        }
        // chirp with the exact quadratic residue: n^2 mod 2N in integers, angle = -pi r / N
        std::vector<double2> ch((size_t)N);
        for (int64_t n = 0; n < N; ++n) {
            const int64_t r = (int64_t)(((unsigned __int128)n * (unsigned __int128)n) % (unsigned __int128)(2 * N));
            const double a = -3.14159265358979323846 * ((double)r / (double)N);
            ch[(size_t)n].x = std::cos(a);
            ch[(size_t)n].y = std::sin(a);
        }
        SSF_HIP(pl, hipMalloc(&chirp, sizeof(double2) * (size_t)N));
        SSF_HIP(pl, hipMemcpy(chirp, ch.data(), sizeof(double2) * (size_t)N, hipMemcpyHostToDevice));
        // convolution kernels: forward b[m] = conj(w)[|m|] / M, inverse conj of it, |m| < N, wrapped into length M
        std::vector<C> b((size_t)M);
        for (int which = 0; which < 2; ++which) {
            std::fill(b.begin(), b.end(), C{0, 0});
            for (int64_t m = 0; m < N; ++m) {
                C v;
                v.x = (T)(ch[(size_t)m].x / (double)M);
                v.y = (T)((which == 0 ? -ch[(size_t)m].y : ch[(size_t)m].y) / (double)M);
                b[(size_t)m] = v;
                if (m) b[(size_t)(M - m)] = v;
            }
            if (one_launch) {                        // natural-order spectrum of the kernel, computed on the host in double
                std::vector<rx::zc> hb((size_t)M, rx::zc(0.0, 0.0));      // (from the exact chirp, not from its rounded copy)
                for (int64_t m = 0; m < N; ++m) {
                    const rx::zc v(ch[(size_t)m].x / (double)M, (which == 0 ? -ch[(size_t)m].y : ch[(size_t)m].y) / (double)M);
                    hb[(size_t)m] = v;
                    if (m) hb[(size_t)(M - m)] = v;
                }
                rx::host_fft(hb, -1);
                std::vector<fused::cx<T>> hs((size_t)M);
                for (int64_t m = 0; m < M; ++m) {
                    hs[(size_t)m].re = (T)hb[(size_t)m].real();
                    hs[(size_t)m].im = (T)hb[(size_t)m].imag();
                }
                SSF_HIP(pl, hipMalloc(&Bhat[which], sizeof(fused::cx<T>) * (size_t)M));
                SSF_HIP(pl, hipMemcpy(Bhat[which], hs.data(), sizeof(fused::cx<T>) * (size_t)M, hipMemcpyHostToDevice));
                continue;
            }
            int rc = conv->set_kernel(which, b.data());
            if (rc) return fail(pl, rc, "Bluestein kernel: " + conv->error());
        }
        return SSF_OK;
    }

    int init() {
        if (!blue) std::call_once(g_rocfft_once, [] { rocfft_setup(); });
        SSF_HIP(pl, hipMalloc(&bufA, field_bytes));
        SSF_HIP(pl, hipMalloc(&bufB, field_bytes));
        SSF_HIP(pl, hipMalloc(&Ehd, field_bytes));
        SSF_HIP(pl, hipMalloc(&F, field_bytes));
        SSF_HIP(pl, hipMalloc(&lin, row_bytes));
        SSF_HIP(pl, hipMalloc(&P, sizeof(T) * (size_t)N * (size_t)((nrows + 1) / 2)));
        SSF_HIP(pl, hipMalloc(&part, sizeof(double) * (3 * kMaxPartials + 4)));
        SSF_HIP(pl, hipHostMalloc(&res_h, 4 * sizeof(double)));
        SSF_HIP(pl, hipEventCreate(&ev0));
        SSF_HIP(pl, hipEventCreate(&ev1));
        E = bufA;
        if (blue) {
            fwd = (rocfft_plan)(void *)this;             // direction tags only (never dereferenced)
            inv = (rocfft_plan)(void *)&M;
            rows = make_fused_rows(pl, N, nrows);        // (nullptr unless the length is short and 5-smooth)
            return init_bluestein();
        }
        const size_t len = (size_t)N;
        const rocfft_precision pr = sizeof(T) == 8 ? rocfft_precision_double : rocfft_precision_single;
        if (rocfft_plan_create(&fwd, rocfft_placement_notinplace, rocfft_transform_type_complex_forward, pr, 1, &len,
                               (size_t)nrows, nullptr) != rocfft_status_success ||
            rocfft_plan_create(&inv, rocfft_placement_notinplace, rocfft_transform_type_complex_inverse, pr, 1, &len,
                               (size_t)nrows, nullptr) != rocfft_status_success)
            return fail(pl, SSF_ERR_FFT, "rocfft_plan_create failed");
        size_t w1 = 0, w2 = 0;
        rocfft_plan_get_work_buffer_size(fwd, &w1);
        rocfft_plan_get_work_buffer_size(inv, &w2);
        const size_t w = w1 > w2 ? w1 : w2;
        if (rocfft_execution_info_create(&info) != rocfft_status_success)
            return fail(pl, SSF_ERR_FFT, "rocfft_execution_info_create failed");
        if (w) {
            SSF_HIP(pl, hipMalloc(&work, w));
            rocfft_execution_info_set_work_buffer(info, work, w);
        }
        rocfft_execution_info_set_stream(info, pl->stream);
        E = bufA;
        return SSF_OK;
    }

    ~RocfftEngine() override {
        if (fwd && !blue) rocfft_plan_destroy(fwd);
        if (inv && !blue) rocfft_plan_destroy(inv);
        if (info) rocfft_execution_info_destroy(info);
        delete conv;
        delete rows;
        for (auto *q : Bhat)
            if (q) (void)hipFree(q);
        if (chirp) (void)hipFree(chirp);
        for (void *p : {(void *)bufA, (void *)bufB, (void *)Ehd, (void *)F, (void *)lin, (void *)P, (void *)part,
                        (void *)work, (void *)noise_d})
            if (p) (void)hipFree(p);
        for (C *s : snaps) (void)hipFree(s);
        if (res_h) (void)hipHostFree(res_h);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }

    int upload(const void *field, bool aos) override {
        E = bufA;
        hipError_t e = pl->stager.h2d(aos ? F : E, field, field_bytes, pl->stream);
        if (e != hipSuccess) return fail(pl, SSF_ERR_HIP, std::string("upload: ") + hipGetErrorString(e));
        if (aos) {
            k_aos_to_soa<C><<<grid_for(N * nrows), kBlock, 0, pl->stream>>>(F, E, N, nrows);
            SSF_HIP(pl, hipStreamSynchronize(pl->stream));
        }
        for (C *s : snaps) (void)hipFree(s);
        snaps.clear();
        n_sunk = 0;
        return SSF_OK;
    }
    int n_snapshots() const override { return (int)snaps.size(); }
    int download(void *field, int which, bool aos) override {
        const C *src = which < 0 ? E : snaps[(size_t)which];
        if (aos) {
            k_soa_to_aos<C><<<grid_for(N * nrows), kBlock, 0, pl->stream>>>(src, F, N, nrows);
            src = F;
        }
        hipError_t e = pl->stager.d2h(field, src, field_bytes, pl->stream);
        if (e != hipSuccess) return fail(pl, SSF_ERR_HIP, std::string("download: ") + hipGetErrorString(e));
        return SSF_OK;
    }

  private:
    int fft(rocfft_plan p, C *in, C *out) {
        if (blue && !conv) {                         // M <= 8192: one launch per batched transform
            const int inverse = p == inv ? 1 : 0;
            int lg = 0;
            while ((1ll << lg) < M) ++lg;
            BlueArgs<T> a{(const fused::cx<T> *)in, (fused::cx<T> *)out, chirp, Bhat[inverse], (long long)N, (long long)nrows, lg, inverse};
            const int tpf = (int)(M / 16), block = tpf >= 256 ? tpf : 256, fpw = block / tpf;
            const int grid = (nrows + fpw - 1) / fpw;
            const size_t lds = (size_t)fpw * fused::lds_slots_per_fft((int)M) * sizeof(fused::cx<T>);
            if (block <= 256) k_blue<T, 256><<<grid, block, lds, pl->stream>>>(a);      // (LDS cap raised once, init_bluestein)
            else k_blue<T, 512><<<grid, block, lds, pl->stream>>>(a);
            SSF_HIP(pl, hipGetLastError());
            return SSF_OK;
        }
        if (blue) {
            const int inverse = p == inv ? 1 : 0;
            C *wk = (C *)conv->work();
            k_blue_pre<T><<<grid_for(M * nrows), kBlock, 0, pl->stream>>>(in, wk, chirp, N, M, nrows, inverse);
            SSF_HIP(pl, hipGetLastError());
            if (int rc = conv->run(inverse)) return fail(pl, rc, "Bluestein convolution: " + conv->error());
            k_blue_post<T><<<grid_for(N * nrows), kBlock, 0, pl->stream>>>(wk, out, chirp, N, M, nrows, inverse);
            SSF_HIP(pl, hipGetLastError());
            return SSF_OK;
        }
        void *ib[1] = {in}, *ob[1] = {out};
        if (rocfft_execute(p, ib, ob, info) != rocfft_status_success) return fail(pl, SSF_ERR_FFT, "rocfft_execute failed");
        return SSF_OK;
    }
    void ensure_lin(const Derived &d, double hzh, double scale, double a, double b) {
        if (hzh == lin_hz && scale == lin_scale && a == lin_a && b == lin_b && d.w_scale == lin_w) return;
        k_make_lin<T><<<grid_for(N), kBlock, 0, pl->stream>>>(lin, N, d.w_scale, a, b, hzh, scale);
        lin_hz = hzh; lin_scale = scale; lin_a = a; lin_b = b; lin_w = d.w_scale;
    }
    // out = ifft(fft(in) * lin)  with 1/N folded into lin
    int lin_step(C *in, C *out) {
        if (rows) {                                  // one launch: the row lives in LDS, the operator comes from the bin index
            const int rc = rows->lin(in, out, lin_hz, lin_a, lin_b, lin_w, lin_scale);
            return rc ? fail(pl, rc, "row transform: " + rows->error()) : SSF_OK;
        }
        int rc = fft(fwd, in, F);
        if (rc) return rc;
        k_mul_lin<T><<<grid_for(N * nrows), kBlock, 0, pl->stream>>>(F, lin, N, nrows);
        return fft(inv, F, out);
    }
    int read_results() {
        SSF_HIP(pl, hipMemcpyAsync(res_h, part + 3 * kMaxPartials, 4 * sizeof(double), hipMemcpyDeviceToHost, pl->stream));
        SSF_HIP(pl, hipStreamSynchronize(pl->stream));
        return SSF_OK;
    }
    int n_sunk = 0;              // snapshots handed to the plan's sink since the last upload
    int snapshot() {
        if (pl->sink.active()) {                     // streamed out (ssf_snapshots.h), not kept
            SSF_HIP(pl, pl->sink.capture(E, (long long)pl->N, pl->nrows, pl->stream));
            ++n_sunk;
            return SSF_OK;
        }
        C *s = nullptr;
        SSF_HIP(pl, hipMalloc(&s, field_bytes));
        snaps.push_back(s);
        SSF_HIP(pl, hipMemcpyAsync(s, E, field_bytes, hipMemcpyDeviceToDevice, pl->stream));
        return SSF_OK;
    }
    static bool wants_snapshot(const ssf_params &p, int span) {
        for (int i = 0; i < p.n_save; ++i)
            if (p.save_spans[i] == span) return true;
        return false;
    }
    int amp_fwd(const ssf_params &p, const Derived &d, int span, int span_rel, const void *noise, double ideal_gain) {
        const int64_t total = N * nrows;
        if (p.amp == SSF_AMP_EDFA) {
            const C *nz = nullptr;
            if (noise) {
                if (!noise_d) SSF_HIP(pl, hipMalloc(&noise_d, field_bytes));
                SSF_HIP(pl, hipMemcpyAsync(noise_d, (const char *)noise + (size_t)span_rel * field_bytes, field_bytes,
                                           hipMemcpyDefault, pl->stream));
                nz = noise_d;
            }
            const bool dev_noise = !noise && p.rng_seed != 0;
            k_amp<T><<<grid_for(total), kBlock, 0, pl->stream>>>(E, total, (T)std::sqrt(d.G_lin), nz,
                                                                  dev_noise ? std::sqrt(d.p_noise / 2) : 0.0,
                                                                  (unsigned long long)p.rng_seed, (unsigned)span, N, (unsigned)p.rng_row_offset);
        } else if (p.amp == SSF_AMP_IDEAL) {
            k_amp<T><<<grid_for(total), kBlock, 0, pl->stream>>>(E, total, (T)ideal_gain, nullptr);
        }
        return SSF_OK;
    }

    int run_nlse(const ssf_params &p, const Derived &d, int s0, int s1, const void *noise, ssf_stats *st) {
        const int nsteps = (int)std::floor(p.Lspan / p.hz);
        const int64_t total = N * nrows;
        C *other = (E == bufA) ? bufB : bufA;
        for (int span = s0; span <= s1; ++span) {
            ensure_lin(d, p.hz / 2, 1.0, d.lin_a, d.lin_b);
            int rc = fft(fwd, E, F);                                       // channels.py:216
            if (rc) return rc;
            for (int s = 0; s < nsteps; ++s) {
                k_mul_lin<T><<<grid_for(total), kBlock, 0, pl->stream>>>(F, lin, N, nrows);
                if ((rc = fft(inv, F, other))) return rc;
                k_nl_nlse<T><<<grid_for(total), kBlock, 0, pl->stream>>>(other, total, (T)(1.0 / (double)N),
                                                                          (T)(p.gamma * p.hz));
                if ((rc = fft(fwd, other, F))) return rc;
                k_mul_lin<T><<<grid_for(total), kBlock, 0, pl->stream>>>(F, lin, N, nrows);
            }
            if ((rc = fft(inv, F, other))) return rc;                       // channels.py:232
            k_amp<T><<<grid_for(total), kBlock, 0, pl->stream>>>(other, total, (T)(1.0 / (double)N), nullptr);
            std::swap(E, other);
            if ((rc = amp_fwd(p, d, span, span - s0, noise, std::exp(d.alpha_lin / 2 * nsteps * p.hz)))) return rc;
            if (wants_snapshot(p, span) && (rc = snapshot())) return rc;
            st->steps += nsteps;
            st->transforms += (int64_t)nrows * (2 * (int64_t)nsteps + 2);
        }
        return SSF_OK;
    }

    int run_manakov(const ssf_params &p, const Derived &d, int s0, int s1, const void *noise, ssf_stats *st,
                    TraceSink &ts) {
        const int K = nrows / 2;
        const int64_t total = N * nrows, ktotal = N * K;
        const double sgn = p.direction >= 0 ? 1.0 : -1.0;
        const int gp = grid_for(ktotal), gc = grid_for(total);
        std::vector<double> lims((size_t)(p.maxIter > 0 ? p.maxIter : 1));
        double *pmax = part, *pnum = part + kMaxPartials, *pden = part + 2 * kMaxPartials, *res = part + 3 * kMaxPartials;
        int rc;
        for (int span = s0; span <= s1; ++span) {
            if (p.direction < 0 && (p.amp == SSF_AMP_EDFA || p.amp == SSF_AMP_IDEAL))    // equalization.py:1090-1092
                k_amp<T><<<gc, kBlock, 0, pl->stream>>>(E, total, (T)std::exp(-d.alpha_lin / 2 * p.Lspan), nullptr);
            double z = 0;
            while (z < p.Lspan) {
                C *Ec = E;                                      // E_conv == E at every step start
                C *X = (E == bufA) ? bufB : bufA;               // receives the next iterate
                k_power<T><<<gp, kBlock, 0, pl->stream>>>(E, P, K, N, (T)d.c8g, pmax);
                double hz_;
                if (p.nlprMethod) {                             // channels.py:392-397
                    k_finish<<<1, kBlock, 0, pl->stream>>>(nullptr, nullptr, pmax, gp, res);
                    if ((rc = read_results())) return rc;
                    if ((rc = coupled(res_h + 2, 1, 1))) return rc;                     // max over the rows of every plan
                    const double cand = p.maxNlinPhaseRot / res_h[2];
                    hz_ = (p.Lspan - z >= cand) ? cand : p.Lspan - z;
                } else if (p.Lspan - z < p.hz) {
                    hz_ = p.Lspan - z;
                } else {
                    hz_ = p.hz;
                }
                ensure_lin(d, hz_ / 2, 1.0 / (double)N, d.lin_a, d.lin_b);
                if ((rc = lin_step(E, Ehd))) return rc;                                  // channels.py:409-410
                int iters = 0;
                for (int it = 0; it < p.maxIter; ++it) {
                    k_rot<T><<<gp, kBlock, 0, pl->stream>>>(Ehd, P, Ec, X, K, N, (T)d.c8g, (T)(sgn * hz_));
                    if ((rc = lin_step(X, X))) return rc;                                // channels.py:420-421
                    k_conv<T><<<gc, kBlock, 0, pl->stream>>>(X, Ec, total, pnum, pden);
                    k_finish<<<1, kBlock, 0, pl->stream>>>(pnum, pden, nullptr, gc, res);
                    if ((rc = read_results())) return rc;
                    if ((rc = coupled(res_h, 2, 0))) return rc;                         // norms over the rows of every plan
                    const double lim = std::sqrt(res_h[0]) / std::sqrt(res_h[1]);        // channels.py:517-519
                    lims[(size_t)it] = lim;
                    std::swap(Ec, X);                                                    // E_conv = E_fd
                    iters = it + 1;
                    if (lim < p.tol) break;
                    if (it == p.maxIter - 1) st->nonconverged_steps++;
                }
                E = Ec;
                z += hz_;
                st->steps++;
                st->iterations += iters;
                st->transforms += (int64_t)nrows * (2 + 2 * (int64_t)iters);
                ts.step(hz_, iters, lims.data());
            }
            if (p.direction >= 0 && (rc = amp_fwd(p, d, span, span - s0, noise, std::exp(d.alpha_lin / 2 * p.Lspan)))) return rc;
            if (wants_snapshot(p, span) && (rc = snapshot())) return rc;
        }
        return SSF_OK;
    }

  public:
    int execute(const ssf_params &p, int s0, int s1, const void *noise, ssf_stats *st, ssf_trace *trace) override {
        const Derived d = derive(p);
        TraceSink ts;
        ts.begin(trace, p.maxIter);
        SSF_HIP(pl, hipEventRecord(ev0, pl->stream));
        int rc = p.model == SSF_MODEL_NLSE ? run_nlse(p, d, s0, s1, noise, st) : run_manakov(p, d, s0, s1, noise, st, ts);
        if (rc) return rc;
        SSF_HIP(pl, hipEventRecord(ev1, pl->stream));
        SSF_HIP(pl, hipStreamSynchronize(pl->stream));
        float ms = 0;
        SSF_HIP(pl, hipEventElapsedTime(&ms, ev0, ev1));
        st->device_ms += ms;
        st->n_snapshots = (int32_t)snaps.size() + n_sunk;
        return SSF_OK;
    }

    // Eo = ifft(fft(Ei) * exp(-alpha/2 L + j beta2/2 w^2 L))     (channels.py:97)
    int linear_channel(double Fs, double Fc, double alpha, double D, double L) override {
        ssf_params p{};
        p.Fs = Fs; p.Fc = Fc; p.alpha = alpha; p.D = D; p.direction = 1; p.Lspan = 1; p.NF = 4.5;
        const Derived d = derive(p);
        ensure_lin(d, L, 1.0 / (double)N, d.lin_a, d.lin_b);
        C *other = (E == bufA) ? bufB : bufA;
        int rc = lin_step(E, other);
        if (rc) return rc;
        E = other;
        SSF_HIP(pl, hipStreamSynchronize(pl->stream));
        return SSF_OK;
    }
};

}  // namespace

static Engine *make_general(ssf_plan *plan, bool bluestein) {
    int rc;
    Engine *e;
    if (plan->precision == SSF_C128) {
        auto *x = new RocfftEngine<double>(plan, bluestein);
        rc = x->init();
        e = x;
    } else {
        auto *x = new RocfftEngine<float>(plan, bluestein);
        rc = x->init();
        e = x;
    }
    if (rc != SSF_OK) {
        delete e;
        return nullptr;
    }
    return e;
}
Engine *make_rocfft_engine(ssf_plan *plan) { return make_general(plan, false); }
Engine *make_general_engine(ssf_plan *plan) { return make_general(plan, true); }

bool general_supports(int64_t N, int nrows, int precision) {
    if (N < 2 || nrows < 1) return false;
    int64_t m = 256;
    while (m < 2 * N - 1) m *= 2;
    return fused_supports(m, nrows, precision);
}

}  // namespace ssf
