// ssf_internal.h -- shared host-side declarations for libssf_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ssf.h"

namespace ssf {

// scipy.constants literals used by the reference (channels.py:187, devices.py:721)
constexpr double kC = 299792458.0;
constexpr double kH = 6.62607015e-34;
constexpr double kPi = 3.14159265358979323846;

// Constants the reference derives at the top of each call
// (channels.py:187-199 / 344-356, equalization.py:1063-1077, devices.py:712-722).
struct Derived {
    double alpha_lin;   // alpha / (10 log10 e)            [1/km]
    double beta2;       // -(D lambda^2) / (2 pi c_kms)
    double w_scale;     // 2 pi Fs ;  omega_k = w_scale * (k_signed / N)
    double c8g;         // (8/9) gamma
    double lin_a;       // real part of argLimOp: -alpha/2 (fwd) or +alpha/2 (DBP)
    double lin_b;       // imag coefficient of argLimOp: +beta2/2 (fwd) or -beta2/2 (DBP)
    double G_lin;       // EDFA linear gain, G = alpha * Lspan dB
    double p_noise;     // ASE power in Fs
};

inline Derived derive(const ssf_params &p) {
    Derived d;
    const double c_kms = kC / 1e3;
    const double lam = c_kms / p.Fc;
    d.alpha_lin = p.alpha / (10.0 * std::log10(std::exp(1.0)));
    d.beta2 = -(p.D * lam * lam) / (2.0 * kPi * c_kms);
    d.w_scale = 2.0 * kPi * p.Fs;
    d.c8g = (8.0 / 9.0) * p.gamma;
    const double s = p.direction >= 0 ? 1.0 : -1.0;
    d.lin_a = -s * (d.alpha_lin / 2.0);
    d.lin_b = s * (d.beta2 / 2.0);
    const double G = p.alpha * p.Lspan;
    const double NF_lin = std::pow(10.0, p.NF / 10.0);
    d.G_lin = std::pow(10.0, G / 10.0);
    const double nsp = (d.G_lin * NF_lin - 1.0) / (2.0 * (d.G_lin - 1.0));
    d.p_noise = (d.G_lin - 1.0) * nsp * kH * p.Fc * p.Fs;
    return d;
}

// Host-side trace accumulation shared by the engines.
struct TraceSink {
    ssf_trace *t = nullptr;
    int maxIter = 1;
    void begin(ssf_trace *tr, int mi) {
        t = tr;
        maxIter = mi > 0 ? mi : 1;
    }
    void step(double hz, int iters, const double *lims) {
        if (!t) return;
        if (t->count < t->capacity) {
            const int64_t i = t->count;
            if (t->hz) t->hz[i] = hz;
            if (t->iters) t->iters[i] = iters;
            if (t->lims)
                for (int k = 0; k < maxIter; ++k) t->lims[i * maxIter + k] = k < iters && lims ? lims[k] : NAN;
        }
        t->count++;
    }
};

class Engine {
  public:
    virtual ~Engine() {}
    virtual int upload(const void *soa) = 0;
    virtual int execute(const ssf_params &p, int span_first, int span_last, const void *noise,
                        ssf_stats *stats, ssf_trace *trace) = 0;
    virtual int download(void *soa) = 0;
    virtual int download_snapshots(void *soa) = 0;
    virtual int linear_channel(double Fs, double Fc, double alpha, double D, double L) = 0;
    virtual int id() const = 0;
};

}  // namespace ssf

struct ssf_plan {
    int device = 0;
    int64_t N = 0;
    int nrows = 0;
    int precision = SSF_C128;
    int engine_id = 0;
    hipStream_t stream = nullptr;
    ssf::Engine *engine = nullptr;
    ssf_stats stats{};
    std::string err;
    bool has_field = false;
};

namespace ssf {

// Engine factories (engine_rocfft.hip, engine_fused.hip).  Return nullptr and fill
// plan->err on failure.
Engine *make_rocfft_engine(ssf_plan *plan);
Engine *make_fused_engine(ssf_plan *plan);
bool fused_supports(int64_t N, int nrows, int precision);

inline int fail(ssf_plan *p, int code, const std::string &msg) {
    if (p) p->err = msg;
    return code;
}

#define SSF_HIP(plan, call)                                                                  \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess)                                                               \
            return ssf::fail((plan), e__ == hipErrorOutOfMemory ? SSF_ERR_OOM : SSF_ERR_HIP, \
                             std::string(#call) + ": " + hipGetErrorString(e__));            \
    } while (0)

}  // namespace ssf
