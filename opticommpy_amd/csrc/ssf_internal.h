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
#include "ssf_copy.h"
#include "ssf_derived.h"
#include "ssf_snapshots.h"

namespace ssf {

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

// comm_rccl.hip: device-buffer all-gather on the caller's stream (no synchronisation)
int comm_allgather_on(ssf_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank, hipStream_t st);
int comm_nranks(const ssf_comm *c);
int comm_device(const ssf_comm *c);
const char *comm_error(const ssf_comm *c);

class Engine {
  public:
    virtual ~Engine() {}
    virtual int upload(const void *field, bool aos) = 0;
    virtual int execute(const ssf_params &p, int span_first, int span_last, const void *noise,
                        ssf_stats *stats, ssf_trace *trace) = 0;
    virtual int download(void *field, int which, bool aos) = 0;      // which: -1 current, >= 0 snapshot
    virtual int n_snapshots() const = 0;
    virtual int linear_channel(double Fs, double Fc, double alpha, double D, double L) = 0;
    virtual int id() const = 0;
    virtual int pipeline() const { return SSF_PIPE_DEVICE; }
    virtual int unit_stats(int, ssf_stats *) { return SSF_ERR_UNSUPPORTED; }
    virtual int set_lanes(int) { return SSF_OK; }
    virtual int set_coupling(ssf_reduce_fn, void *) { return SSF_ERR_UNSUPPORTED; }
    virtual int set_coupling_comm(ssf_comm *) { return SSF_ERR_UNSUPPORTED; }
    virtual int set_profiling(int) { return SSF_ERR_UNSUPPORTED; }
    virtual int kernel_times(ssf_kernel_times *) { return SSF_ERR_UNSUPPORTED; }
};

}  // namespace ssf

struct ssf_plan {
    int device = 0;
    int64_t N = 0;
    int nrows = 0;
    int units = 1;               // rows form `units` independent fields (ssf_plan_set_units)
    int lanes = 1;               // plans sharing this GPU concurrently (ssf_plan_set_lanes)
    bool coupled = false;        // a coupling communicator / reducer is attached (ssf_set_coupling[_comm]): the engine must stay
    int precision = SSF_C128;
    int engine_id = 0;
    hipStream_t stream = nullptr;
    ssf::Engine *engine = nullptr;
    ssf_stats stats{};
    std::string err;
    bool has_field = false;
    ssf::Stager stager;
    ssf::SnapshotSink sink;      // destination of streamed snapshots (ssf_set_snapshot_sink), inactive by default
};

namespace ssf {

// Engine factories (engine_rocfft.hip, engine_fused.hip).  Return nullptr and fill
// plan->err on failure.
Engine *make_rocfft_engine(ssf_plan *plan);
Engine *make_fused_engine(ssf_plan *plan);
bool fused_supports(int64_t N, int nrows, int precision);
int fused_couple_reduce_selftest(int nranks, int npart, const double *parts, double *out5, std::string *err);
int fused_overlap_save(int device, int64_t sigLen, int nrows, int precision, int log2nfft, int K, const void *Hfft,
                        const void *in, void *out, std::string *err);

// Circular convolution of every row of an (nrows, M) block with one of two fixed kernels, M = 2^m, on the fused kernels
// (column FWD, row FFT . multiplier-array . IFFT, column INV: three launches on the plan's stream).  The general-length
// engine builds its length-N transforms from it (Bluestein): any N the reference accepts runs on the hand-written kernels.
class FusedConv {
  public:
    virtual ~FusedConv() {}
    virtual void *work() = 0;                                        // the (nrows, M) block, convolved in place
    virtual int set_kernel(int which, const void *b_host) = 0;       // M complex values (already scaled by 1 / M)
    virtual int run(int which) = 0;
    virtual std::string error() const = 0;
};
FusedConv *make_fused_conv(ssf_plan *plan, int64_t M, int nrows);

// FFT . H . IFFT of every row of an (nrows, N) block in ONE launch, N = 2^a 3^b 5^c small enough for a row to live in LDS
// (mixed_fft.h, at most 8192 values): the linear step of the general-length engine at the short notebook lengths
// (1500, 3000, 6000 ...), which the column x row split does not take (fewer than four factors of two) and a Bluestein
// convolution serves with three launches of two to four times the length.
class FusedRows {
  public:
    virtual ~FusedRows() {}
    // out = ifft(fft(in) * exp((lin_a + j lin_b w^2) hzh)) * scale * N, w = w_scale * fftfreq(N) (in may be out)
    virtual int lin(const void *in, void *out, double hzh, double lin_a, double lin_b, double w_scale, double scale) = 0;
    virtual std::string error() const = 0;
};
FusedRows *make_fused_rows(ssf_plan *plan, int64_t N, int nrows);      // nullptr: this length is not served
bool fused_rows_supports(int64_t N);
// general-length engine with its transforms on the fused kernels (Bluestein) instead of rocFFT; nullptr if N is out of range
Engine *make_general_engine(ssf_plan *plan);
bool general_supports(int64_t N, int nrows, int precision);

// receiver front-end (engine_rx.hip)
int rx_run(int device, int mode, int64_t N, int nmodes, const ssf_rx_params *p, const void *in0, const void *lo,
           const double *un, void *out, std::string *err);
int rx_fir(int device, int64_t sigLen, int ncols, int ntaps, const void *taps, const void *in, void *out, std::string *err);
int rx_axpy(int device, int64_t n, double alpha, const void *x, void *y, std::string *err);
int rx_fir_long(int device, int64_t inLen, int64_t outLen, int ncols, int64_t ntaps, const void *taps, int64_t shift, const void *in,
                void *out, std::string *err);
int rx_overlap_save(int device, int64_t sigLen, int ncols, int nfft, int K, const void *Hfft, const void *in, void *out,
                    std::string *err);
int rx_delay(int device, int64_t N, double delay, double Fs, const void *in, void *out, std::string *err);
int rx_chain(int device, int64_t N, const ssf_rx_params *p, const void *Es, const void *Elo, const void *taps, int ntaps, int SpSin,
             int decFactor, const void *edcH, int edcK, int edc_nfft, void *out, int32_t *sampDelay, std::string *err);
enum { kOptEdfa = 0, kOptPbs = 1, kOptHybrid = 2 };      // (= rx_kernels.h: OPT_EDFA / OPT_PBS / OPT_HYBRID)
int rx_optics(int device, int op, int64_t n, int ncols, double p0, double p1, unsigned long long seed, unsigned row0, const void *a,
              const void *b, void *o0, void *o1, std::string *err);
int mk_nlin_phase(int device, int64_t n, double gamma, const void *Ex, const void *Ey, const double *Pch, double *phi,
                  std::string *err);
int mk_convergence(int device, int64_t n, const void *xfd, const void *yfd, const void *xc, const void *yc, double *lim,
                   std::string *err);
int tx_wdm(int device, const ssf_tx_params *p, const void *symbols, const double *taps, const double *phi, const double *amp,
           const double *deltaF, void *out, double *power_out, std::string *err);
int rx_decimate(int device, int64_t N, int ncols, int SpSin, int decFactor, const void *in, void *out, int32_t *sampDelay,
                std::string *err);

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
