// ssf_api.hip -- the extern "C" boundary declared in include/ssf.h.
#include <map>
#include <memory>
#include <thread>

#include "ssf_internal.h"

using namespace ssf;

namespace {
thread_local std::string g_err;   // errors raised without a plan (plan creation, discovery)

int set_err(int code, const std::string &m) {
    g_err = m;
    return code;
}

int check_params(ssf_plan *pl, const ssf_params *p, int s0, int s1) {
    if (!p) return fail(pl, SSF_ERR_BAD_ARG, "params is NULL");
    if (p->model != SSF_MODEL_NLSE && p->model != SSF_MODEL_MANAKOV) return fail(pl, SSF_ERR_BAD_ARG, "bad model");
    if (p->model == SSF_MODEL_MANAKOV && (pl->nrows % 2)) return fail(pl, SSF_ERR_BAD_ARG, "Manakov model needs an even number of rows");
    if (p->model == SSF_MODEL_NLSE && p->direction < 0) return fail(pl, SSF_ERR_BAD_ARG, "NLSE model has no back-propagation mode");
    if (!(p->Fs > 0) || !(p->Fc > 0)) return fail(pl, SSF_ERR_BAD_ARG, "Fs and Fc must be positive");
    if (!(p->Lspan > 0)) return fail(pl, SSF_ERR_BAD_ARG, "Lspan must be positive");
    if (!(p->hz > 0) && !(p->model == SSF_MODEL_MANAKOV && p->nlprMethod)) return fail(pl, SSF_ERR_BAD_ARG, "hz must be positive");
    if (p->model == SSF_MODEL_MANAKOV && p->maxIter < 1) return fail(pl, SSF_ERR_BAD_ARG, "maxIter must be >= 1");
    if (p->Nspans < 0 || s0 < 1 || s1 > p->Nspans) return fail(pl, SSF_ERR_BAD_ARG, "span range outside [1, Nspans]");
    if (p->amp < SSF_AMP_NONE || p->amp > SSF_AMP_EDFA) return fail(pl, SSF_ERR_BAD_ARG, "bad amp");
    if (p->amp == SSF_AMP_EDFA && p->direction >= 0) {
        if (!(p->alpha * p->Lspan > 0)) return fail(pl, SSF_ERR_BAD_ARG, "EDFA gain should be a positive scalar");   // devices.py:709
        if (!(p->NF >= 3)) return fail(pl, SSF_ERR_BAD_ARG, "The minimal EDFA noise figure is 3 dB");                 // devices.py:710
    }
    if (p->n_save < 0 || (p->n_save > 0 && !p->save_spans)) return fail(pl, SSF_ERR_BAD_ARG, "bad save_spans");
    return SSF_OK;
}
}  // namespace

extern "C" {

const char *ssf_version(void) { return "opticommpy_amd-ssf 0.1 (gfx950)"; }

int ssf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) return set_err(SSF_ERR_HIP, hipGetErrorString(e));
    return n;
}

int ssf_device_info(int device, ssf_device_info_t *out) {
    if (!out) return set_err(SSF_ERR_BAD_ARG, "out is NULL");
    hipDeviceProp_t pr;
    hipError_t e = hipGetDeviceProperties(&pr, device);
    if (e != hipSuccess) return set_err(SSF_ERR_NO_DEVICE, hipGetErrorString(e));
    std::memset(out, 0, sizeof(*out));
    std::snprintf(out->name, sizeof(out->name), "%s", pr.name);
    std::snprintf(out->arch, sizeof(out->arch), "%s", pr.gcnArchName);
    out->compute_units = pr.multiProcessorCount;
    out->total_mem_bytes = (int64_t)pr.totalGlobalMem;
    out->lds_per_block_bytes = (int64_t)pr.sharedMemPerBlock;
    return SSF_OK;
}

const char *ssf_last_error(const ssf_plan *plan) { return plan ? plan->err.c_str() : g_err.c_str(); }

static bool smooth13(int64_t n) {
    for (int q : {2, 3, 5, 7, 11, 13})
        while (n % q == 0) n /= q;
    return n == 1;
}

int ssf_plan_create(int device, int64_t N, int32_t nrows, int32_t precision, int32_t engine, ssf_plan **out) {
    if (!out) return set_err(SSF_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (N < 2 || nrows < 1) return set_err(SSF_ERR_BAD_ARG, "N must be >= 2 and nrows >= 1");
    if (precision != SSF_C64 && precision != SSF_C128) return set_err(SSF_ERR_BAD_ARG, "bad precision");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_err(SSF_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_err(SSF_ERR_NO_DEVICE, "device index out of range");
    if (hipSetDevice(device) != hipSuccess) return set_err(SSF_ERR_HIP, "hipSetDevice failed");
    auto *pl = new ssf_plan();
    pl->device = device;
    pl->N = N;
    pl->nrows = nrows;
    pl->precision = precision;
    if (hipStreamCreateWithFlags(&pl->stream, hipStreamNonBlocking) != hipSuccess) {
        delete pl;
        return set_err(SSF_ERR_HIP, "hipStreamCreate failed");
    }
    (void)pl->stager.init();          // falls back to plain copies if pinned memory is unavailable
    // SSF_ENGINE_FUSED / AUTO: the fused pipeline when it takes N natively, otherwise the general-length engine with its
    // transforms built from the fused kernels (Bluestein); rocFFT only on request (the on-GPU cross-check) or, under AUTO,
    // for lengths beyond the Bluestein range (2N - 1 > 2^22 complex128 / 2^23 complex64)
    const bool native = fused_supports(N, nrows, precision), general = !native && general_supports(N, nrows, precision);
    int want = engine;
    if (want == SSF_ENGINE_AUTO) {
        want = (native || general) ? SSF_ENGINE_FUSED : SSF_ENGINE_ROCFFT;
        // AUTO is never the slower engine: lengths the hand-written kernels only reach through a Bluestein convolution, but
        // whose prime factors are all <= 13 (30 030 = 2 3 5 7 11 13), are one or two native kernels for rocFFT (measured,
        // tools/bench_lengths.py: 30 030: 4 653 against 3 446 steps/s); primes and other lengths stay on the fused kernels
        // (10 007: 3 753 against 2 950; rocFFT itself falls back to Bluestein there).  SSF_ENGINE_FUSED still forces them.
        // The one-launch LDS rows (2^a 3^b 5^c <= 8192) win from about 4000 samples on (6000: 7 491 against 4 918 steps/s); below,
        // a single workgroup per row is one long latency chain and rocFFT's small kernels are faster (1500: 6 040 against
        // 7 427, 3000: 5 477 against 6 393; gpurun_out/r3e/lengths.txt).
        const bool rows_win = fused_rows_supports(N) && N >= 4000;
        if (!native && general && !rows_win && smooth13(N)) want = SSF_ENGINE_ROCFFT;
    }
    if (want == SSF_ENGINE_FUSED && !native && !general) {
        (void)hipStreamDestroy(pl->stream);
        delete pl;
        return set_err(SSF_ERR_UNSUPPORTED, "fused engine: N beyond the range of its transforms (use SSF_ENGINE_ROCFFT)");
    }
    pl->engine = want == SSF_ENGINE_ROCFFT ? make_rocfft_engine(pl) : native ? make_fused_engine(pl) : make_general_engine(pl);
    if (!pl->engine) {
        const int code = pl->err.find("out of memory") != std::string::npos ? SSF_ERR_OOM : SSF_ERR_HIP;
        set_err(code, pl->err.empty() ? "engine creation failed" : pl->err);
        (void)hipStreamDestroy(pl->stream);
        delete pl;
        return code;
    }
    pl->engine_id = pl->engine->id();
    *out = pl;
    return SSF_OK;
}

int ssf_plan_set_lanes(ssf_plan *plan, int32_t n_lanes) {
    if (!plan || !plan->engine) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    if (n_lanes < 1) return fail(plan, SSF_ERR_BAD_ARG, "ssf_plan_set_lanes: n_lanes must be >= 1");
    plan->lanes = n_lanes;
    return plan->engine->set_lanes(n_lanes);
}

int ssf_plan_pipeline(const ssf_plan *plan) {
    if (!plan || !plan->engine) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    return plan->engine->pipeline();
}

// ssf_plan_set_units rebuilds the engine; if that fails AND the rebuild with the old unit count fails too (out of memory twice) the
// plan is left without one: every entry point that needs it says so instead of dereferencing a null pointer
#define SSF_NEED_ENGINE(plan)                                                                                        \
    do {                                                                                                             \
        if (!(plan)->engine) return fail((plan), SSF_ERR_STATE, "the plan has no engine (ssf_plan_set_units failed)"); \
    } while (0)

int ssf_plan_set_units(ssf_plan *plan, int32_t n_units) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    if (n_units < 1 || plan->nrows % n_units) return fail(plan, SSF_ERR_BAD_ARG, "ssf_plan_set_units: nrows must be a multiple of n_units");
    if (n_units == plan->units) return SSF_OK;
    // (rebuilding the engine would silently drop an attached coupling communicator / reducer: the caller would go on believing
    //  the plan is coupled -- and a plan of independent units is not a coupled batch anyway)
    if (plan->coupled) return fail(plan, SSF_ERR_STATE, "ssf_plan_set_units: detach the coupling communicator / reducer first (ssf_set_coupling[_comm](plan, NULL))");
    if (plan->engine_id != SSF_ENGINE_FUSED || !fused_supports(plan->N, plan->nrows, plan->precision))
        return fail(plan, SSF_ERR_UNSUPPORTED, "independent units need the natively split fused engine");
    if (n_units > 65535) return fail(plan, SSF_ERR_BAD_ARG, "ssf_plan_set_units: at most 65535 units");
    SSF_HIP(plan, hipSetDevice(plan->device));
    (void)plan->sink.sync();
    delete plan->engine;
    plan->engine = nullptr;
    const int old = plan->units;
    plan->units = n_units;
    plan->has_field = false;
    plan->engine = make_fused_engine(plan);
    if (plan->engine) (void)plan->engine->set_lanes(plan->lanes);
    if (!plan->engine) {                                    // (out of memory?) back to what worked
        const std::string why = plan->err;
        plan->units = old;
        plan->engine = make_fused_engine(plan);
        return fail(plan, why.find("out of memory") != std::string::npos ? SSF_ERR_OOM : SSF_ERR_HIP,
                    why.empty() ? "engine creation failed" : why);
    }
    return SSF_OK;
}

int ssf_get_unit_stats(ssf_plan *plan, int32_t unit, ssf_stats *out) {
    if (!plan || !out) return set_err(SSF_ERR_BAD_ARG, "ssf_get_unit_stats: NULL argument");
    ssf_stats u{};
    SSF_NEED_ENGINE(plan);
    int rc = plan->engine->unit_stats(unit, &u);
    if (rc) return fail(plan, rc, "ssf_get_unit_stats: no such unit (or not a plan of independent units)");
    *out = plan->stats;
    out->steps = u.steps;
    out->iterations = u.iterations;
    out->transforms = u.transforms;
    out->nonconverged_steps = u.nonconverged_steps;
    out->decided_ahead = u.decided_ahead;
    out->rebuilt_iterates = u.rebuilt_iterates;
    out->recovered_fields = u.recovered_fields;
    out->bytes_algorithmic = (double)u.transforms * 2.0 * (plan->precision == SSF_C128 ? 16.0 : 8.0) * (double)plan->N;
    return SSF_OK;
}

int ssf_plan_destroy(ssf_plan *plan) {
    if (!plan) return SSF_OK;
    (void)hipSetDevice(plan->device);
    (void)plan->sink.sync();
    delete plan->engine;
    if (plan->stream) (void)hipStreamDestroy(plan->stream);
    delete plan;
    return SSF_OK;
}

static int upload_common(ssf_plan *plan, const void *field, bool aos) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    if (!field) return fail(plan, SSF_ERR_BAD_ARG, "field is NULL");
    SSF_HIP(plan, hipSetDevice(plan->device));
    SSF_NEED_ENGINE(plan);
    int rc = plan->engine->upload(field, aos);
    if (rc) return rc;
    plan->stats = ssf_stats{};
    plan->stats.engine = plan->engine_id;
    plan->has_field = true;
    return SSF_OK;
}
int ssf_upload(ssf_plan *plan, const void *field_soa) { return upload_common(plan, field_soa, false); }
int ssf_upload_aos(ssf_plan *plan, const void *field_aos) { return upload_common(plan, field_aos, true); }

int ssf_execute(ssf_plan *plan, const ssf_params *params, int32_t span_first, int32_t span_last, const void *noise,
                ssf_stats *stats, ssf_trace *trace) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    if (!plan->has_field) return fail(plan, SSF_ERR_STATE, "ssf_execute before ssf_upload");
    int rc = check_params(plan, params, span_first, span_last);
    if (!rc && plan->sink.active()) {          // every capture of this call must fit the caller's (N, ld) array
        long long ncap = 0;
        for (int i = 0; i < params->n_save; ++i)
            if (params->save_spans[i] >= span_first && params->save_spans[i] <= span_last) ++ncap;
        if (((long long)plan->sink.count() + ncap) * plan->nrows > plan->sink.leading())
            return fail(plan, SSF_ERR_BAD_ARG, "snapshot sink: the captures of this call do not fit the destination's columns");
    }
    if (rc) return rc;
    SSF_HIP(plan, hipSetDevice(plan->device));
    if (trace) trace->count = 0;
    if (span_first <= span_last) {
        SSF_NEED_ENGINE(plan);
        rc = plan->engine->execute(*params, span_first, span_last, noise, &plan->stats, trace);
        if (rc) return rc;
    }
    const double s = plan->precision == SSF_C128 ? 16.0 : 8.0;
    plan->stats.bytes_algorithmic = (double)plan->stats.transforms * 2.0 * s * (double)plan->N;
    if (stats) *stats = plan->stats;
    return SSF_OK;
}

static int download_common(ssf_plan *plan, void *dst, int which, bool aos) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    if (!dst) return fail(plan, SSF_ERR_BAD_ARG, "destination is NULL");
    if (!plan->has_field) return fail(plan, SSF_ERR_STATE, "download before ssf_upload");
    SSF_NEED_ENGINE(plan);
    if (which < -1 || which >= plan->engine->n_snapshots()) return fail(plan, SSF_ERR_BAD_ARG, "no such snapshot");
    SSF_HIP(plan, hipSetDevice(plan->device));
    return plan->engine->download(dst, which, aos);
}
int ssf_download(ssf_plan *plan, void *field_soa) { return download_common(plan, field_soa, -1, false); }
int ssf_download_aos(ssf_plan *plan, int32_t which, void *field_aos) { return download_common(plan, field_aos, which, true); }

int ssf_download_snapshots(ssf_plan *plan, void *snap_soa) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    if (!snap_soa) return fail(plan, SSF_ERR_BAD_ARG, "snapshot buffer is NULL");
    const size_t fb = (size_t)plan->N * plan->nrows * (plan->precision == SSF_C128 ? 16 : 8);
    SSF_NEED_ENGINE(plan);
    for (int i = 0; i < plan->engine->n_snapshots(); ++i) {
        int rc = download_common(plan, (char *)snap_soa + (size_t)i * fb, i, false);
        if (rc) return rc;
    }
    return SSF_OK;
}

int ssf_set_snapshot_sink(ssf_plan *plan, void *dst, int64_t ld, int32_t first_index) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    if (dst && (ld < plan->nrows || first_index < 0)) return fail(plan, SSF_ERR_BAD_ARG, "ssf_set_snapshot_sink: ld < nrows or negative index");
    SSF_HIP(plan, hipSetDevice(plan->device));
    SSF_HIP(plan, plan->sink.set(plan->device, dst, ld, first_index, plan->precision == SSF_C128 ? 16 : 8));
    return SSF_OK;
}

int ssf_sync_snapshots(ssf_plan *plan) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    SSF_HIP(plan, hipSetDevice(plan->device));
    SSF_HIP(plan, hipStreamSynchronize(plan->stream));        // device destinations: ordered on the plan's stream
    SSF_HIP(plan, plan->sink.sync());
    return SSF_OK;
}

int ssf_run(ssf_plan *plan, const ssf_params *params, const void *in, void *out, void *snaps, const void *noise,
            ssf_stats *stats, ssf_trace *trace) {
    int rc = ssf_upload(plan, in);
    if (rc) return rc;
    rc = ssf_execute(plan, params, 1, params ? params->Nspans : 0, noise, stats, trace);
    if (rc) return rc;
    if (out && (rc = ssf_download(plan, out))) return rc;
    if (snaps && plan->stats.n_snapshots > 0 && (rc = ssf_download_snapshots(plan, snaps))) return rc;
    return SSF_OK;
}

int ssf_set_coupling(ssf_plan *plan, ssf_reduce_fn reduce, void *ctx) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    SSF_NEED_ENGINE(plan);
    int rc = plan->engine->set_coupling(reduce, ctx);
    if (rc == SSF_OK) plan->coupled = reduce != nullptr;
    return rc ? fail(plan, rc, "coupled batches need the general-length engine (create the plan with SSF_ENGINE_ROCFFT)") : SSF_OK;
}

int ssf_set_coupling_comm(ssf_plan *plan, ssf_comm *comm) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    SSF_NEED_ENGINE(plan);
    if (comm && ssf::comm_device(comm) != plan->device) return fail(plan, SSF_ERR_BAD_ARG, "ssf_set_coupling_comm: the communicator lives on another device");
    if (comm && plan->units > 1) return fail(plan, SSF_ERR_UNSUPPORTED, "ssf_set_coupling_comm: a plan of independent units is not a coupled batch");
    int rc = plan->engine->set_coupling_comm(comm);
    if (rc == SSF_OK) plan->coupled = comm != nullptr;
    return rc ? fail(plan, rc, "device-side coupling needs the device-resident fused pipeline (else: ssf_set_coupling on SSF_ENGINE_ROCFFT)") : SSF_OK;
}

static int rx_check_device(int device);
int ssf_couple_reduce_selftest(int device, int32_t nranks, int32_t npart, const double *parts, double *out5) {
    if (int rc = rx_check_device(device)) return rc;
    if (hipSetDevice(device) != hipSuccess) return set_err(SSF_ERR_HIP, "hipSetDevice");
    std::string err;
    int rc = ssf::fused_couple_reduce_selftest(nranks, npart, parts, out5, &err);
    return rc ? set_err(rc, err.empty() ? "ssf_couple_reduce_selftest: bad argument" : err) : SSF_OK;
}

int ssf_set_profiling(ssf_plan *plan, int32_t enable) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    SSF_NEED_ENGINE(plan);
    int rc = plan->engine->set_profiling(enable);
    return rc ? fail(plan, rc, "per-kernel profiling is only available on the fused engine") : SSF_OK;
}

int ssf_get_kernel_times(ssf_plan *plan, ssf_kernel_times *out) {
    if (!plan) return set_err(SSF_ERR_BAD_ARG, "plan is NULL");
    if (!out) return fail(plan, SSF_ERR_BAD_ARG, "out is NULL");
    SSF_NEED_ENGINE(plan);
    int rc = plan->engine->kernel_times(out);
    return rc ? fail(plan, rc, "per-kernel profiling is only available on the fused engine") : SSF_OK;
}

namespace {
// memory shape of the fused row stage, no arithmetic (see include/ssf.h)
__global__ void __launch_bounds__(256) k_burst_copy(const double2 *__restrict__ a, double2 *__restrict__ b) {
    const double2 *g = a + (size_t)blockIdx.x * 4096;
    double2 *o = b + (size_t)blockIdx.x * 4096;
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = g[threadIdx.x + 256 * q];
#pragma unroll
    for (int q = 0; q < 16; ++q) o[threadIdx.x + 256 * q] = v[q];
}
}  // namespace

int ssf_device_copy_bandwidth(int device, int64_t bytes, int32_t launches, double *gbs) {
    if (!gbs || bytes < 65536 || bytes % 65536 || launches < 1) return set_err(SSF_ERR_BAD_ARG, "ssf_device_copy_bandwidth: bytes must be a positive multiple of 64 KiB");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return set_err(SSF_ERR_BAD_ARG, "no such device");
    if (hipSetDevice(device) != hipSuccess) return set_err(SSF_ERR_HIP, "hipSetDevice failed");
    double2 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SSF_OK;
    float ms = 0;
    const unsigned nwg = (unsigned)(bytes / 65536);
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) rc = set_err(SSF_ERR_OOM, "hipMalloc failed");
    if (!rc && (hipMemset(a, 0, bytes) != hipSuccess || hipMemset(b, 0, bytes) != hipSuccess)) rc = set_err(SSF_ERR_HIP, "hipMemset failed");
    if (!rc && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = set_err(SSF_ERR_HIP, "hipEventCreate failed");
    if (!rc) {
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_burst_copy, dim3(nwg), dim3(256), 0, 0, i & 1 ? b : a, i & 1 ? a : b);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_burst_copy, dim3(nwg), dim3(256), 0, 0, i & 1 ? b : a, i & 1 ? a : b);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess)
            rc = set_err(SSF_ERR_HIP, "copy probe failed");
        else *gbs = 2.0 * (double)bytes * launches / ((double)ms * 1e-3) / 1e9;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}

int ssf_linear_channel(ssf_plan *plan, double Fs, double Fc, double alpha, double D, double L, const void *in,
                       void *out) {
    int rc = in ? ssf_upload(plan, in) : SSF_OK;                    // NULL: the field the plan holds (ssf_upload_aos)
    if (rc) return rc;
    SSF_NEED_ENGINE(plan);
    if ((rc = plan->engine->linear_channel(Fs, Fc, alpha, D, L))) return rc;
    return out ? ssf_download(plan, out) : SSF_OK;                  // NULL: left in the plan (ssf_download_aos)
}

int ssf_overlap_save(int device, int64_t sigLen, int32_t nrows, int32_t precision, int32_t nfft, int32_t K,
                     const void *Hfft, const void *sig_in, void *sig_out) {
    if (sigLen < 1 || nrows < 1 || !Hfft || !sig_in || !sig_out) return set_err(SSF_ERR_BAD_ARG, "ssf_overlap_save: bad argument");
    if (precision != SSF_C64 && precision != SSF_C128) return set_err(SSF_ERR_BAD_ARG, "bad precision");
    int lg = 0;
    while ((1 << lg) < nfft) ++lg;
    const int lgmax = precision == SSF_C128 ? 13 : 14;
    if ((1 << lg) != nfft || lg < 4 || lg > lgmax)
        return set_err(SSF_ERR_UNSUPPORTED, "ssf_overlap_save: nfft must be a power of two in [16, 8192 (c128) / 16384 (c64)]");
    if (K < 1 || K > nfft) return set_err(SSF_ERR_BAD_ARG, "FFT size is smaller than filter length");   // core.py:1012
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_err(SSF_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_err(SSF_ERR_NO_DEVICE, "device index out of range");
    std::string err;
    // complex128 goes through the receiver pipeline's pooled backend (stream, pinned staging and device
    // blocks are kept between calls); complex64 through the one-shot path of the fused engine
    int rc = precision == SSF_C128 ? ssf::rx_overlap_save(device, sigLen, nrows, nfft, K, Hfft, sig_in, sig_out, &err)
                                   : fused_overlap_save(device, sigLen, nrows, precision, lg, K, Hfft, sig_in, sig_out, &err);
    if (rc) set_err(rc, err);
    return rc;
}

int ssf_mgpu_run(int32_t n_dev, const int32_t *dev_ids, int32_t n_units, int64_t N, int32_t rows_per_unit,
                 int32_t precision, int32_t engine, const ssf_params *params, const void *fields_in,
                 void *fields_out, ssf_stats *stats) {
    if (n_dev < 1 || !dev_ids || n_units < 1 || !params || !fields_in || !fields_out)
        return set_err(SSF_ERR_BAD_ARG, "ssf_mgpu_run: bad argument");
    const size_t unit_bytes = (size_t)rows_per_unit * (size_t)N * (precision == SSF_C128 ? 16 : 8);
    // Two lanes (plan + stream + host thread) per device when it has more than one unit: one unit's transfers overlap
    // the other's kernels, and the kernels of two independent fields fill each other's load / store phases
    // (measured +12-15 % field-steps/s).  Units are independent: the results do not depend on it.
    int lanes = 2;
    if (const char *e = std::getenv("SSF_MGPU_LANES")) lanes = std::max(1, std::min(4, std::atoi(e)));
    // Small units (a launch is one ~10 us latency chain whatever its size): all the units of a device go through ONE plan as
    // independent units -- every launch carries all of them, own control block and decisions per unit, results bit-equal to
    // one plan per unit (ssf_plan_set_units).  SSF_MGPU_BATCH=0 turns it off.
    bool batch = N <= (1ll << 18) && n_units > n_dev && params->model == SSF_MODEL_MANAKOV && (rows_per_unit % 2) == 0 &&
                 engine != SSF_ENGINE_ROCFFT && fused_supports(N, rows_per_unit, precision);
    if (const char *e = std::getenv("SSF_MGPU_BATCH")) batch = batch && std::atoi(e) != 0;
    if (batch) {
        std::vector<int> brc((size_t)n_dev, SSF_OK);
        std::vector<std::string> berr((size_t)n_dev);
        std::vector<std::thread> bt;
        for (int d = 0; d < n_dev; ++d) {
            const int u0 = (int)((int64_t)n_units * d / n_dev), u1 = (int)((int64_t)n_units * (d + 1) / n_dev);
            bt.emplace_back([=, &brc, &berr] {
                constexpr int kMaxBatch = 64;
                for (int b0 = u0; b0 < u1 && brc[(size_t)d] == SSF_OK; b0 += kMaxBatch) {
                    const int nb = std::min(kMaxBatch, u1 - b0);
                    ssf_plan *pl = nullptr;
                    int rc = ssf_plan_create(dev_ids[d], N, rows_per_unit * nb, precision, engine, &pl);
                    if (!rc && nb > 1) rc = ssf_plan_set_units(pl, nb);
                    ssf_params p = *params;
                    p.rng_row_offset = params->rng_row_offset + b0 * rows_per_unit;     // unit u draws rows u * rows_per_unit ...
                    ssf_stats st{};
                    if (!rc) rc = ssf_run(pl, &p, (const char *)fields_in + (size_t)b0 * unit_bytes,
                                          (char *)fields_out + (size_t)b0 * unit_bytes, nullptr, nullptr, &st, nullptr);
                    for (int u = 0; u < nb && !rc && stats; ++u) {
                        if (nb > 1) rc = ssf_get_unit_stats(pl, u, &stats[b0 + u]);
                        else stats[b0 + u] = st;
                    }
                    if (rc) {
                        brc[(size_t)d] = rc;
                        berr[(size_t)d] = ssf_last_error(pl);
                    }
                    if (pl) ssf_plan_destroy(pl);
                }
            });
        }
        for (auto &t : bt) t.join();
        for (int d = 0; d < n_dev; ++d)
            if (brc[(size_t)d]) return set_err(brc[(size_t)d], "device " + std::to_string(dev_ids[d]) + ": " + berr[(size_t)d]);
        return SSF_OK;
    }
    const int nslots = n_dev * lanes;
    std::vector<int> rcs((size_t)nslots, SSF_OK);
    std::vector<std::string> errs((size_t)nslots);
    std::vector<std::thread> th;
    for (int d = 0; d < n_dev; ++d) {
        // contiguous block of units per device: [u0, u1)
        const int u0 = (int)((int64_t)n_units * d / n_dev), u1 = (int)((int64_t)n_units * (d + 1) / n_dev);
        for (int l = 0; l < lanes; ++l) {
            const int slot = d * lanes + l;
            th.emplace_back([=, &rcs, &errs] {
                if (u0 + l >= u1) return;
                ssf_plan *pl = nullptr;
                int rc = ssf_plan_create(dev_ids[d], N, rows_per_unit, precision, engine, &pl);
                if (rc) {
                    rcs[(size_t)slot] = rc;
                    errs[(size_t)slot] = ssf_last_error(nullptr);
                    return;
                }
                if (std::min(lanes, u1 - u0) > 1) (void)ssf_plan_set_lanes(pl, std::min(lanes, u1 - u0));
                for (int u = u0 + l; u < u1 && rc == SSF_OK; u += lanes) {
                    ssf_stats st{};
                    ssf_params pu = *params;                   // a shared seed keys ONE noise stream: unit u draws its own rows of it
                    pu.rng_row_offset = params->rng_row_offset + u * rows_per_unit;
                    rc = ssf_run(pl, &pu, (const char *)fields_in + (size_t)u * unit_bytes,
                                 (char *)fields_out + (size_t)u * unit_bytes, nullptr, nullptr, &st, nullptr);
                    if (stats) stats[u] = st;
                }
                if (rc) {
                    rcs[(size_t)slot] = rc;
                    errs[(size_t)slot] = ssf_last_error(pl);
                }
                ssf_plan_destroy(pl);
            });
        }
    }
    for (auto &t : th) t.join();
    for (int i = 0; i < nslots; ++i)
        if (rcs[(size_t)i]) return set_err(rcs[(size_t)i], "device " + std::to_string(dev_ids[i / lanes]) + ": " + errs[(size_t)i]);
    return SSF_OK;
}

// ---- device-resident arrays ------------------------------------------------------------------
int ssf_device_malloc(int device, int64_t bytes, void **ptr) {
    if (!ptr || bytes < 1) return set_err(SSF_ERR_BAD_ARG, "ssf_device_malloc: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return set_err(SSF_ERR_NO_DEVICE, "device index out of range");
    if (hipSetDevice(device) != hipSuccess) return set_err(SSF_ERR_HIP, "hipSetDevice failed");
    hipError_t e = hipMalloc(ptr, (size_t)bytes);
    if (e != hipSuccess) return set_err(e == hipErrorOutOfMemory ? SSF_ERR_OOM : SSF_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    return SSF_OK;
}

int ssf_device_free(int device, void *ptr) {
    if (!ptr) return SSF_OK;
    if (hipSetDevice(device) != hipSuccess) return set_err(SSF_ERR_HIP, "hipSetDevice failed");
    hipError_t e = hipFree(ptr);
    return e == hipSuccess ? SSF_OK : set_err(SSF_ERR_HIP, std::string("hipFree: ") + hipGetErrorString(e));
}

int ssf_device_memcpy(int device, void *dst, const void *src, int64_t bytes) {
    if (!dst || !src || bytes < 0) return set_err(SSF_ERR_BAD_ARG, "ssf_device_memcpy: bad argument");
    if (hipSetDevice(device) != hipSuccess) return set_err(SSF_ERR_HIP, "hipSetDevice failed");
    const bool dst_dev = ssf::on_device(dst), src_dev = ssf::on_device(src);
    hipError_t e;
    if (dst_dev != src_dev && bytes >= (1 << 20)) {
        // large host <-> device copies of pageable memory go through the pinned double buffer (a plain
        // hipMemcpy of pageable memory runs at a fraction of the link rate)
        // one lane (stream + pinned double buffer + events) per device and host thread: events and streams belong to
        // the device they were created on
        struct Lane {
            hipStream_t st = nullptr;
            ssf::Stager stg;
        };
        thread_local std::map<int, std::unique_ptr<Lane>> lanes;
        std::unique_ptr<Lane> &slot = lanes[device];
        if (!slot) {
            slot.reset(new Lane());
            if (hipStreamCreateWithFlags(&slot->st, hipStreamNonBlocking) != hipSuccess) {
                slot.reset();
                return set_err(SSF_ERR_HIP, "hipStreamCreate failed");
            }
            (void)slot->stg.init();            // (current device = `device`; falls back to plain copies without pinned memory)
        }
        Lane &lane = *slot;
        e = dst_dev ? lane.stg.h2d(dst, src, (size_t)bytes, lane.st) : lane.stg.d2h(dst, src, (size_t)bytes, lane.st);
    } else {
        e = hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDefault);
    }
    return e == hipSuccess ? SSF_OK : set_err(SSF_ERR_HIP, std::string("ssf_device_memcpy: ") + hipGetErrorString(e));
}

// ---- receiver side (engine_rx.hip) -----------------------------------------------------------
static int rx_check_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_err(SSF_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_err(SSF_ERR_NO_DEVICE, "device index out of range");
    return SSF_OK;
}

int ssf_fir_filter(int device, int64_t sigLen, int32_t ncols, int32_t ntaps, const void *taps, const void *in, void *out) {
    if (!taps || !in || !out) return set_err(SSF_ERR_BAD_ARG, "ssf_fir_filter: NULL argument");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::rx_fir(device, sigLen, ncols, ntaps, taps, in, out, &err);
    return rc ? set_err(rc, "ssf_fir_filter: " + err) : SSF_OK;
}

int ssf_fir_long(int device, int64_t inLen, int64_t outLen, int32_t ncols, int64_t ntaps, const void *taps, int64_t shift,
                 const void *in, void *out) {
    if (!taps || !in || !out) return set_err(SSF_ERR_BAD_ARG, "ssf_fir_long: NULL argument");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::rx_fir_long(device, inLen, outLen, ncols, ntaps, taps, shift, in, out, &err);
    return rc ? set_err(rc, "ssf_fir_long: " + err) : SSF_OK;
}

int ssf_device_axpy(int device, int64_t n, double alpha, const double *x, double *y) {
    if (!x || !y) return set_err(SSF_ERR_BAD_ARG, "ssf_device_axpy: NULL argument");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::rx_axpy(device, n, alpha, x, y, &err);
    return rc ? set_err(rc, "ssf_device_axpy: " + err) : SSF_OK;
}

int ssf_delay_signal(int device, int64_t N, double delay, double Fs, const void *in, void *out) {
    if (!in || !out) return set_err(SSF_ERR_BAD_ARG, "ssf_delay_signal: NULL argument");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::rx_delay(device, N, delay, Fs, in, out, &err);
    return rc ? set_err(rc, "ssf_delay_signal: " + err) : SSF_OK;
}

int ssf_edfa(int device, int64_t n, int32_t ncols, double G_lin, double p_noise, int64_t rng_seed, int32_t rng_row_offset,
             const void *field_in, const void *noise, void *field_out) {
    if (!field_in || !field_out || n < 1 || ncols < 1) return set_err(SSF_ERR_BAD_ARG, "ssf_edfa: bad argument");
    if (!(G_lin > 0) || p_noise < 0) return set_err(SSF_ERR_BAD_ARG, "ssf_edfa: the gain must be positive, the noise power non-negative");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    const double sigma = (!noise && rng_seed != 0) ? std::sqrt(p_noise / 2) : 0.0;                  // core.py:739-763
    int rc = ssf::rx_optics(device, ssf::kOptEdfa, n, ncols, std::sqrt(G_lin), sigma, (unsigned long long)rng_seed,
                            (unsigned)rng_row_offset, field_in, noise, field_out, nullptr, &err);
    return rc ? set_err(rc, "ssf_edfa: " + err) : SSF_OK;
}

int ssf_pbs(int device, int64_t N, int32_t ncols, double theta, const void *E, void *Ex, void *Ey) {
    if (!E || !Ex || !Ey || N < 1) return set_err(SSF_ERR_BAD_ARG, "ssf_pbs: bad argument");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::rx_optics(device, ssf::kOptPbs, N, ncols, std::cos(theta), std::sin(theta), 0, 0, E, nullptr, Ex, Ey, &err);
    return rc ? set_err(rc, "ssf_pbs: " + err) : SSF_OK;
}

int ssf_optical_hybrid_2x4(int device, int64_t N, const void *Es, const void *Elo, void *Eo) {
    if (!Es || !Elo || !Eo || N < 1) return set_err(SSF_ERR_BAD_ARG, "ssf_optical_hybrid_2x4: bad argument");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::rx_optics(device, ssf::kOptHybrid, N, 1, 0.0, 0.0, 0, 0, Es, Elo, Eo, nullptr, &err);
    return rc ? set_err(rc, "ssf_optical_hybrid_2x4: " + err) : SSF_OK;
}

int ssf_nlin_phase_rot(int device, int64_t n, double gamma, const void *Ex, const void *Ey, const double *Pch, double *phi) {
    if (!Ex || !Ey || !Pch || !phi) return set_err(SSF_ERR_BAD_ARG, "ssf_nlin_phase_rot: NULL argument");
    std::string err;
    const int rc = mk_nlin_phase(device, n, gamma, Ex, Ey, Pch, phi, &err);
    return rc ? set_err(rc, "ssf_nlin_phase_rot: " + err) : SSF_OK;
}

int ssf_convergence_condition(int device, int64_t n, const void *Ex_fd, const void *Ey_fd, const void *Ex_conv,
                              const void *Ey_conv, double *lim) {
    if (!Ex_fd || !Ey_fd || !Ex_conv || !Ey_conv || !lim) return set_err(SSF_ERR_BAD_ARG, "ssf_convergence_condition: NULL argument");
    std::string err;
    const int rc = mk_convergence(device, n, Ex_fd, Ey_fd, Ex_conv, Ey_conv, lim, &err);
    return rc ? set_err(rc, "ssf_convergence_condition: " + err) : SSF_OK;
}

int ssf_decimate(int device, int64_t N, int32_t ncols, int32_t SpSin, int32_t decFactor, const void *in, void *out,
                 int32_t *sampDelay) {
    if (!in || !out) return set_err(SSF_ERR_BAD_ARG, "ssf_decimate: NULL argument");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::rx_decimate(device, N, ncols, SpSin, decFactor, in, out, sampDelay, &err);
    return rc ? set_err(rc, "ssf_decimate: " + err) : SSF_OK;
}

int ssf_rx_run(int device, int32_t mode, int64_t N, int32_t nmodes, const ssf_rx_params *params, const void *in0,
               const void *lo, const double *unit_normals, void *out) {
    if (!params || !in0 || !out) return set_err(SSF_ERR_BAD_ARG, "ssf_rx_run: NULL argument");
    if (mode < SSF_RX_PHOTODIODE || mode > SSF_RX_IQ_MIXING) return set_err(SSF_ERR_BAD_ARG, "ssf_rx_run: unknown mode");
    if ((mode == SSF_RX_COHERENT || mode == SSF_RX_PDM_COHERENT) && !lo) return set_err(SSF_ERR_BAD_ARG, "ssf_rx_run: the LO field is NULL");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::rx_run(device, mode, N, nmodes, params, in0, lo, unit_normals, out, &err);
    return rc ? set_err(rc, "ssf_rx_run: " + err) : SSF_OK;
}

int ssf_rx_chain(int device, int64_t N, const ssf_rx_params *params, const void *Es, const void *Elo, const void *taps, int32_t ntaps,
                 int32_t SpSin, int32_t decFactor, const void *edc_Hfft, int32_t edc_K, int32_t edc_nfft, void *sig_out,
                 int32_t *sampDelay) {
    if (!params || !Es || !Elo || !taps || !edc_Hfft || !sig_out) return set_err(SSF_ERR_BAD_ARG, "ssf_rx_chain: NULL argument");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::rx_chain(device, N, params, Es, Elo, taps, ntaps, SpSin, decFactor, edc_Hfft, edc_K, edc_nfft, sig_out, sampDelay, &err);
    return rc ? set_err(rc, "ssf_rx_chain: " + err) : SSF_OK;
}

int ssf_wdm_tx(int device, const ssf_tx_params *params, const void *symbols, const double *taps, const double *phi,
               const double *amp, const double *deltaF, void *sig_out, double *power_out) {
    if (!params || !symbols || !taps || !amp || !deltaF || !sig_out) return set_err(SSF_ERR_BAD_ARG, "ssf_wdm_tx: NULL argument");
    if (int rc = rx_check_device(device)) return rc;
    std::string err;
    int rc = ssf::tx_wdm(device, params, symbols, taps, phi, amp, deltaF, sig_out, power_out, &err);
    return rc ? set_err(rc, "ssf_wdm_tx: " + err) : SSF_OK;
}

}  // extern "C"
