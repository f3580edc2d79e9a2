// Host-side sequencing of the receiver front-end stages (Backend = HIP or the CPU emulator).
// Everything between the upload of the inputs and the download of the result stays in device
// memory: PBS -> [polarisation delay] -> hybrid + photodiodes -> [low-pass FIR] -> IQ imbalance ->
// skew filters -> result.  Filters are derived here from the raw parameters with the same
// formulas as the reference, so the host wrapper and the library cannot drift.
//
// Reference: optic/models/devices.py:289-668, optic/dsp/core.py:352-392 (lowPassFIR),
// 880-922 (delaySignal), 925-970 (iqMixing), 973-1046 (blockwiseFFTConv).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <string>
#include <vector>

#include "fused_kernels.h"
#include "rx_kernels.h"
#include "ssf.h"

namespace ssf {
namespace rx {

typedef std::complex<double> zc;
constexpr double kPi = 3.14159265358979323846;

// in-place radix-2 FFT (n = power of two), sign = -1 forward / +1 inverse (unscaled).  The n/2 twiddles of
// a size are computed once per thread (direct cos / sin of every angle, no recurrences) and reused.
inline const std::vector<zc> &host_twiddles(size_t n) {
    thread_local std::vector<std::vector<zc>> cache(32);
    size_t lg = 0;
    while (((size_t)1 << lg) < n) ++lg;
    std::vector<zc> &w = cache[lg];
    if (w.size() != n / 2) {
        w.resize(n / 2);
        for (size_t k = 0; k < n / 2; ++k) {
            const double ang = -2.0 * kPi * (double)k / (double)n;
            w[k] = zc(std::cos(ang), std::sin(ang));
        }
    }
    return w;
}
inline void host_fft(std::vector<zc> &a, int sign) {
    const size_t n = a.size();
    if (n < 2) return;
    const std::vector<zc> &w = host_twiddles(n);
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t step = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const zc t = sign < 0 ? w[k * step] : std::conj(w[k * step]);
                const zc u = a[i + k], v = a[i + k + len / 2] * t;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

// optic/dsp/core.py:352-392
inline std::vector<double> low_pass_fir(double fc, double fs, int N, int gauss) {
    std::vector<double> h((size_t)N);
    const double fu = fc / fs, d = (N - 1) / 2.0;
    double sum = 0;
    for (int n = 0; n < N; ++n) {
        const double x = n - d;
        if (!gauss) {
            const double t = kPi * 2 * fu * x;                       // np.sinc(y) = sin(pi y) / (pi y)
            h[(size_t)n] = (2 * fu) * (t == 0.0 ? 1.0 : std::sin(t) / t);
        } else {
            const double a = kPi * fu * x;
            h[(size_t)n] = std::sqrt(2 * kPi / std::log(2.0)) * fu * std::exp(-(2 / std::log(2.0)) * a * a);
        }
        sum += h[(size_t)n];
    }
    for (auto &v : h) v /= sum;
    return h;
}

// H = fft(zero-padded taps) / nfft: what ols_body multiplies with (the 1/nfft of the inverse transform folded in)
inline std::vector<zc> ols_filter_from_taps(const zc *taps, int K, int nfft) {
    std::vector<zc> h((size_t)nfft, zc(0, 0));
    for (int i = 0; i < K; ++i) h[(size_t)i] = taps[i];
    host_fft(h, -1);
    for (auto &v : h) v /= (double)nfft;
    int lg = 0;
    while ((1 << lg) < nfft) ++lg;
    fused::ols_permute_filter(h.data(), lg);                         // the order the kernel reads (fused_kernels.h: ols_body_x)
    return h;
}

// delaySignal's filter (core.py:909-916 + blockwiseFFTConv's freqDomainFilter branch, core.py:1015-1020):
// K = nfft/2 frequency samples exp(-j 2 pi f delay) -> centred impulse response -> zero-padded -> fft
inline std::vector<zc> ols_filter_from_delay(double delay, double Fs, int K, int nfft) {
    std::vector<zc> Hk((size_t)K);
    for (int i = 0; i < K; ++i) {
        const int kk = i < (K + 1) / 2 ? i : i - K;                  // np.fft.fftfreq(K, 1 / Fs)
        const double f = (double)kk / ((double)K * (1.0 / Fs));
        const double ang = -2.0 * kPi * f * delay;
        Hk[(size_t)i] = zc(std::cos(ang), std::sin(ang));
    }
    host_fft(Hk, +1);                                                // ifft (scaled below)
    std::vector<zc> taps((size_t)K);
    for (int i = 0; i < K; ++i) taps[(size_t)((i + K / 2) % K)] = Hk[(size_t)i] / (double)K;   // fftshift (K even)
    return ols_filter_from_taps(taps.data(), K, nfft);
}

struct OlsGeom {
    int nfft, lg, K, d, discard, D;
    long long numBlocks;
};
inline OlsGeom ols_geometry(long long sigLen, int K, int nfft) {
    OlsGeom g;
    g.nfft = nfft;
    g.lg = 0;
    while ((1 << g.lg) < nfft) ++g.lg;
    g.K = K;
    g.d = nfft - K + 1;
    g.discard = K - 1;
    g.D = (K - 1) / 2;
    g.numBlocks = (sigLen + K - 1 + g.d - 1) / g.d;                  // core.py:1023-1025
    return g;
}
// transform size for a K-tap 'same' filter: the block advance nfft - K + 1 should be most of the block
// Measured (profiles/r5_rx_ab.txt): a 2048-point block with two columns side by side is a 256-thread workgroup with 68 KiB of LDS,
// two per CU; a 4096-point pair needs 512 threads and 136 KiB, one per CU, and is ~30 % slower per byte -- so 2048 points up to
// 682 taps (two thirds of the block is output), 4096 up to 2048 taps, 8192 above
inline int fir_nfft(int K) {
    int nfft = 256;
    while (nfft < 8 * K && nfft < 2048) nfft <<= 1;
    while (nfft < 3 * K && nfft < 4096) nfft <<= 1;
    while (nfft < 2 * K && nfft < 8192) nfft <<= 1;          // (2049 ... 4096 taps: 4096-point blocks would advance by a few samples)
    while (nfft < K) nfft <<= 1;
    return nfft;
}
#ifndef SSF_DELAY_NFFT
#define SSF_DELAY_NFFT 2048
#endif
constexpr int kDelayNfft = SSF_DELAY_NFFT;   // block size of the 512-tap fractional-delay filters (delaySignal, polarisation delay, IQ skew)
constexpr int kMaxNfft = 8192;            // c128 rows of the LDS transform (engine_fused_impl.h: k_ols)

template <class Backend> struct RxCore {
    Backend &be;
    std::string err;
    std::vector<void *> owned;
    explicit RxCore(Backend &b) : be(b) {}
    ~RxCore() {
        for (void *p : owned) be.free(p);
    }
    Cd *dalloc(size_t n) {
        void *p = be.alloc(sizeof(Cd) * (n ? n : 1));
        if (p) owned.push_back(p);
        return (Cd *)p;
    }
    int fail(int rc, const std::string &m) {
        err = m;
        return rc;
    }
    Cd *upload_filter(const std::vector<zc> &H) {
        Cd *d = dalloc(H.size());
        if (d) be.h2d(d, H.data(), sizeof(Cd) * H.size());
        return d;
    }

    // y[:keep] = roll(blockwiseFFTConv(in zero-extended to sigLen, filter), -roll) for `ncols` columns
    int ols(const Cd *in, int in_ld, long long inLen, long long sigLen, Cd *out, int out_ld, long long keep, int ncols,
            const Cd *H, int Hstride, int K, int nfft, int roll, int in_up = 1) {
        be.launch_ols(ols_args(in, in_ld, inLen, sigLen, out, out_ld, keep, ncols, H, Hstride, K, nfft, roll, in_up));
        return SSF_OK;
    }
    fused::OlsArgs<double> ols_args(const Cd *in, int in_ld, long long inLen, long long sigLen, Cd *out, int out_ld, long long keep, int ncols,
                                    const Cd *H, int Hstride, int K, int nfft, int roll, int in_up = 1) {
        const OlsGeom g = ols_geometry(sigLen, K, nfft);
        fused::OlsArgs<double> a{};
        a.in = in;
        a.out = out;
        a.H = H;
        a.sigLen = sigLen;
        a.njobs = g.numBlocks * ncols;
        a.nrows = ncols;
        a.log2nfft = g.lg;
        a.d = g.d;
        a.discard = g.discard;
        a.D = g.D;
        a.inLen = inLen;
        a.keep = keep;
        a.in_ld = in_ld;
        a.out_ld = out_ld;
        a.Hstride = Hstride;
        a.roll = roll;
        a.in_up = in_up;
        return a;
    }

    // delaySignal on columns [c0, c0 + 2) of a (N, ld) array with delays (dl[0], dl[1]) of equal magnitude
    int delay_pair(const Cd *in, Cd *out, int ld, long long N, const double *dl, int ncols, double Fs) {
        // the reference's filter: NFFT = 1024 -> a 512-tap impulse response (core.py:880, 909-916).  The block
        // size of the overlap-save evaluation does not change the convolution, so larger blocks are
        // used here (kDelayNfft = 2048: 75 % of every transform is output, 50 % with 1024; 4096-point
        // blocks measured slower, see fir_nfft)
        const int K = 512, nfft = kDelayNfft;
        const long long padLen = (long long)std::ceil(std::fabs(dl[0] * Fs));
        Cd *dH = delay_filters(dl, ncols, Fs, K, nfft);
        if (!dH) return fail(SSF_ERR_OOM, "out of device memory");
        return ols(in, ld, N, N + padLen, out, ld, N, ncols, dH, nfft, K, nfft, 1);
    }

    // an input array where the kernels can read it: a device pointer as it is (no copy), a host array uploaded
    const Cd *resident(const void *p, size_t n) {
        if (be.is_resident(p)) return (const Cd *)p;
        Cd *d = dalloc(n);
        if (d) be.h2d_big(d, p, sizeof(Cd) * n);
        return d;
    }
    // ... and a result where it belongs: a device destination is written by the kernels themselves (unless it is the input:
    // the filters read their neighbourhood), a host destination gets a scratch block that is downloaded at the end
    Cd *result_buffer(void *out, const void *in, size_t n) {
        if (be.is_resident(out) && out != in) return (Cd *)out;
        return dalloc(n);
    }
    int finish(void *out, const Cd *res, size_t n) {
        be.sync();
        if (!be.ok()) return fail(SSF_ERR_HIP, be.last_error());
        if ((const void *)res != out) be.d2h_big(out, res, sizeof(Cd) * n);
        return be.ok() ? SSF_OK : fail(SSF_ERR_HIP, be.last_error());
    }
    // filters derived from a few numbers (delays, the photodiodes' low-pass) or from taps the caller passes again and again (a
    // matched filter): the backend may keep their device images between calls (host FFT + a synchronising upload per filter cost
    // as much as the kernels of a 2^20-sample call)
    struct FilterKey {
        int kind, K, nfft, x;
        double a, b;
        unsigned long long h;
        unsigned long long h2 = 0;    // content keys (kinds 3, 4): a second, independent hash of the same values (with the length in K / nfft:
                                      // 128 bits of content + length -- a single 64-bit FNV collision would silently apply the wrong filter)
        bool operator==(const FilterKey &o) const {
            return kind == o.kind && K == o.K && nfft == o.nfft && x == o.x && a == o.a && b == o.b && h == o.h && h2 == o.h2;
        }
    };
    template <class Gen> Cd *cached_filter(const FilterKey &k, Gen &&gen) {
        if (void *d = be.filter_lookup(&k, sizeof(k))) return (Cd *)d;
        const std::vector<zc> H = gen();
        if (void *d = be.filter_store(&k, sizeof(k), H.data(), sizeof(Cd) * H.size())) return (Cd *)d;
        return upload_filter(H);
    }
    Cd *delay_filters(const double *dl, int n, double Fs, int K, int nfft) {         // n delay filters side by side
        FilterKey k{1, K, nfft, n, dl[0], Fs, 0};
        for (int c = 1; c < n; ++c) {
            unsigned long long bits;
            std::memcpy(&bits, &dl[c], 8);
            k.h = k.h * 1099511628211ull + bits;
        }
        return cached_filter(k, [&] {
            std::vector<zc> H;
            for (int c = 0; c < n; ++c) {
                const std::vector<zc> h = ols_filter_from_delay(dl[c], Fs, K, nfft);
                H.insert(H.end(), h.begin(), h.end());
            }
            return H;
        });
    }
    Cd *lowpass_filter(double B, double fs, int ntaps, int fType, int nfft) {
        FilterKey k{2, ntaps, nfft, fType, B, fs, 0};
        return cached_filter(k, [&] {
            const std::vector<double> h = low_pass_fir(B, fs, ntaps, fType);
            std::vector<zc> hz(h.begin(), h.end());
            return ols_filter_from_taps(hz.data(), ntaps, nfft);
        });
    }
    static unsigned long long content_hash(const zc *v, size_t n) {              // FNV-1a over the 64-bit words of the values
        unsigned long long h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) {
            unsigned long long w[2];
            std::memcpy(w, &v[i], 16);
            h = (h ^ w[0]) * 1099511628211ull;
            h = (h ^ w[1]) * 1099511628211ull;
        }
        return h;
    }
    static unsigned long long content_hash2(const zc *v, size_t n) {             // splitmix64-mixed words, position-dependent sum
        unsigned long long h = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < n; ++i) {
            unsigned long long w[2];
            std::memcpy(w, &v[i], 16);
            for (int q = 0; q < 2; ++q) {
                unsigned long long z = w[q] + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * i + q + 1);
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                h = (h << 5 | h >> 59) + (z ^ (z >> 31));
            }
        }
        return h;
    }
    Cd *taps_filter(const zc *taps, int K, int nfft) {
        FilterKey k{3, K, nfft, 0, 0.0, 0.0, content_hash(taps, (size_t)K), content_hash2(taps, (size_t)K)};
        return cached_filter(k, [&] { return ols_filter_from_taps(taps, K, nfft); });
    }
    static void iq_gains(const ssf_rx_params &p, int k, Cd *k1o, Cd *k2o) {      // core.py:952-959
        const double amp = std::pow(10.0, p.ampImb[k] / 20) - 1, ph = p.phaseImb[k];
        const zc ep(std::cos(ph / 2), std::sin(ph / 2)), em(std::cos(ph / 2), -std::sin(ph / 2));
        const zc k1 = (1 - amp) * ep / 2.0 + (1 + amp) * em / 2.0, k2 = (1 - amp) * em / 2.0 - (1 + amp) * ep / 2.0;
        *k1o = mk<double>(k1.real(), k1.imag());
        *k2o = mk<double>(k2.real(), k2.imag());
    }
    RxOlsArgs rx_ols_args(long long inLen, long long sigLen, Cd *out, int out_ld, long long keep, int ncols, const Cd *H, int Hstride,
                          int K, int nfft, int roll) {
        const OlsGeom g = ols_geometry(sigLen, K, nfft);
        RxOlsArgs r{};
        fused::OlsArgs<double> &a = r.o;
        a.out = out;
        a.H = H;
        a.sigLen = sigLen;
        a.njobs = g.numBlocks * ncols;
        a.nrows = ncols;
        a.log2nfft = g.lg;
        a.d = g.d;
        a.discard = g.discard;
        a.D = g.D;
        a.inLen = inLen;
        a.keep = keep;
        a.in_ld = ncols;
        a.out_ld = out_ld;
        a.Hstride = Hstride;
        a.roll = roll;
        a.in_up = 1;
        return r;
    }
    // coherentReceiver / pdmCoherentReceiver / iqMixing in at most three launches (rx_kernels.h: rx_ols_body); device inputs are
    // read where they are and a device result is written where it belongs
    int run_coherent(int mode, long long N, const ssf_rx_params &p, const void *in0, const void *lo, const double *un, void *out) {
        const bool iq_only = mode == SSF_RX_IQ_MIXING;
        const int nm = mode == SSF_RX_PDM_COHERENT ? 2 : iq_only ? 1 : 1;
        const bool quiet = p.ideal != 0;
        const bool noisy = !iq_only && !quiet && (p.shotNoise || p.thermalNoise);
        const bool lowpass = !iq_only && !quiet && p.bandwidthLimitation;
        const double fs_pd = p.Fs_pd > 0 ? p.Fs_pd : p.Fs;
        int ntaps = p.N;
        if (ntaps % 2 == 0) ++ntaps;                                 // devices.py:361-365
        const Cd *sig = resident(in0, (size_t)N * nm);
        const Cd *dlo = iq_only ? nullptr : resident(lo, (size_t)N);
        // (a device result that IS one of the inputs goes through a scratch block like fir / decimate's: the filters read the
        //  neighbourhood of every sample they write -- iqMixing with a skew reads in0 itself)
        const bool alias = out == in0 || (!iq_only && out == lo);
        Cd *result = be.is_resident(out) && !alias ? (Cd *)out : dalloc((size_t)N * nm);
        if (!sig || (!iq_only && !dlo) || !result) return fail(SSF_ERR_OOM, "out of device memory");
        double *dun = nullptr;
        if (noisy && un) {
            const int npd = 4 * nm;
            dun = (double *)be.alloc(sizeof(double) * (size_t)N * npd * 2);
            if (!dun) return fail(SSF_ERR_OOM, "out of device memory");
            owned.push_back(dun);
            be.h2d_big(dun, un, sizeof(double) * (size_t)N * npd * 2);
        }
        Cd k1[2], k2[2];
        bool skew = false;
        for (int k = 0; k < nm; ++k) {
            iq_gains(p, k, &k1[k], &k2[k]);
            skew = skew || p.timeSkew[k] != 0;
        }
        const Cd *detected = sig;                                    // iqMixing by itself: the input is the detected signal
        if (!iq_only) {
            DetArgs det{};
            det.in0 = sig;
            det.lo = dlo;
            det.N = N;
            det.nm = nm;
            det.es_scale[0] = det.es_scale[1] = 1.0;
            det.lo_scale[0] = det.lo_scale[1] = 1.0;
            const double q = 1.602176634e-19, kB = 1.380649e-23;     // scipy.constants (CODATA 2018, exact)
            det.pd.R = p.R;
            det.pd.IpdSat = p.IpdSat;
            det.pd.saturate = !quiet && p.currentSaturation;
            det.pd.shot = !quiet && p.shotNoise;
            det.pd.thermal = !quiet && p.thermalNoise;
            det.pd.shot_k = fs_pd * q;
            det.pd.Id = p.Id;
            det.pd.thermal_sigma = std::sqrt(fs_pd * (4 * kB * (p.Tc + 273.15) * p.B / p.RL) / (2 * p.B));
            det.pd.seed = (unsigned long long)p.rng_seed;
            det.pd.un = dun;
            bool detected_already = false;
            if (mode == SSF_RX_PDM_COHERENT) {
                det.pbs = 1;
                det.c = std::cos(p.polRotation);
                det.s = std::sin(p.polRotation);
                if (p.pdl != 0) {
                    det.es_scale[0] = std::pow(10.0, -(p.pdl / 2) / 20);
                    det.es_scale[1] = std::pow(10.0, (p.pdl / 2) / 20);
                }
                det.lo_scale[0] = std::cos(kPi / 4);
                det.lo_scale[1] = -std::sin(kPi / 4);
                if (p.polDelay != 0) {                               // devices.py:656-658: x delayed by -polDelay/2, y by +polDelay/2
                    const int K = 512, nfft = kDelayNfft;                  // (delay_pair's filter and block size)
                    const double dl[2] = {-p.polDelay / 2, p.polDelay / 2};
                    const long long padLen = (long long)std::ceil(std::fabs(dl[0] * p.Fs));
                    Cd *dH = delay_filters(dl, 2, p.Fs, K, nfft), *fld = dalloc((size_t)N * 2);
                    if (!dH || !fld) return fail(SSF_ERR_OOM, "out of device memory");
                    // ideal photodiodes and no skew: what follows the delay filters is element-wise (hybrid, photodiodes, IQ imbalance)
                    // and rides in their stores -- the notebook's receiver is ONE launch (round 6; two before)
                    const bool det_in_stores = !lowpass && !skew && !det.pd.shot && !det.pd.thermal;
                    RxOlsArgs r = rx_ols_args(N, N + padLen, det_in_stores ? result : fld, 2, N, 2, dH, nfft, K, nfft, 1);
                    r.pre = det_in_stores ? PRE_PBS_DET : PRE_PBS;
                    r.det = det;
                    if (det_in_stores) {
                        r.N = N;
                        for (int k = 0; k < nm; ++k) {
                            r.k1[k] = k1[k];
                            r.k2[k] = k2[k];
                        }
                        detected_already = true;
                        detected = result;
                    }
                    be.launch_rx_ols(r);
                    det.in0 = fld;
                    det.pbs = 0;
                }
            }
            if (!detected_already) {
            Cd *dst = skew ? dalloc((size_t)N * nm) : result;
            if (!dst) return fail(SSF_ERR_OOM, "out of device memory");
            if (lowpass) {
                const int nfft = fir_nfft(ntaps);
                Cd *dH = lowpass_filter(p.B, fs_pd, ntaps, p.fType, nfft);
                if (!dH) return fail(SSF_ERR_OOM, "out of device memory");
                RxOlsArgs r = rx_ols_args(N, N, dst, nm, N, nm, dH, 0, ntaps, nfft, 0);
                r.pre = PRE_DET;
                r.det = det;
                r.post = skew ? POST_PLAIN : POST_IQF;
                r.N = N;
                for (int k = 0; k < nm; ++k) {
                    r.k1[k] = k1[k];
                    r.k2[k] = k2[k];
                }
                be.launch_rx_ols(r);
            } else {
                DetKernelArgs d{};
                d.det = det;
                d.out = dst;
                d.iqf = skew ? 0 : 1;
                for (int k = 0; k < nm; ++k) {
                    d.k1[k] = k1[k];
                    d.k2[k] = k2[k];
                }
                be.launch_det(d);
            }
            detected = dst;
            }
        }
        if (skew) {                                                  // iqMixing with a skew (core.py:962-968)
            // sI delayed by -skew/2, sQ by +skew/2; the zero padding of delaySignal (core.py:905-909) differs between the
            // polarisations when their skews do: then one launch per polarisation
            long long pad[2] = {0, 0};
            for (int k = 0; k < nm; ++k) pad[k] = (long long)std::ceil(std::fabs(p.timeSkew[k] / 2 * p.Fs));
            const bool together = nm == 1 || pad[0] == pad[1];
            const int K = 512, nfft = kDelayNfft;
            double dls[4];
            for (int k = 0; k < nm; ++k)
                for (int part = 0; part < 2; ++part) dls[2 * k + part] = (part ? 1.0 : -1.0) * p.timeSkew[k] / 2;
            Cd *dH = delay_filters(dls, 2 * nm, p.Fs, K, nfft);
            if (!dH) return fail(SSF_ERR_OOM, "out of device memory");
            for (int k = 0; k < (together ? 1 : nm); ++k) {
                const int ncols = together ? 2 * nm : 2;
                RxOlsArgs r = rx_ols_args(N, N + pad[k], result, nm, N, ncols, dH + (together ? 0 : (size_t)2 * k * nfft), nfft, K, nfft, 1);
                r.pre = PRE_IQ;
                r.post = POST_PART;
                r.o.in = detected;
                r.nm = nm;
                r.pol0 = together ? 0 : k;
                for (int q2 = 0; q2 < nm; ++q2) {
                    r.k1[q2] = k1[q2];
                    r.k2[q2] = k2[q2];
                }
                be.launch_rx_ols(r);
            }
        } else if (iq_only) {
            IqfArgs f{};
            f.in = detected;
            f.out = result;
            f.N = N;
            f.nm = nm;
            for (int k = 0; k < nm; ++k) {
                f.k1[k] = k1[k];
                f.k2[k] = k2[k];
            }
            be.launch_iqf(f);
        }
        if (in_chain) return be.ok() ? SSF_OK : fail(SSF_ERR_HIP, be.last_error());      // (chain(): more launches follow on the stream)
        be.sync();
        if (!be.ok()) return fail(SSF_ERR_HIP, be.last_error());
        if ((void *)result != out) be.d2h_big(out, result, sizeof(Cd) * (size_t)N * nm);
        return be.ok() ? SSF_OK : fail(SSF_ERR_HIP, be.last_error());
    }
    bool in_chain = false;

    // pdmCoherentReceiver -> firFilter (matched filter) -> decimate -> edc in ONE call (include/ssf.h: ssf_rx_chain): the stages'
    // launches follow each other on the stream, the host waits once at the end; the decimation's variance search rides in the
    // matched filter's stores and its gather in the compensating filter's loads (rx_kernels.h: chain_ols_body).  Results are those
    // of the four calls made one after the other (reference: optic/models/devices.py:574-668, optic/dsp/core.py:87-125, 435-491,
    // optic/dsp/equalization.py:36-122).
    int chain(long long N, const ssf_rx_params &p, const void *Es, const void *Elo, const void *taps, int ntaps, int SpSin,
              int decFactor, const void *edcH, int edcK, int edc_nfft, void *out, int *sampDelay_out) {
        const int nm = 2;
        if (N < 1 || ntaps < 1 || SpSin < 1 || decFactor < 1 || edcK < 1 || edcK > edc_nfft) return fail(SSF_ERR_BAD_ARG, "bad size");
        if (N % SpSin) return fail(SSF_ERR_BAD_ARG, "cannot reshape array: length is not a multiple of SpSin");   // core.py:477
        if (ntaps > kMaxNfft / 2) return fail(SSF_ERR_UNSUPPORTED, "receiver chain: matched filter of at most 4096 taps");
        const long long Nout = (N + decFactor - 1) / decFactor;
        const int nfft = fir_nfft(ntaps);
        const OlsGeom g = ols_geometry(N, ntaps, nfft);
        const fused::OlsLaunch lo = fused::ols_launch(g.lg, nm, g.numBlocks * nm);
        const int nclass = SpSin * nm;
        const bool fuse_stats = chain_ols_supported(lo) && ((nfft / 16) % SpSin) == 0 && nclass <= 256;
        // 1. receiver into a device block
        Cd *S = dalloc((size_t)N * nm);
        if (!S) return fail(SSF_ERR_OOM, "out of device memory");
        in_chain = true;
        int rc = run(SSF_RX_PDM_COHERENT, N, nm, p, Es, Elo, nullptr, S);
        in_chain = false;
        if (rc) return rc;
        // 2. matched filter; the class sums of the decimation in its stores where the geometry allows
        Cd *F = dalloc((size_t)N * nm), *dH = taps_filter((const zc *)taps, ntaps, nfft);
        int *ddelay = (int *)be.alloc(sizeof(int) * 8);
        if (ddelay) owned.push_back(ddelay);
        if (!F || !dH || !ddelay) return fail(SSF_ERR_OOM, "out of device memory");
        ChainOlsArgs ca{};
        ca.o = ols_args(S, nm, N, N, F, nm, N, nm, dH, 0, ntaps, nfft, 0);
        if (fuse_stats) {
            const int nparts = (int)lo.grid * (lo.threads / ((nfft / 16) * lo.C));      // one partial per block group
            double *dpart = (double *)be.alloc(sizeof(double) * 3 * (size_t)nparts * nclass);
            if (!dpart) return fail(SSF_ERR_OOM, "out of device memory");
            owned.push_back(dpart);
            ca.part = dpart;
            ca.SpS = SpSin;
            be.launch_chain_ols(ca, CH_STATS);
            ChainFinishArgs fa{dpart, ddelay, nparts, nclass, nm, SpSin, (double)(N / SpSin)};
            be.launch_chain_finish(fa);
        } else {                                                     // the filter, then decimate's own two passes
            be.launch_ols(ca.o);
            const int nthreads = 256 / nclass * nclass;
            if (nclass > 256) return fail(SSF_ERR_UNSUPPORTED, "decimate: SpSin * columns <= 256");
            const int nblocks = (int)std::max<long long>(1, std::min<long long>(512, (N * nm + 4 * nthreads - 1) / (4 * nthreads)));
            Cd *dmean = dalloc((size_t)nclass);
            double *dpart = (double *)be.alloc(sizeof(double) * 2 * (size_t)nblocks * nclass);
            if (dpart) owned.push_back(dpart);
            if (!dmean || !dpart) return fail(SSF_ERR_OOM, "out of device memory");
            DecSumArgs sa{F, nullptr, dpart, N * nm, nclass};
            DecFinishArgs fa{dpart, dmean, ddelay, nblocks, nclass, nm, SpSin, (double)(N / SpSin)};
            be.launch_dec_sum(sa, nblocks, nthreads);
            be.launch_dec_finish(fa);
            sa.mean = dmean;
            fa.mean = nullptr;
            be.launch_dec_sum(sa, nblocks, nthreads);
            be.launch_dec_finish(fa);
        }
        // 3. edc on the decimated signal, gathered in its loads
        int lg = 0;
        while ((1 << lg) < edc_nfft) ++lg;
        if ((1 << lg) != edc_nfft || lg < 4 || lg > 13) return fail(SSF_ERR_UNSUPPORTED, "receiver chain: edc block size must be a power of two in [16, 8192]");
        Cd *dHe = response_filter(edcH, edcK, edc_nfft);
        Cd *b = be.is_resident(out) && out != Es && out != Elo ? (Cd *)out : dalloc((size_t)Nout * nm);
        if (!dHe || !b) return fail(SSF_ERR_OOM, "out of device memory");
        const OlsGeom ge = ols_geometry(Nout, edcK, edc_nfft);
        const fused::OlsLaunch le = fused::ols_launch(ge.lg, nm, ge.numBlocks * nm);
        ChainOlsArgs ce{};
        ce.o = ols_args(F, nm, Nout, Nout, b, nm, Nout, nm, dHe, 0, edcK, edc_nfft, 0);
        if (chain_ols_supported(le)) {
            ce.delay = ddelay;
            ce.dec = decFactor;
            ce.Nfull = N;
            be.launch_chain_ols(ce, CH_GATHER);
        } else {                                                     // gather by itself, then the filter
            Cd *dec = dalloc((size_t)Nout * nm);
            if (!dec) return fail(SSF_ERR_OOM, "out of device memory");
            DecGatherArgs ga{};
            ga.in = F;
            ga.out = dec;
            ga.N = N;
            ga.Nout = Nout;
            ga.ncols = nm;
            ga.dec = decFactor;
            ga.delay = ddelay;
            be.launch_dec_gather(ga);
            ce.o.in = dec;
            be.launch_ols(ce.o);
        }
        if (sampDelay_out) {
            int dl[8];
            be.d2h(dl, ddelay, sizeof(int) * (size_t)nm);            // (waits for the stream)
            for (int c = 0; c < nm; ++c) sampDelay_out[c] = dl[c];
        }
        return finish(out, b, (size_t)Nout * nm);
    }

    // run one of the ssf_rx_mode pipelines; in0 / lo / un / out are HOST pointers
    int run(int mode, long long N, int nmodes, const ssf_rx_params &p, const void *in0, const void *lo, const double *un,
            void *out) {
        const bool coherent = mode == SSF_RX_COHERENT || mode == SSF_RX_PDM_COHERENT;
        const bool iq_only = mode == SSF_RX_IQ_MIXING;
        const int nin = mode == SSF_RX_PHOTODIODE ? nmodes : mode == SSF_RX_BALANCED_PD ? 2 : mode == SSF_RX_PDM_COHERENT ? 2 : 1;
        const int npd = mode == SSF_RX_PHOTODIODE ? 1 : 2;
        if (N < 1 || nin < 1) return fail(SSF_ERR_BAD_ARG, "bad size");
        if (!iq_only && !(p.R > 0)) return fail(SSF_ERR_BAD_ARG, "PD responsivity should be a positive scalar");
        const bool quiet = p.ideal != 0;
        const bool noisy = !iq_only && !quiet && (p.shotNoise || p.thermalNoise);
        const bool lowpass = !iq_only && !quiet && p.bandwidthLimitation;
        const double fs_pd = p.Fs_pd > 0 ? p.Fs_pd : p.Fs;           // paramPD.Fs (noise, low-pass) vs paramFE.Fs (delays): ssf.h
        if (!iq_only && !quiet && !(fs_pd >= 2 * p.B)) return fail(SSF_ERR_BAD_ARG, "Sampling frequency Fs needs to be at least twice of B.");
        int ntaps = p.N;
        if (ntaps % 2 == 0) ++ntaps;                                 // devices.py:361-365
        if (lowpass && (ntaps < 1 || ntaps > kMaxNfft / 2)) return fail(SSF_ERR_UNSUPPORTED, "photodiode filter: 1 <= N <= 4096 taps");
        if (coherent || iq_only) return run_coherent(mode, N, p, in0, lo, un, out);

        // photodiode / balancedPD: detection -> [low-pass FIR] -> real photocurrent, ONE launch: the detection rides in the filter's
        // loads, the real part is what its stores write (an element-wise launch when there is no filter); a device input is read
        // where it is and a device result written where it belongs
        const Cd *sig = resident(in0, (size_t)N * nin);
        double *res = be.is_resident(out) ? (double *)out : (double *)dalloc((size_t)(N + 1) / 2);
        double *dun = nullptr;
        if (!sig || !res) return fail(SSF_ERR_OOM, "out of device memory");
        if (noisy && un) {
            dun = (double *)be.alloc(sizeof(double) * (size_t)N * npd * 2);
            if (!dun) return fail(SSF_ERR_OOM, "out of device memory");
            owned.push_back(dun);
            be.h2d_big(dun, un, sizeof(double) * (size_t)N * npd * 2);
        }
        PdFront fr{};
        fr.in0 = sig;
        fr.N = N;
        fr.mode = mode == SSF_RX_PHOTODIODE ? RX_PHOTODIODE : RX_BALANCED;
        fr.nm = nin;
        const double q = 1.602176634e-19, kB = 1.380649e-23;         // scipy.constants (CODATA 2018, exact)
        fr.pd.R = p.R;
        fr.pd.IpdSat = p.IpdSat;
        fr.pd.saturate = !quiet && p.currentSaturation;
        fr.pd.shot = !quiet && p.shotNoise;
        fr.pd.thermal = !quiet && p.thermalNoise;
        fr.pd.shot_k = fs_pd * q;
        fr.pd.Id = p.Id;
        fr.pd.thermal_sigma = std::sqrt(fs_pd * (4 * kB * (p.Tc + 273.15) * p.B / p.RL) / (2 * p.B));
        fr.pd.seed = (unsigned long long)p.rng_seed;
        fr.pd.un = dun;
        if (lowpass) {
            const int nfft = fir_nfft(ntaps);
            Cd *dH = lowpass_filter(p.B, fs_pd, ntaps, p.fType, nfft);
            if (!dH) return fail(SSF_ERR_OOM, "out of device memory");
            RxOlsArgs r = rx_ols_args(N, N, (Cd *)res, 1, N, 1, dH, 0, ntaps, nfft, 0);
            r.pre = PRE_PD;
            r.post = POST_REAL;
            r.front = fr;
            be.launch_rx_ols(r);
        } else {
            FrontArgs fa{fr, res};
            be.launch_front(fa);
        }
        be.sync();
        if (!be.ok()) return fail(SSF_ERR_HIP, be.last_error());
        if ((void *)res != out) be.d2h_big(out, res, sizeof(double) * (size_t)N);
        return be.ok() ? SSF_OK : fail(SSF_ERR_HIP, be.last_error());
    }

    // y += alpha x, n float64 values, host or device pointers (a device y is updated in place)
    int axpy(long long n, double alpha, const void *x, void *y) {
        if (n < 1) return fail(SSF_ERR_BAD_ARG, "bad size");
        const size_t nc = (size_t)(n + 1) / 2;                       // (dalloc counts complex values)
        const double *dx = (const double *)resident(x, nc);
        double *dy = be.is_resident(y) ? (double *)y : (double *)dalloc(nc);
        if (!dx || !dy) return fail(SSF_ERR_OOM, "out of device memory");
        if ((void *)dy != y) be.h2d_big(dy, y, sizeof(double) * (size_t)n);
        AxpyArgs a{dx, dy, n, alpha};
        be.launch_axpy(a);
        be.sync();
        if (!be.ok()) return fail(SSF_ERR_HIP, be.last_error());
        if ((void *)dy != y) be.d2h_big(y, dy, sizeof(double) * (size_t)n);
        return be.ok() ? SSF_OK : fail(SSF_ERR_HIP, be.last_error());
    }

    // edfa / pbs / opticalHybrid2x4 by themselves (rx_kernels.h: optics_body): host or device pointers, a device result is written in place
    int optics(int op, long long n, int ncols, double p0, double p1, unsigned long long seed, unsigned row0, const void *a,
               const void *b, void *o0, void *o1) {
        if (n < 1 || !a || !o0) return fail(SSF_ERR_BAD_ARG, "bad argument");
        if (op == OPT_PBS && ((ncols != 1 && ncols != 2) || !o1)) return fail(SSF_ERR_BAD_ARG, "pbs: E must be (N,) or (N, 2)");
        if (op == OPT_HYBRID && !b) return fail(SSF_ERR_BAD_ARG, "opticalHybrid2x4: Elo is NULL");
        if (op == OPT_EDFA && ncols < 1) return fail(SSF_ERR_BAD_ARG, "edfa: bad column count");
        const size_t n_in = op == OPT_PBS ? (size_t)n * ncols : (size_t)n, n_o0 = op == OPT_HYBRID ? 4 * (size_t)n : (size_t)n;
        OpticsArgs g{};
        g.op = op;
        g.ncols = ncols;
        g.a = resident(a, n_in);
        g.b = b ? resident(b, (size_t)n) : nullptr;
        // (element-wise: an in-place EDFA is fine; the splitter's and the hybrid's outputs must not be their inputs)
        const bool alias0 = op != OPT_EDFA && (o0 == a || o0 == b), alias1 = o1 && (o1 == a || o1 == b || o1 == o0);
        g.o0 = be.is_resident(o0) && !alias0 ? (Cd *)o0 : dalloc(n_o0);
        g.o1 = !o1 ? nullptr : be.is_resident(o1) && !alias1 ? (Cd *)o1 : dalloc((size_t)n);
        if (!g.a || (b && !g.b) || !g.o0 || (o1 && !g.o1)) return fail(SSF_ERR_OOM, "out of device memory");
        g.n = n;
        g.p0 = p0;
        g.p1 = p1;
        g.seed = seed;
        g.row0 = row0;
        be.launch_optics(g);
        be.sync();
        if (!be.ok()) return fail(SSF_ERR_HIP, be.last_error());
        if ((void *)g.o0 != o0) be.d2h_big(o0, g.o0, sizeof(Cd) * n_o0);
        if (o1 && (void *)g.o1 != o1) be.d2h_big(o1, g.o1, sizeof(Cd) * (size_t)n);
        return be.ok() ? SSF_OK : fail(SSF_ERR_HIP, be.last_error());
    }

    // firFilter (core.py:87-125): 'same'-mode convolution of every column with the taps
    int fir(long long sigLen, int ncols, int ntaps, const void *taps, const void *in, void *out) {
        if (sigLen < 1 || ncols < 1 || ntaps < 1) return fail(SSF_ERR_BAD_ARG, "bad size");
        if (ntaps > kMaxNfft / 2) return fail(SSF_ERR_UNSUPPORTED, "firFilter: at most 4096 taps");
        const int nfft = fir_nfft(ntaps);
        const size_t n = (size_t)sigLen * ncols;
        const Cd *a = resident(in, n);
        Cd *b = result_buffer(out, in, n), *dH = taps_filter((const zc *)taps, ntaps, nfft);
        if (!a || !b || !dH) return fail(SSF_ERR_OOM, "out of device memory");
        int rc = ols(a, ncols, sigLen, sigLen, b, ncols, sigLen, ncols, dH, 0, ntaps, nfft, 0);
        if (rc) return rc;
        return finish(out, b, n);
    }

    // FIR of any length, on the device: out[n, m] = sum_t taps[t] * in[n + shift - t, m] for n in [0, outLen), `in` (inLen samples
    // per column) extended with zeros on both sides.  blockwiseFFTConv's result (optic/dsp/core.py:1043-1046) is shift =
    // (K - 1) / 2, outLen = inLen; delaySignal's pad / roll(-1) / cut (core.py:905-922) is the same with shift + 1.  The impulse
    // response is cut into segments h_p of at most kMaxNfft / 2 taps, conv(x, h)[k] = sum_p conv(x, h_p)[k - p0]: every segment is
    // one overlap-save launch that adds its part in place (OlsArgs::acc), so the cost is ceil(K / 4096) passes over the signal.
    int fir_long(long long inLen, long long outLen, int ncols, long long K, const void *taps, long long shift, const void *in,
                 void *out) {
        if (inLen < 1 || outLen < 1 || ncols < 1 || K < 1) return fail(SSF_ERR_BAD_ARG, "bad size");
        const Cd *a = resident(in, (size_t)inLen * ncols);
        Cd *acc = result_buffer(out, in, (size_t)outLen * ncols);
        if (!a || !acc) return fail(SSF_ERR_OOM, "out of device memory");
        be.memset(acc, 0, sizeof(Cd) * (size_t)outLen * ncols);
        const long long S = kMaxNfft / 2;
        for (long long p0 = 0; p0 < K; p0 += S) {
            const int Kp = (int)std::min(S, K - p0);
            const int nfft = K <= S ? fir_nfft(Kp) : kMaxNfft;
            const long long sh = shift - p0;                         // out[n] += full_p[n + sh]
            const long long d = nfft - Kp + 1;
            const long long f_lo = std::max(0ll, sh), f_hi = std::min(outLen + sh, inLen + Kp - 1);
            if (f_hi <= f_lo) continue;                              // this segment only meets the zero extension
            Cd *dH = upload_filter(ols_filter_from_taps((const zc *)taps + p0, Kp, nfft));
            if (!dH) return fail(SSF_ERR_OOM, "out of device memory");
            fused::OlsArgs<double> g{};
            g.in = a;
            g.out = acc;
            g.H = dH;
            g.sigLen = outLen;
            g.nrows = ncols;
            g.log2nfft = 0;
            while ((1 << g.log2nfft) < nfft) ++g.log2nfft;
            g.d = (int)d;
            g.discard = Kp - 1;
            g.D = 0;
            fused::ols_defaults(g);
            g.inLen = inLen;
            g.Dx = sh;
            g.blk0 = f_lo / d;
            g.njobs = ((f_hi + d - 1) / d - g.blk0) * ncols;
            g.acc = 1;
            be.launch_ols(g);
        }
        return finish(out, acc, (size_t)outLen * ncols);
    }

    // simpleWDMTx's signal path (tx.py:178-217) for all channels and polarisations; symbols (nCh, nPol, nSymbols),
    // taps (ntaps real), phi (nCh, N) -- (1, N) with p.phi_rows = 1 -- or null, amp[nCh] = sqrt(Pch / nPol), deltaF[nCh]; out (N, nPol), N = nSymbols * SpS
    int wdm_tx(const ssf_tx_params &p, const void *symbols, const double *taps, const double *phi, const double *amp,
               const double *deltaF, void *out, double *power_out) {
        const long long nS = p.nSymbols, N = nS * p.SpS;
        const int nCh = p.nChannels, nPol = p.nPolModes, K = p.ntaps;
        if (nS < 1 || p.SpS < 1 || nCh < 1 || nPol < 1 || K < 1 || !(p.Fs > 0)) return fail(SSF_ERR_BAD_ARG, "ssf_wdm_tx: bad size");
        if (K > kMaxNfft / 2) return fail(SSF_ERR_UNSUPPORTED, "ssf_wdm_tx: at most 4096 pulse-shaping taps");
        const int nfft = fir_nfft(K), nblocks = 256;
        std::vector<zc> tz((size_t)K);
        for (int i = 0; i < K; ++i) tz[(size_t)i] = zc(taps[i], 0.0);
        Cd *dH = upload_filter(ols_filter_from_taps(tz.data(), K, nfft));
        Cd *dsym = dalloc((size_t)nCh * nPol * nS), *sig = dalloc((size_t)N), *mod = dalloc((size_t)N), *acc = dalloc((size_t)N * nPol);
        double *dpart = (double *)be.alloc(sizeof(double) * nblocks), *dphi = nullptr;
        if (dpart) owned.push_back(dpart);
        if (!dH || !dsym || !sig || !mod || !acc || !dpart) return fail(SSF_ERR_OOM, "out of device memory");
        const bool dev_pn = !phi && p.pn_seed != 0 && p.pn_sigma > 0;          // random walk generated on the device
        const int pn_chunks = (int)((N + kPnChunk - 1) / kPnChunk);
        double *dcs = nullptr;
        std::vector<double> cs((size_t)pn_chunks);
        if (phi || dev_pn) {
            dphi = (double *)be.alloc(sizeof(double) * (size_t)N);
            if (dev_pn) dcs = (double *)be.alloc(sizeof(double) * (size_t)pn_chunks);
            if (dphi) owned.push_back(dphi);
            if (dcs) owned.push_back(dcs);
            if (!dphi || (dev_pn && !dcs)) return fail(SSF_ERR_OOM, "out of device memory");
        }
        be.h2d_big(dsym, symbols, sizeof(Cd) * (size_t)nCh * nPol * nS);
        be.memset(acc, 0, sizeof(Cd) * (size_t)N * nPol);
        std::vector<double> part((size_t)nblocks);
        const double erLin = std::pow(10.0, 60.0 / 10), gamma = 2 * std::sqrt(erLin) / (erLin + 1);   // devices.py:185-188 defaults
        for (int ch = 0; ch < nCh; ++ch) {
            if (phi && p.phi_rows != 0 && p.phi_rows != 1 && p.phi_rows != nCh) return fail(SSF_ERR_BAD_ARG, "ssf_wdm_tx: phi_rows is 0, 1 or nChannels");
            if (phi && (p.phi_rows != 1 || ch == 0)) be.h2d_big(dphi, phi + (p.phi_rows == 1 ? 0 : (size_t)ch * N), sizeof(double) * (size_t)N);
            if (dev_pn) {
                PnArgs pa{nullptr, dcs, N, p.pn_sigma, (unsigned long long)p.pn_seed, (unsigned)ch};
                be.launch_pn(pa, pn_chunks);
                be.sync();
                be.d2h(cs.data(), dcs, sizeof(double) * (size_t)pn_chunks);
                double run = 0;                                      // exclusive scan of the chunk sums (a few hundred values)
                for (int c = 0; c < pn_chunks; ++c) {
                    const double v = cs[(size_t)c];
                    cs[(size_t)c] = run;
                    run += v;
                }
                be.h2d(dcs, cs.data(), sizeof(double) * (size_t)pn_chunks);
                pa.phi = dphi;
                be.launch_pn(pa, pn_chunks);
            }
            for (int mode = 0; mode < nPol; ++mode) {
                const Cd *sym = dsym + ((size_t)ch * nPol + mode) * nS;
                int rc = ols(sym, 1, N, N, sig, 1, N, 1, dH, 0, K, nfft, 0, p.SpS);
                if (rc) return rc;
                AbsMaxArgs ma{sig, dpart, N};
                be.launch_absmax(ma, nblocks);
                be.sync();
                be.d2h(part.data(), dpart, sizeof(double) * nblocks);
                double mx = 0;
                for (double v : part) mx = v > mx ? v : mx;
                IqmArgs ia{};
                ia.sig = sig;
                ia.phi = dphi;
                ia.out = mod;
                ia.part = dpart;
                ia.N = N;
                ia.inv_max = 1.0 / mx;
                ia.mzmScale = p.mzmScale;
                ia.Vpi = 2;
                ia.VbI = ia.VbQ = -2;
                ia.sp = std::sqrt(1 + gamma);
                ia.sm = std::sqrt(1 - gamma);
                ia.rotQ = mk<double>(std::cos(kPi * 1 / 2), std::sin(kPi * 1 / 2));
                be.launch_iqm(ia, nblocks);
                be.sync();
                be.d2h(part.data(), dpart, sizeof(double) * nblocks);
                double pw = 0;
                for (double v : part) pw += v;
                const double mean = pw / (double)N;
                ShiftAddArgs sa{mod, acc, N, nPol, mode, 1.0 / std::sqrt(mean), amp[ch], 2 * kPi * deltaF[ch], 1.0 / p.Fs};
                be.launch_shift_add(sa);
                if (power_out) power_out[(size_t)ch * nPol + mode] = amp[ch] * amp[ch];   // signalPower(sqrt(P) pnorm(x)) = P
            }
        }
        be.sync();
        be.d2h_big(out, acc, sizeof(Cd) * (size_t)N * nPol);
        return be.ok() ? SSF_OK : fail(SSF_ERR_HIP, be.last_error());
    }

    // decimate (core.py:435-491); sampDelay_out[ncols] receives the chosen sampling phases
    int decimate(long long N, int ncols, int SpSin, int decFactor, const void *in, void *out, int *sampDelay_out) {
        if (N < 1 || ncols < 1 || ncols > 8 || SpSin < 1 || decFactor < 1) return fail(SSF_ERR_BAD_ARG, "decimate: 1 <= columns <= 8, SpSin >= 1, decFactor >= 1");
        if (N % SpSin) return fail(SSF_ERR_BAD_ARG, "cannot reshape array: length is not a multiple of SpSin");   // core.py:477
        const long long Nout = (N + decFactor - 1) / decFactor;
        const int nclass = SpSin * ncols;
        if (nclass > 256) return fail(SSF_ERR_UNSUPPORTED, "decimate: SpSin * columns <= 256");
        const int nthreads = 256 / nclass * nclass;                  // every thread keeps one (phase, column) class
        // the grid stride must keep the class too: nblocks * nthreads is a multiple of nclass by construction
        const int nblocks = (int)std::max<long long>(1, std::min<long long>(512, (N * ncols + 4 * nthreads - 1) / (4 * nthreads)));
        const Cd *a = resident(in, (size_t)N * ncols);
        Cd *b = result_buffer(out, in, (size_t)Nout * ncols);
        double *dpart = (double *)be.alloc(sizeof(double) * 3 * (size_t)nblocks * nclass);
        int *ddelay = (int *)be.alloc(sizeof(int) * 8);
        if (dpart) owned.push_back(dpart);
        if (ddelay) owned.push_back(ddelay);
        if (!a || !b || !dpart || !ddelay) return fail(SSF_ERR_OOM, "out of device memory");
        // per (phase, column) class sum x and sum |x|^2 in ONE pass (round 6: two passes -- mean, then variance about it -- and two
        // reductions before), the partials added up and the phases picked by one small launch: three launches back to back with the
        // gather, no host round trip in between.  var = sum |x|^2 / M - |sum x / M|^2: the phases' variances differ by far more than
        // the rounding of either form (np.var's two-pass form included)
        DecStatsArgs sa{a, dpart, N * ncols, nclass, ncols};
        be.launch_dec_stats(sa, nblocks, nthreads);
        ChainFinishArgs fa{dpart, ddelay, nblocks, nclass, ncols, SpSin, (double)(N / SpSin)};
        be.launch_chain_finish(fa);
        DecGatherArgs ga{};
        ga.in = a;
        ga.out = b;
        ga.N = N;
        ga.Nout = Nout;
        ga.ncols = ncols;
        ga.dec = decFactor;
        ga.delay = ddelay;
        be.launch_dec_gather(ga);
        if (sampDelay_out) {
            int dl[8];
            be.d2h(dl, ddelay, sizeof(int) * (size_t)ncols);         // (waits for the stream)
            for (int c = 0; c < ncols; ++c) sampDelay_out[c] = dl[c];
        }
        return finish(out, b, (size_t)Nout * ncols);
    }

    // blockwiseFFTConv with a caller-supplied frequency response (edc: ssf_overlap_save); Hfft = fft(zero-padded
    // impulse response), nfft values
    // the device image of a caller-supplied block response (edc): scaled by the ifft's 1 / NFFT and put in the kernels' register
    // order -- built only when the cache does not hold it (edc designs the same filter call after call; building it first cost
    // every call 10 - 20 us of host time in front of its first launch)
    Cd *response_filter(const void *Hfft, int K, int nfft) {
        const FilterKey key{4, K, nfft, 0, 0.0, 0.0, content_hash((const zc *)Hfft, (size_t)nfft), content_hash2((const zc *)Hfft, (size_t)nfft)};
        return cached_filter(key, [&] {
            std::vector<zc> H((size_t)nfft);
            for (int i = 0; i < nfft; ++i) H[(size_t)i] = ((const zc *)Hfft)[i] / (double)nfft;
            int lg = 0;
            while ((1 << lg) < nfft) ++lg;
            fused::ols_permute_filter(H.data(), lg);
            return H;
        });
    }
    int overlap_save(long long sigLen, int ncols, int nfft, int K, const void *Hfft, const void *in, void *out) {
        const size_t n = (size_t)sigLen * ncols;
        const Cd *a = resident(in, n);
        Cd *b = result_buffer(out, in, n), *dH = response_filter(Hfft, K, nfft);
        if (!a || !b || !dH) return fail(SSF_ERR_OOM, "out of device memory");
        int rc = ols(a, ncols, sigLen, sigLen, b, ncols, sigLen, ncols, dH, 0, K, nfft, 0);
        if (rc) return rc;
        return finish(out, b, n);
    }

    // delaySignal (core.py:880-922) of one column
    // channels.py:471-493: phi (n doubles) from Ex, Ey (n complex128 each) and the real part of Pch (n doubles)
    int nlin_phase(long long n, double gamma, const void *Ex, const void *Ey, const double *Pch, double *phi) {
        if (n < 1) return fail(SSF_ERR_BAD_ARG, "bad size");
        Cd *x = dalloc((size_t)n), *y = dalloc((size_t)n);
        double *pc = (double *)dalloc((size_t)(n + 1) / 2), *ph = (double *)dalloc((size_t)(n + 1) / 2);
        if (!x || !y || !pc || !ph) return fail(SSF_ERR_OOM, "out of device memory");
        be.h2d_big(x, Ex, sizeof(Cd) * (size_t)n);
        be.h2d_big(y, Ey, sizeof(Cd) * (size_t)n);
        be.h2d_big(pc, Pch, sizeof(double) * (size_t)n);
        NlinPhaseArgs a{x, y, pc, ph, n, (8.0 / 9.0) * gamma};
        be.launch_nlin_phase(a);
        be.sync();
        be.d2h_big(phi, ph, sizeof(double) * (size_t)n);
        return be.ok() ? SSF_OK : fail(SSF_ERR_HIP, be.last_error());
    }
    // channels.py:496-519: sqrt(|Ex_fd - Ex_conv|^2 + |Ey_fd - Ey_conv|^2) / sqrt(|Ex_conv|^2 + |Ey_conv|^2)
    int convergence(long long n, const void *xfd, const void *yfd, const void *xc, const void *yc, double *lim) {
        if (n < 1 || !lim) return fail(SSF_ERR_BAD_ARG, "bad size");
        const int nblocks = (int)std::min<long long>(512, (n + 255) / 256);
        Cd *d[4];
        for (auto &q : d) q = dalloc((size_t)n);
        double *dpart = (double *)dalloc((size_t)nblocks);          // 2 * nblocks doubles
        if (!d[0] || !d[1] || !d[2] || !d[3] || !dpart) return fail(SSF_ERR_OOM, "out of device memory");
        const void *src[4] = {xfd, yfd, xc, yc};
        for (int i = 0; i < 4; ++i) be.h2d_big(d[i], src[i], sizeof(Cd) * (size_t)n);
        ConvSumsArgs a{d[0], d[1], d[2], d[3], dpart, dpart + nblocks, n};
        be.launch_conv_sums(a, nblocks);
        be.sync();
        std::vector<double> part((size_t)2 * nblocks);
        be.d2h(part.data(), dpart, sizeof(double) * 2 * nblocks);
        if (!be.ok()) return fail(SSF_ERR_HIP, be.last_error());
        double num = 0, den = 0;                                    // fixed order
        for (int i = 0; i < nblocks; ++i) {
            num += part[i];
            den += part[(size_t)nblocks + i];
        }
        *lim = std::sqrt(num) / std::sqrt(den);
        return SSF_OK;
    }
    int delay(long long N, double delay_s, double Fs, const void *in, void *out) {
        if (N < 1 || !(Fs > 0)) return fail(SSF_ERR_BAD_ARG, "bad size");
        const Cd *a = resident(in, (size_t)N);
        Cd *b = result_buffer(out, in, (size_t)N);
        if (!a || !b) return fail(SSF_ERR_OOM, "out of device memory");
        const double dl[1] = {delay_s};
        int rc = delay_pair(a, b, 1, N, dl, 1, Fs);
        if (rc) return rc;
        return finish(out, b, (size_t)N);
    }
};

}  // namespace rx
}  // namespace ssf
