// Device bodies of the coherent receiver front-end.  Same convention as
// fused_kernels.h: every body is a template over a Ctx (HIP: DevCtx, tests: the CPU fiber
// emulator), so the code that runs on the GPU is the code the CPU tests execute.
//
// Reference (file:line under the reference checkout):
//   pbs                  optic/models/devices.py:223-260
//   opticalHybrid2x4     optic/models/devices.py:462-500
//   photodiode           optic/models/devices.py:289-399
//   balancedPD           optic/models/devices.py:402-459
//   coherentReceiver     optic/models/devices.py:503-571
//   pdmCoherentReceiver  optic/models/devices.py:574-668
//   iqMixing             optic/dsp/core.py:925-970
// The FIR / fractional-delay filters between these stages are overlap-save launches (ols_body).
#pragma once
#include "fused_kernels.h"
#include "ssf_rng.h"

namespace ssf {
namespace rx {

using fused::cx;
using fused::mk;
typedef cx<double> Cd;

// element i (0 or 1) of a two-entry kernel-argument array: a run-time subscript would move the whole argument block into scratch
template <typename V> SSF_HD V sel2(const V (&a)[2], int i) { return i ? a[1] : a[0]; }

// ---- photodiode model shared by every detection mode (devices.py:352-399, without the filter)
struct PdModel {
    double R, IpdSat;
    double shot_k;       // Fs * q:            Is = sqrt(shot_k * (ipd + Id)) * u    (devices.py:381-383)
    double Id;
    double thermal_sigma;// sqrt(Fs * 2 kB T / RL)                                   (devices.py:387-390)
    int saturate, shot, thermal;   // all 0 for the ideal model
    unsigned long long seed;       // device noise (Philox), used when `un` is null
    const double *un;    // host-supplied unit normals, [(pd * 2 + kind) * N + n], or null
};
// unit normals (shot, thermal) of `cnt` consecutive photodiodes starting at pd0 (even) at sample n: z[2 k] / z[2 k + 1] for
// photodiode pd0 + k.  Supplied by the host ([(pd * 2 + kind) * N + n]: seeded parity runs) or generated, two photodiodes per call.
SSF_HD void pd_normals(const PdModel &m, long long n, long long N, int pd0, int cnt, double *z) {
    if (m.un) {
        for (int k = 0; k < 2 * cnt; ++k) z[k] = m.un[(size_t)(pd0 * 2 + k) * N + n];
    } else {
        for (int k = 0; k < cnt; k += 2) {
            float g[4];
            gauss_quad((unsigned long long)n, (unsigned)((pd0 + k) >> 1), 0x5044u, m.seed, g);
            z[2 * k] = g[0];
            z[2 * k + 1] = g[1];
            if (k + 1 < cnt) {
                z[2 * k + 2] = g[2];
                z[2 * k + 3] = g[3];
            }
        }
    }
}
// photocurrent from the optical power and the photodiode's two unit normals (devices.py:367-390)
SSF_HD double pd_current_pw(const PdModel &m, double pw, double us, double ut) {
    double i = m.R * pw;
    if (m.saturate && i > m.IpdSat) i = m.IpdSat;
    if (m.shot) i += sqrt(m.shot_k * (i + m.Id)) * us;
    if (m.thermal) i += m.thermal_sigma * ut;
    return i;
}
SSF_HD double pd_current(const PdModel &m, Cd e, double us, double ut) { return pd_current_pw(m, e.re * e.re + e.im * e.im, us, ut); }

enum { RX_PHOTODIODE = 0, RX_BALANCED = 1 };

// ---- detection stage of photodiode / balancedPD: the real photocurrent before the photodiodes' low-pass filter
//   RX_PHOTODIODE  in0 = (N, nm) field: R * sum_modes |E|^2 (+ noise)             (devices.py:352-399)
//   RX_BALANCED    in0 = (N, 2): columns E1, E2: i1 - i2                           (devices.py:402-459)
// Subtracting before the (linear, common) filter instead of after it is exact.  (The coherent receivers: det_sample below.)
struct PdFront {
    const Cd *in0;
    long long N;
    int mode, nm;
    PdModel pd;
};
template <int NOISE = 2> SSF_HD double pd_front_sample(const PdFront &a, long long n) {
    const bool noisy = NOISE != 0 && (a.pd.shot || a.pd.thermal);
    double z[4] = {0, 0, 0, 0};
    if (a.mode == RX_PHOTODIODE) {
        if (noisy) pd_normals(a.pd, n, a.N, 0, 1, z);
        double pw = 0;                                  // devices.py:356-359: one photocurrent from the summed mode powers
        for (int k = 0; k < a.nm; ++k) {
            const Cd e = a.in0[n * a.nm + k];
            pw += e.re * e.re + e.im * e.im;
        }
        return pd_current_pw(a.pd, pw, z[0], z[1]);
    }
    if (noisy) pd_normals(a.pd, n, a.N, 0, 2, z);
    return pd_current(a.pd, a.in0[2 * n], z[0], z[1]) - pd_current(a.pd, a.in0[2 * n + 1], z[2], z[3]);
}
// without a filter behind it (ideal photodiodes / bandwidthLimitation off): one element-wise pass, (N,) float64 out
struct FrontArgs {
    PdFront f;
    double *out;
};
template <int NOISE, class Ctx> SSF_HD void front_loop(Ctx &ctx, const FrontArgs &a) {
    for (long long n = (long long)ctx.bid * ctx.nthreads + ctx.tid; n < a.f.N; n += (long long)ctx.nblocks * ctx.nthreads)
        a.out[n] = pd_front_sample<NOISE>(a.f, n);
}
template <class Ctx> SSF_HD void front_body(Ctx &ctx, const FrontArgs &a) {
    if (a.f.pd.shot || a.f.pd.thermal) front_loop<1>(ctx, a);
    else front_loop<0>(ctx, a);
}

// =====================================================================================================
// The coherent receivers as at most three launches (round 5).  Every stage that is not a filter rides in the loads or the stores
// of the filter next to it (fused_kernels.h: ols_body_x) -- or, where no filter follows, in one element-wise kernel:
//   [PBS rotation -> polarisation delay filters]                 rx_ols_body, PRE_PBS                       (paramFE.polDelay != 0)
//   [PBS, PDL, LO split,] hybrid, photodiodes, balanced pairs -> [low-pass FIR] -> IQ imbalance
//                                                                rx_ols_body, PRE_DET (+ POST_IQF), or det_body without a filter
//   IQ imbalance -> skew filters of I and Q -> S = I + j Q        rx_ols_body, PRE_IQ + POST_PART            (a timeSkew != 0)
// Before: pbs, delay filter, front, low-pass filter, iqmix, two skew filters, combine -- eight launches and as many passes.
// photodiode / balancedPD the same way: pd_front_sample in the low-pass filter's loads, the real part in its stores
//                                                                rx_ols_body, PRE_PD + POST_REAL, or front_body without a filter
struct DetArgs {
    const Cd *in0;            // (N, nm) signal: before the PBS when `pbs` is set, behind it (and the delay filters) otherwise
    const Cd *lo;             // (N,)
    long long N;
    int nm;                   // 1: coherentReceiver, 2: pdmCoherentReceiver
    int pbs;                  // rotate the two columns here, E @ [[c, -s], [s, c]] (devices.py:223-260)
    double c, s;
    double es_scale[2];       // PDL (devices.py:660-662)
    double lo_scale[2];       // LO split by the PBS at pi / 4 (devices.py:653): cos, -sin; (1, 1) for one polarisation
    PdModel pd;
};
// the detected sample sI + j sQ of polarisation p at time n (devices.py:487-499, 562-563; photodiode slots as front_body) from the
// signal sample `es` behind the PBS (and the polarisation delay filters)
template <int NOISE = 2> SSF_HD Cd det_core(const DetArgs &a, long long n, int p, Cd es, Cd lo) {
    const double ks = sel2(a.es_scale, p), kl = sel2(a.lo_scale, p);
    es = mk<double>(es.re * ks, es.im * ks);
    lo = mk<double>(lo.re * kl, lo.im * kl);
    const Cd e0 = mk<double>(0.5 * es.re - 0.5 * lo.re, 0.5 * es.im - 0.5 * lo.im);      //  Es/2 -  Elo/2
    const Cd e1 = mk<double>(-0.5 * es.im - 0.5 * lo.im, 0.5 * es.re + 0.5 * lo.re);     // jEs/2 + jElo/2
    const Cd e2 = mk<double>(-0.5 * es.im - 0.5 * lo.re, 0.5 * es.re - 0.5 * lo.im);     // jEs/2 -  Elo/2
    const Cd e3 = mk<double>(-0.5 * es.re - 0.5 * lo.im, -0.5 * es.im + 0.5 * lo.re);    // -Es/2 + jElo/2
    double z[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // NOISE = 0: the caller knows that neither noise source is on (the generator
    if (NOISE != 0 && (a.pd.shot || a.pd.thermal)) pd_normals(a.pd, n, a.N, 4 * p, 4, z);   // is not compiled into those kernels)
    const double sI = pd_current(a.pd, e1, z[0], z[1]) - pd_current(a.pd, e0, z[2], z[3]);
    const double sQ = pd_current(a.pd, e2, z[4], z[5]) - pd_current(a.pd, e3, z[6], z[7]);
    return mk<double>(sI, sQ);
}
template <int NOISE = 2> SSF_HD Cd det_sample(const DetArgs &a, long long n, int p) {
    Cd es;
    if (a.pbs) {
        const Cd e0 = a.in0[2 * n], e1 = a.in0[2 * n + 1];
        es = p == 0 ? mk<double>(e0.re * a.c + e1.re * a.s, e0.im * a.c + e1.im * a.s)
                    : mk<double>(e1.re * a.c - e0.re * a.s, e1.im * a.c - e0.im * a.s);
    } else es = a.in0[n * a.nm + p];
    return det_core<NOISE>(a, n, p, es, a.lo[n]);
}
// IQ imbalance (core.py:952-960): s' = k1 s + k2 conj(s)
SSF_HD Cd iq_mix(Cd k1, Cd k2, Cd s) { return k1 * s + k2 * fused::conj(s); }

enum { PRE_PLAIN = 0, PRE_PBS = 1, PRE_DET = 2, PRE_IQ = 3, PRE_PD = 4, PRE_PBS_DET = 5 };   // (5: PRE_PBS loads, POST_DET stores -- compile-time)
enum { POST_PLAIN = 0, POST_IQF = 1, POST_PART = 2, POST_REAL = 3 };
struct RxOlsArgs {
    fused::OlsArgs<double> o; // geometry, filters, o.in (PRE_PLAIN / PRE_IQ), o.out
    int pre, post;
    DetArgs det;              // PRE_PBS: in0, c, s;  PRE_DET: everything
    PdFront front;            // PRE_PD: photodiode / balancedPD (one column: the photocurrent as a real signal; POST_REAL: o.out is (N,) float64)
    Cd k1[2], k2[2];          // PRE_IQ / POST_IQF, per polarisation
    int nm;                   // PRE_IQ / POST_PART: polarisations of the signal (columns of o.in / o.out)
    int pol0;                 // ... the launch's columns are I and Q of the polarisations pol0, pol0 + 1, ...
    long long N;              // POST_IQF: signal length -- the last sample is zero, as delaySignal's np.roll(-1) of an unpadded
                              // signal leaves it when the skew is zero (core.py:905-922: y[N - 1] = conv[0] = x[-1] = 0)
};
// PRE (and, for the detection, whether the photodiodes are noisy) are compile-time: the load loop of ols_body_x carries sixteen
// copies of the stage, and with every stage's code in every kernel the compiler gave up unrolling it (the value array then
// lived in scratch memory)
template <int LG, int C, int PRE, int NOISE, class Ctx> SSF_HD void rx_ols_body(Ctx &ctx, const RxOlsArgs &a) {
    fused::ols_body_x<double, LG, C>(
        ctx, a.o,
        [&](long long src, int m) -> Cd {
            if constexpr (PRE == PRE_PBS || PRE == PRE_PBS_DET) {
                const Cd e0 = a.det.in0[2 * src], e1 = a.det.in0[2 * src + 1];
                return m == 0 ? mk<double>(e0.re * a.det.c + e1.re * a.det.s, e0.im * a.det.c + e1.im * a.det.s)
                              : mk<double>(e1.re * a.det.c - e0.re * a.det.s, e1.im * a.det.c - e0.im * a.det.s);
            } else if constexpr (PRE == PRE_DET) {
                return det_sample<NOISE>(a.det, src, m);
            } else if constexpr (PRE == PRE_PD) {
                return mk<double>(pd_front_sample<NOISE>(a.front, src), 0.0);
            } else if constexpr (PRE == PRE_IQ) {        // column m = 2 p + part: the real (I) or imaginary (Q) part of s'_p, as a real signal
                const int pl = a.pol0 + (m >> 1);
                const Cd t = iq_mix(sel2(a.k1, pl), sel2(a.k2, pl), a.o.in[src * a.nm + pl]);
                return mk<double>((m & 1) ? t.im : t.re, 0.0);
            } else return a.o.in[src * a.o.in_ld + m];
        },
        [&](long long n, int m, Cd v) {
            if constexpr (PRE == PRE_PBS_DET) {          // the polarisation delay filter's stores detect: hybrid, ideal photodiodes, IQ
                const Cd t = det_core<0>(a.det, n, m, v, a.det.lo[n]);   // imbalance and the zero-skew rule (det_loop) -- no pass of their own
                a.o.out[n * a.o.out_ld + m] = n == a.N - 1 ? mk<double>(0.0, 0.0) : iq_mix(sel2(a.k1, m), sel2(a.k2, m), t);
                return;                                  // (its own instantiation: 54.0 -> 52.5 us without the other stores' code beside it; the
            }                                            //  LO samples fetched ahead of the store loop -- 64 more registers -- 68 us: round 6)
            if (a.post == POST_IQF) {
                const Cd t = iq_mix(sel2(a.k1, m), sel2(a.k2, m), v);
                a.o.out[n * a.o.out_ld + m] = n == a.N - 1 ? mk<double>(0.0, 0.0) : t;
            } else if (a.post == POST_PART) {            // S_p = Re(filtered I) + j Re(filtered Q)   (core.py:963-968)
                double *o = (double *)(a.o.out + n * a.nm + a.pol0 + (m >> 1));
                o[m & 1] = v.re;
            } else if (a.post == POST_REAL) {            // `return ipd.real` (devices.py:399)
                ((double *)a.o.out)[n] = v.re;
            } else a.o.out[n * a.o.out_ld + m] = v;
        });
}
// f(LG, C, PRE, NOISE) -- integral constants -- for the instantiation a fused launch needs; false: no such kernel (the pipeline
// only asks for the delay-filter geometry, kDelayNfft points and two columns at a time, around PRE_PBS / PRE_IQ)
template <class F> inline bool rx_ols_dispatch(const RxOlsArgs &a, const fused::OlsLaunch &o, F &&f) {
    using std::integral_constant;
    if (o.lg == 0) return false;
    if (a.pre == PRE_DET) {
        const bool noisy = a.det.pd.shot || a.det.pd.thermal;
        fused::ols_dispatch(o, [&](auto lg, auto cc) {
            if (noisy) f(lg, cc, integral_constant<int, PRE_DET>{}, integral_constant<int, 1>{});
            else f(lg, cc, integral_constant<int, PRE_DET>{}, integral_constant<int, 0>{});
        });
        return true;
    }
    if (a.pre == PRE_PD) {                           // (one column)
        if (o.C != 1) return false;
        const bool noisy = a.front.pd.shot || a.front.pd.thermal;
        fused::ols_dispatch(o, [&](auto lg, auto) {
            if (noisy) f(lg, integral_constant<int, 1>{}, integral_constant<int, PRE_PD>{}, integral_constant<int, 1>{});
            else f(lg, integral_constant<int, 1>{}, integral_constant<int, PRE_PD>{}, integral_constant<int, 0>{});
        });
        return true;
    }
    if ((o.lg != 11 && o.lg != 12) || (a.pre != PRE_PBS && a.pre != PRE_IQ && a.pre != PRE_PBS_DET)) return false;
    auto stage = [&](auto lg, auto cc) {
        if (a.pre == PRE_PBS) f(lg, cc, integral_constant<int, PRE_PBS>{}, integral_constant<int, 0>{});
        else if (a.pre == PRE_PBS_DET) f(lg, cc, integral_constant<int, PRE_PBS_DET>{}, integral_constant<int, 0>{});
        else f(lg, cc, integral_constant<int, PRE_IQ>{}, integral_constant<int, 0>{});
    };
    if (o.C == 2) {
        if (o.lg == 11) stage(integral_constant<int, 11>{}, integral_constant<int, 2>{});
        else stage(integral_constant<int, 12>{}, integral_constant<int, 2>{});
        return true;
    }
    return false;
}
// detection without a filter behind it (ideal photodiodes / bandwidthLimitation off): one element-wise pass
struct DetKernelArgs {
    DetArgs det;
    Cd *out;                  // (N, nm)
    int iqf;                  // apply the IQ imbalance and the zero-skew rule here (nothing follows)
    Cd k1[2], k2[2];
};
template <int NOISE, class Ctx> SSF_HD void det_loop(Ctx &ctx, const DetKernelArgs &a) {
    const long long total = a.det.N * a.det.nm;
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < total; i += (long long)ctx.nblocks * ctx.nthreads) {
        const long long n = a.det.nm == 2 ? i >> 1 : i;
        const int p = (int)(i - n * a.det.nm);
        Cd v = det_sample<NOISE>(a.det, n, p);
        if (a.iqf) v = n == a.det.N - 1 ? mk<double>(0.0, 0.0) : iq_mix(sel2(a.k1, p), sel2(a.k2, p), v);
        a.out[i] = v;
    }
}
template <class Ctx> SSF_HD void det_body(Ctx &ctx, const DetKernelArgs &a) {
    if (a.det.pd.shot || a.det.pd.thermal) det_loop<1>(ctx, a);       // (two loops: the quiet one carries no generator code)
    else det_loop<0>(ctx, a);
}
// iqMixing by itself with no skew: S = k1 s + k2 conj(s), last sample zero
struct IqfArgs {
    const Cd *in;
    Cd *out;
    long long N;
    int nm;
    Cd k1[2], k2[2];
};
template <class Ctx> SSF_HD void iqf_body(Ctx &ctx, const IqfArgs &a) {
    const long long total = a.N * a.nm;
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < total; i += (long long)ctx.nblocks * ctx.nthreads) {
        const long long n = i / a.nm;
        const int p = (int)(i - n * a.nm);
        a.out[i] = n == a.N - 1 ? mk<double>(0.0, 0.0) : iq_mix(sel2(a.k1, p), sel2(a.k2, p), a.in[i]);
    }
}

// ---- y += alpha x on float64 arrays (balancedPD's i1 - i2 when the photocurrents are device arrays: devices.py:456-458)
struct AxpyArgs {
    const double *x;
    double *y;
    long long n;
    double alpha;
};
template <class Ctx> SSF_HD void axpy_body(Ctx &ctx, const AxpyArgs &a) {
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < a.n; i += (long long)ctx.nblocks * ctx.nthreads) a.y[i] += a.alpha * a.x[i];
}

// =====================================================================================================
// The two helpers of the Manakov step that the reference also exports on their own (SURVEY.md 8a rows 3, 4;
// optic/models/channels.py:471-493 and 496-519, cupy twins optic/models/modelsGPU.py:514-561).  Inside
// manakovSSF / manakovDBP they are fused into the column kernel (fused_kernels.h: mk_advance); these stand-alone
// kernels serve callers that use the functions by themselves.
struct NlinPhaseArgs {      // phi = ((8/9) gamma (Pch + Ex conj(Ex) + Ey conj(Ey)) / 2).real
    const Cd *Ex, *Ey;
    const double *Pch;
    double *phi;
    long long n;
    double c8g;             // (8/9) gamma
};
template <class Ctx> SSF_HD void nlin_phase_body(Ctx &ctx, const NlinPhaseArgs &a) {
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < a.n; i += (long long)ctx.nblocks * ctx.nthreads) {
        const double px = a.Ex[i].re * a.Ex[i].re + a.Ex[i].im * a.Ex[i].im;
        const double py = a.Ey[i].re * a.Ey[i].re + a.Ey[i].im * a.Ey[i].im;
        a.phi[i] = a.c8g * (a.Pch[i] + px + py) / 2;
    }
}
struct ConvSumsArgs {       // block partials of |Ex_fd - Ex_conv|^2 + |Ey_fd - Ey_conv|^2 and |Ex_conv|^2 + |Ey_conv|^2
    const Cd *xfd, *yfd, *xc, *yc;
    double *pnum, *pden;    // nblocks each
    long long n;
};
template <class Ctx> SSF_HD void conv_sums_body(Ctx &ctx, const ConvSumsArgs &a) {
    double num = 0, den = 0;
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < a.n; i += (long long)ctx.nblocks * ctx.nthreads) {
        const double dxr = a.xfd[i].re - a.xc[i].re, dxi = a.xfd[i].im - a.xc[i].im;
        const double dyr = a.yfd[i].re - a.yc[i].re, dyi = a.yfd[i].im - a.yc[i].im;
        num += dxr * dxr + dxi * dxi + dyr * dyr + dyi * dyi;
        den += a.xc[i].re * a.xc[i].re + a.xc[i].im * a.xc[i].im + a.yc[i].re * a.yc[i].re + a.yc[i].im * a.yc[i].im;
    }
    fused::block_sum2(ctx, num, den, (double *)ctx.lds);
    if (ctx.tid == 0) {
        a.pnum[ctx.bid] = num;
        a.pden[ctx.bid] = den;
    }
}

// =====================================================================================================
// The amplifier and the passive optics the reference also exports by themselves, as element-wise device passes (inside
// ssfm / manakovSSF the amplifier is the span epilogue, fused_kernels.h: amp_body; inside the coherent receivers the splitter and
// the hybrid ride in the loads of the filters, det_sample above):
//   OPT_EDFA    out = in sqrt(G) + noise         optic/models/devices.py:671-726  (noise: the caller's array -- the reference's seeded
//               np.random draws -- or CN(0, 2 sigma^2) from Philox4x32-10: sample = n, row = row0 + column)
//   OPT_PBS     (Ex, Ey) = [ex, ey] @ [[c, -s], [s, c]]   optic/models/devices.py:223-260  (a one-column input is [ex, 0])
//   OPT_HYBRID  T @ [Es, 0, 0, Elo], T the 2x4 90-degree hybrid's 4 x 4 matrix   optic/models/devices.py:462-500  -> (4, n)
enum { OPT_EDFA = 0, OPT_PBS = 1, OPT_HYBRID = 2 };      // (= ssf_internal.h: kOptEdfa ...)
struct OpticsArgs {
    int op, ncols;
    const Cd *a, *b;        // EDFA: field, noise (or null); PBS: field (n, ncols); HYBRID: Es, Elo
    Cd *o0, *o1;            // EDFA: out; PBS: Ex, Ey; HYBRID: (4, n)
    long long n;            // EDFA: elements (samples x columns); PBS / HYBRID: samples
    double p0, p1;          // EDFA: sqrt(G_lin), sigma per quadrature (0: no device noise); PBS: cos, sin
    unsigned long long seed;
    unsigned row0;
};
template <class Ctx> SSF_HD void optics_body(Ctx &ctx, const OpticsArgs &a) {
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < a.n; i += (long long)ctx.nblocks * ctx.nthreads) {
        if (a.op == OPT_EDFA) {
            Cd e = mk<double>(a.a[i].re * a.p0, a.a[i].im * a.p0);
            if (a.b) e = e + a.b[i];
            if (a.p1 > 0) {
                double re, im;
                gauss_pair((unsigned long long)(i / a.ncols), a.row0 + (unsigned)(i % a.ncols), 0u, a.seed, a.p1, re, im);
                e = e + mk<double>(re, im);
            }
            a.o0[i] = e;
        } else if (a.op == OPT_PBS) {
            const Cd ex = a.ncols == 2 ? a.a[2 * i] : a.a[i];
            const Cd ey = a.ncols == 2 ? a.a[2 * i + 1] : mk<double>(0.0, 0.0);
            a.o0[i] = mk<double>(ex.re * a.p0 + ey.re * a.p1, ex.im * a.p0 + ey.im * a.p1);
            a.o1[i] = mk<double>(ey.re * a.p0 - ex.re * a.p1, ey.im * a.p0 - ex.im * a.p1);
        } else {
            const Cd s = mk<double>(0.5 * a.a[i].re, 0.5 * a.a[i].im), l = mk<double>(0.5 * a.b[i].re, 0.5 * a.b[i].im);
            const Cd js = mk<double>(-s.im, s.re), jl = mk<double>(-l.im, l.re);
            a.o0[i] = s - l;                        //  Es/2        - Elo/2
            a.o0[a.n + i] = js + jl;                // j Es/2       + j Elo/2
            a.o0[2 * a.n + i] = js - l;             // j Es/2       - Elo/2
            a.o0[3 * a.n + i] = jl - s;             // - Es/2       + j Elo/2
        }
    }
}

// =====================================================================================================
// WDM transmitter (optic/models/tx.py:42-228): after the pulse-shaping filter
// (an overlap-save launch on the zero-stuffed symbols) each (channel, polarisation) needs
//   max |x|                       -> absmax_body      sigTx / np.max(np.abs(sigTx))        (tx.py:204)
//   IQ modulator + mean power     -> iqm_body         iqm(sigLO, mzmScale * sigTx)         (tx.py:211-214)
//   normalise, shift, accumulate  -> shift_add_body   sqrt(P) * pnorm(.), freqShift, +=    (tx.py:215-217)
// Block partials are reduced on the host (a few hundred doubles, fixed order).
struct AbsMaxArgs {
    const Cd *in;
    double *part;      // nblocks
    long long N;
};
template <class Ctx> SSF_HD void absmax_body(Ctx &ctx, const AbsMaxArgs &a) {
    double m = 0;
    for (long long n = (long long)ctx.bid * ctx.nthreads + ctx.tid; n < a.N; n += (long long)ctx.nblocks * ctx.nthreads) {
        const double v = a.in[n].re * a.in[n].re + a.in[n].im * a.in[n].im;
        m = v > m ? v : m;
    }
    m = fused::block_max(ctx, m, (double *)ctx.lds);
    if (ctx.tid == 0) a.part[ctx.bid] = sqrt(m);
}

// IQ modulator with the reference's default bias / extinction (devices.py:147-220 -> core.py:1076-1140):
//   MZM(E, u, Vb) = sqrt(1 + g) E/2 cis(pi (u + Vb) / (2 Vpi)) + sqrt(1 - g) E/2 cis(-pi (u + Vb) / (2 Vpi))
//   Eo = MZM(Ei / sqrt2, Re u, VbI) + MZM(Ei / sqrt2, Im u, VbQ) cis(pi Vphi / Vpi)
struct IqmArgs {
    const Cd *sig;       // pulse-shaped signal
    const double *phi;   // LO phase noise (N) or null: Ei = exp(j phi)
    Cd *out;
    double *part;        // nblocks partial sums of |Eo|^2
    long long N;
    double u_scale;      // mzmScale / max|sig|   (applied as (sig / max) * mzmScale, see body)
    double inv_max, mzmScale;
    double Vpi, VbI, VbQ, sp, sm;   // sp = sqrt(1 + gamma), sm = sqrt(1 - gamma)
    Cd rotQ;             // cis(pi Vphi / Vpi)
};
SSF_HD Cd mzm_out(Cd e, double u, double Vb, double Vpi, double sp, double sm) {
    const double ang = ((u + Vb) / 2 / Vpi) * 3.14159265358979323846;
    double c, s;
    fused::cis_rad_d(ang, c, s);                            // (the lower arm's phase is the negative: the conjugate)
    const Cd h = mk<double>(e.re / 2, e.im / 2);
    const Cd a = h * mk<double>(c, s), b = h * mk<double>(c, -s);
    return mk<double>(sp * a.re + sm * b.re, sp * a.im + sm * b.im);
}
template <class Ctx> SSF_HD void iqm_body(Ctx &ctx, const IqmArgs &a) {
    double acc = 0, unused = 0;
    for (long long n = (long long)ctx.bid * ctx.nthreads + ctx.tid; n < a.N; n += (long long)ctx.nblocks * ctx.nthreads) {
        Cd ei = mk<double>(1.0, 0.0);
        if (a.phi) {
            double c, s;
            fused::sincos_d(a.phi[n], s, c);                 // a random walk: arbitrary magnitude, full-range sincos
            ei = mk<double>(c, s);
        }
        const Cd u = mk<double>(a.mzmScale * (a.sig[n].re * a.inv_max), a.mzmScale * (a.sig[n].im * a.inv_max));
        const Cd e = mk<double>(ei.re / 1.4142135623730951, ei.im / 1.4142135623730951);
        const Cd eI = mzm_out(e, u.re, a.VbI, a.Vpi, a.sp, a.sm);
        const Cd eQ = mzm_out(e, u.im, a.VbQ, a.Vpi, a.sp, a.sm);
        const Cd eo = eI + eQ * a.rotQ;
        a.out[n] = eo;
        acc += eo.re * eo.re + eo.im * eo.im;
    }
    fused::block_sum2(ctx, acc, unused, (double *)ctx.lds);
    if (ctx.tid == 0) a.part[ctx.bid] = acc;
}

// ---- laser phase noise on the device (ssf_tx_params::pn_seed): phi[0] = 0, phi[k] = phi[k - 1] + sigma * g_k, g_k ~ N(0, 1) from
// Philox (counter = k / 2, row = channel: two normals per draw).  A chunk of kPnChunk samples per workgroup; pass 1 leaves every
// chunk's sum, the host turns the (few hundred) sums into offsets, pass 2 writes offset + prefix.
constexpr int kPnChunk = 4096;
struct PnArgs {
    double *phi;              // (N,) pass 2; null in pass 1
    double *csum;             // pass 1: the chunks' sums out;  pass 2: the chunks' offsets in
    long long N;
    double sigma;
    unsigned long long seed;
    unsigned channel;
};
SSF_HD double pn_increment(const PnArgs &a, long long k) {      // the step INTO sample k (k >= 1)
    double g0, g1;
    gauss_pair((unsigned long long)(k >> 1), a.channel, 0x504Eu, a.seed, 1.0, g0, g1);
    return a.sigma * ((k & 1) ? g1 : g0);
}
template <class Ctx> SSF_HD void pn_body(Ctx &ctx, const PnArgs &a) {
    const int per = kPnChunk / ctx.nthreads;                     // consecutive samples per thread
    const long long k0 = (long long)ctx.bid * kPnChunk + (long long)ctx.tid * per;
    double loc = 0;
    for (int j = 0; j < per; ++j) {
        const long long k = k0 + j;
        if (k >= 1 && k < a.N) loc += pn_increment(a, k);
    }
    double *sh = (double *)ctx.lds;
    sh[ctx.tid] = loc;
    ctx.sync();
    if (!a.phi) {                                                // pass 1: the chunk's sum, in thread order (as pass 2 adds them up)
        if (ctx.tid == 0) {
            double s = 0;
            for (int t = 0; t < ctx.nthreads; ++t) s += sh[t];
            a.csum[ctx.bid] = s;
        }
        return;
    }
    double run = a.csum[ctx.bid];                                // pass 2: offset of the chunk + the threads before this one
    for (int t = 0; t < ctx.tid; ++t) run += sh[t];
    for (int j = 0; j < per; ++j) {
        const long long k = k0 + j;
        if (k >= a.N) break;
        if (k >= 1) run += pn_increment(a, k);
        a.phi[k] = run;
    }
}

// acc[n, mode] += amp * (x[n] / rms) * exp(j w t_n),  t_n = n * (1 / Fs), w = 2 pi deltaF  (core.py:1050-1073)
struct ShiftAddArgs {
    const Cd *in;
    Cd *acc;             // (N, npol)
    long long N;
    int npol, mode;
    double inv_rms, amp, w, Ts;
};
template <class Ctx> SSF_HD void shift_add_body(Ctx &ctx, const ShiftAddArgs &a) {
    for (long long n = (long long)ctx.bid * ctx.nthreads + ctx.tid; n < a.N; n += (long long)ctx.nblocks * ctx.nthreads) {
        const double t = (double)n * a.Ts;
        double c, s;
        fused::sincos_d(a.w * t, s, c);                      // |w t| reaches 1e6 rad: the argument is exact as the
        const Cd x = mk<double>(a.amp * (a.in[n].re * a.inv_rms), a.amp * (a.in[n].im * a.inv_rms));   // reference rounds it
        const Cd y = x * mk<double>(c, s);
        Cd &d = a.acc[n * a.npol + a.mode];
        d = mk<double>(d.re + y.re, d.im + y.im);
    }
}

// ---- decimate (optic/dsp/core.py:435-491)
// np.var(x[ph::sps, col]) for every sampling phase of every column, in two passes like numpy
// (mean, then mean |x - mean|^2).  The (N, ncols) array is read as one flat stream: with a thread
// count that is a multiple of nclass = sps * ncols every thread only ever sees one (phase, column)
// class, so the per-class sums are plain register accumulations; each workgroup leaves its partial
// sums (fixed order, deterministic) and dec_finish_body adds the few hundred partials.
struct DecSumArgs {
    const Cd *in;        // (N, ncols)
    const Cd *mean;      // nclass means for pass 2, null for pass 1
    double *part;        // (nblocks, nclass, 2): pass 1 sum re / sum im, pass 2 sum |x - mean|^2 / unused
    long long total;     // N * ncols
    int nclass;
};
template <class Ctx> SSF_HD void dec_sum_body(Ctx &ctx, const DecSumArgs &a) {
    double *red = (double *)ctx.lds;               // nthreads x 2 doubles
    const int cls = ctx.tid % a.nclass;
    const Cd m = a.mean ? a.mean[cls] : mk<double>(0.0, 0.0);
    double s0 = 0, s1 = 0;
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < a.total; i += (long long)ctx.nblocks * ctx.nthreads) {
        const Cd e = a.in[i];
        if (a.mean) s0 += (e.re - m.re) * (e.re - m.re) + (e.im - m.im) * (e.im - m.im);
        else {
            s0 += e.re;
            s1 += e.im;
        }
    }
    red[2 * ctx.tid] = s0;
    red[2 * ctx.tid + 1] = s1;
    ctx.sync();
    if (ctx.tid < a.nclass) {                      // threads of one class are tid, tid + nclass, ...
        double t0 = 0, t1 = 0;
        for (int t = ctx.tid; t < ctx.nthreads; t += a.nclass) {
            t0 += red[2 * t];
            t1 += red[2 * t + 1];
        }
        a.part[((size_t)ctx.bid * a.nclass + ctx.tid) * 2] = t0;
        a.part[((size_t)ctx.bid * a.nclass + ctx.tid) * 2 + 1] = t1;
    }
}
// the partials of one pass added up on the device (one workgroup; fixed order: thread (class c, segment g) adds the partials of
// workgroups g, g + nseg, ... -- consecutive threads read consecutive partials -- then the nseg segment sums of a class are added
// in order): pass 1 leaves the class means, pass 2 the variances and, per column, the first phase of the largest variance
// (core.py:478) -- no host round trip between the passes
struct DecFinishArgs {
    const double *part;  // (nblocks, nclass, 2)
    Cd *mean;            // pass 1: nclass means out; null in pass 2
    int *delay;          // pass 2: ncols sampling phases out
    int nblocks, nclass, ncols, SpS;
    double M;            // samples per class
};
template <class Ctx> SSF_HD void dec_finish_body(Ctx &ctx, const DecFinishArgs &a) {
    double *red = (double *)ctx.lds;               // nthreads x 2 doubles, then nclass variances behind them
    double *var = red + 2 * ctx.nthreads;
    const int nseg = ctx.nthreads / a.nclass, used = nseg * a.nclass;
    double s0 = 0, s1 = 0;
    if (ctx.tid < used)
        for (int w = ctx.tid / a.nclass; w < a.nblocks; w += 8 * nseg) {    // eight loads in flight, added in order
            double v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ww = w + u * nseg;
                const size_t i = ((size_t)(ww < a.nblocks ? ww : w) * a.nclass + ctx.tid % a.nclass) * 2;
                v0[u] = ww < a.nblocks ? a.part[i] : 0.0;
                v1[u] = ww < a.nblocks ? a.part[i + 1] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0 += v0[u];
                s1 += v1[u];
            }
        }
    red[2 * ctx.tid] = s0;
    red[2 * ctx.tid + 1] = s1;
    ctx.sync();
    if (ctx.tid < a.nclass) {
        double t0 = 0, t1 = 0;
        for (int g = 0; g < nseg; ++g) {
            t0 += red[2 * (g * a.nclass + ctx.tid)];
            t1 += red[2 * (g * a.nclass + ctx.tid) + 1];
        }
        if (a.mean) a.mean[ctx.tid] = mk<double>(t0 / a.M, t1 / a.M);
        else var[ctx.tid] = t0 / a.M;
    }
    ctx.sync();
    if (!a.mean && ctx.tid < a.ncols) {            // flat index i = n * ncols + col: class = (n % SpS) * ncols + col
        int best = 0;
        for (int ph = 1; ph < a.SpS; ++ph)
            if (var[ph * a.ncols + ctx.tid] > var[best * a.ncols + ctx.tid]) best = ph;
        a.delay[ctx.tid] = best;
    }
}
// out[j, col] = x[(j * dec + delay[col]) mod N, col]
struct DecGatherArgs {
    const Cd *in;      // (N, ncols)
    Cd *out;           // (Nout, ncols)
    long long N, Nout;
    int ncols, dec;
    const int *delay;  // ncols sampling phases (device memory: dec_finish_body)
};
template <class Ctx> SSF_HD void dec_gather_body(Ctx &ctx, const DecGatherArgs &a) {
    const long long total = a.Nout * a.ncols;
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < total; i += (long long)ctx.nblocks * ctx.nthreads) {
        const long long j = i / a.ncols;
        const int col = (int)(i - j * a.ncols);
        a.out[i] = a.in[((j * a.dec + a.delay[col]) % a.N) * a.ncols + col];
    }
}

// =====================================================================================================
// Receiver chain in one call (RxCore::chain): receiver -> matched filter -> decimate -> edc.  Two fusions around the decimation
// (core.py:435-491) that a sequence of separate calls cannot have:
//   CH_STATS   the matched filter's STORES also accumulate, per (sampling phase, column) class, sum x and sum |x|^2 of what they
//              write -- decimate's variance search without its two passes over the filtered signal.  (Every thread's sixteen
//              outputs are tpf samples apart, so with SpS | tpf they belong to ONE class: three register accumulators per thread.)
//              var = sum |x|^2 / M - |sum x / M|^2: one pass instead of np.var's two; the chosen phase is the largest variance's,
//              and phases differ by far more than the rounding of either form.
//   CH_GATHER  the compensating filter's LOADS pick sample (j dec + delay[col]) mod N of the filtered signal -- the decimated
//              signal is never written or read.
enum { CH_PLAIN = 0, CH_STATS = 1, CH_GATHER = 2 };
struct ChainOlsArgs {
    fused::OlsArgs<double> o;
    double *part;        // CH_STATS: (nblocks, nclass, 3) partial sums: re, im, |x|^2
    int SpS;             // CH_STATS: samples per symbol of the filtered signal (classes = SpS x columns)
    const int *delay;    // CH_GATHER: sampling phase per column (device memory: chain_finish_body)
    int dec;             // CH_GATHER: decimation factor
    long long Nfull;     // CH_GATHER: length of the signal that is sampled
};
template <int LG, int C, int MODE, class Ctx> SSF_HD void chain_ols_body(Ctx &ctx, const ChainOlsArgs &a) {
    double sre = 0, sim = 0, spw = 0;
    fused::ols_body_x<double, LG, C>(
        ctx, a.o,
        [&](long long src, int m) -> Cd {
            if constexpr (MODE == CH_GATHER) return a.o.in[((src * a.dec + a.delay[m]) % a.Nfull) * a.o.in_ld + m];
            else return a.o.in[src * a.o.in_ld + m];
        },
        [&](long long n, int m, Cd v) {
            a.o.out[n * a.o.out_ld + m] = v;
            if constexpr (MODE == CH_STATS) {
                sre += v.re;
                sim += v.im;
                spw += v.re * v.re + v.im * v.im;
            }
        });
    if constexpr (MODE == CH_STATS) {
        // per-class sums of every block group of this workgroup (fixed order: deterministic).  The threads of one group and column
        // whose butterflies are SpS apart share a class; the first SpS butterflies of each add their chain up -- tpf / SpS terms --
        // and leave one partial per (group, class): part[(workgroup * groups + group) * nclass + class]
        constexpr int tpf = 1 << (LG - 4);
        const int nclass = a.SpS * a.o.nrows, tpg = tpf * C, gpw = ctx.nthreads / tpg;
        const int g = ctx.tid / tpg, r = ctx.tid - g * tpg, b = r / C;
        double *red = (double *)ctx.lds;           // nthreads x (re, im, pw)
        ctx.sync();                                // (the transform's LDS traffic is over)
        red[3 * ctx.tid] = sre;
        red[3 * ctx.tid + 1] = sim;
        red[3 * ctx.tid + 2] = spw;
        ctx.sync();
        if (b < a.SpS) {
            double t0 = 0, t1 = 0, t2 = 0;
            for (int t = ctx.tid; t < (g + 1) * tpg; t += a.SpS * C) {
                t0 += red[3 * t];
                t1 += red[3 * t + 1];
                t2 += red[3 * t + 2];
            }
            // the class of this chain: the one its threads stored under -- or, for a chain that stored nothing (beyond the signal's
            // end, an idle group), any class with zeros; every (group, class) slot is written exactly once because the SpS chains of a
            // column cover the SpS phases
            const long long job = (long long)ctx.bid * gpw + g;
            const int ncg = a.o.nrows / C;
            const long long jb = ncg == 1 ? job : job / ncg;
            const int m = (int)(job - jb * ncg) * C + (r - b * C);
            const long long n0 = (jb + a.o.blk0) * a.o.d - a.o.discard - a.o.D - a.o.Dx;
            const int ph = (int)(((n0 + b) % a.SpS + a.SpS) % a.SpS);
            double *o = a.part + (((size_t)ctx.bid * gpw + g) * nclass + (size_t)ph * a.o.nrows + m) * 3;
            o[0] = t0;
            o[1] = t1;
            o[2] = t2;
        }
    }
}
// decimate by itself, ONE pass over the signal (round 6; two passes + two reductions before): per (phase, column) class sum x and
// sum |x|^2 together, left as (nblocks, nclass, 3) partials for chain_finish_body, which forms the variances sum |x|^2 / M -
// |sum x / M|^2 and picks the phases -- the same search as the receiver chain's (chain_ols_body, CH_STATS)
struct DecStatsArgs {
    const Cd *in;        // (N, ncols)
    double *part;        // (nblocks, nclass, 3): sum re, sum im, sum |x|^2
    long long total;     // N * ncols
    int nclass, ncols;
};
template <class Ctx> SSF_HD void dec_stats_body(Ctx &ctx, const DecStatsArgs &a) {
    double *red = (double *)ctx.lds;               // nthreads x 3 doubles
    // sums of x - k with k = the column's first sample: the variance does not change, and a signal that is mostly offset (a
    // photocurrent) does not lose its variance in the difference of two large numbers
    const Cd k = a.in[(ctx.tid % a.nclass) % a.ncols];
    double s0 = 0, s1 = 0, s2 = 0;
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < a.total; i += (long long)ctx.nblocks * ctx.nthreads) {
        const Cd e = mk<double>(a.in[i].re - k.re, a.in[i].im - k.im);
        s0 += e.re;
        s1 += e.im;
        s2 += e.re * e.re + e.im * e.im;
    }
    red[3 * ctx.tid] = s0;
    red[3 * ctx.tid + 1] = s1;
    red[3 * ctx.tid + 2] = s2;
    ctx.sync();
    if (ctx.tid < a.nclass) {                      // threads of one class are tid, tid + nclass, ... (the grid stride keeps the class)
        double t0 = 0, t1 = 0, t2 = 0;
        for (int t = ctx.tid; t < ctx.nthreads; t += a.nclass) {
            t0 += red[3 * t];
            t1 += red[3 * t + 1];
            t2 += red[3 * t + 2];
        }
        double *o = a.part + ((size_t)ctx.bid * a.nclass + ctx.tid) * 3;
        o[0] = t0;
        o[1] = t1;
        o[2] = t2;
    }
}
// the partials added up (one workgroup, fixed order), the variances, and per column the first phase of the largest one (core.py:478)
struct ChainFinishArgs {
    const double *part;  // (nblocks, nclass, 3)
    int *delay;          // ncols sampling phases out
    int nblocks, nclass, ncols, SpS;
    double M;            // samples per class
};
template <class Ctx> SSF_HD void chain_finish_body(Ctx &ctx, const ChainFinishArgs &a) {
    double *red = (double *)ctx.lds;               // nthreads x 3 doubles, then nclass variances
    double *var = red + 3 * (size_t)ctx.nthreads;
    const int nseg = ctx.nthreads / a.nclass, used = nseg * a.nclass;
    double s0 = 0, s1 = 0, s2 = 0;
    if (ctx.tid < used)
        for (int w = ctx.tid / a.nclass; w < a.nblocks; w += 4 * nseg) {      // four partials in flight, added in order
            double v[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ww = w + u * nseg;
                const double *q = a.part + ((size_t)(ww < a.nblocks ? ww : w) * a.nclass + ctx.tid % a.nclass) * 3;
                for (int k = 0; k < 3; ++k) v[u][k] = ww < a.nblocks ? q[k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0 += v[u][0];
                s1 += v[u][1];
                s2 += v[u][2];
            }
        }
    red[3 * ctx.tid] = s0;
    red[3 * ctx.tid + 1] = s1;
    red[3 * ctx.tid + 2] = s2;
    ctx.sync();
    if (ctx.tid < a.nclass) {
        double t0 = 0, t1 = 0, t2 = 0;
        for (int g = 0; g < nseg; ++g) {
            t0 += red[3 * (g * a.nclass + ctx.tid)];
            t1 += red[3 * (g * a.nclass + ctx.tid) + 1];
            t2 += red[3 * (g * a.nclass + ctx.tid) + 2];
        }
        const double mr = t0 / a.M, mi = t1 / a.M;
        var[ctx.tid] = t2 / a.M - (mr * mr + mi * mi);
    }
    ctx.sync();
    if (ctx.tid < a.ncols) {
        int best = 0;
        for (int ph = 1; ph < a.SpS; ++ph)
            if (var[ph * a.ncols + ctx.tid] > var[best * a.ncols + ctx.tid]) best = ph;
        a.delay[ctx.tid] = best;
    }
}
// f(LG, C) for the instantiation a chain launch needs (the matched filter and the compensating filter of a 2-polarisation chain:
// transforms of 2048 ... 8192 points); false: no such kernel -- the chain then runs its stages one by one
template <class F> inline bool chain_ols_dispatch(const fused::OlsLaunch &o, F &&f) {
    using std::integral_constant;
    if (o.lg == 11 && o.C == 2) f(integral_constant<int, 11>{}, integral_constant<int, 2>{});
    else if (o.lg == 12 && o.C == 2) f(integral_constant<int, 12>{}, integral_constant<int, 2>{});
    else if (o.lg == 13 && o.C == 1) f(integral_constant<int, 13>{}, integral_constant<int, 1>{});
    else if (o.lg == 10 && o.C == 2) f(integral_constant<int, 10>{}, integral_constant<int, 2>{});
    else return false;
    return true;
}
inline bool chain_ols_supported(const fused::OlsLaunch &o) { return ((o.lg >= 10 && o.lg <= 12) && o.C == 2) || (o.lg == 13 && o.C == 1); }

}  // namespace rx
}  // namespace ssf
