// fused_kernels.h -- kernel bodies of the fused pipeline, written against an execution
// context `Ctx` (tid / bid / nthreads / nblocks / lds / sync()) so that the same source is
// launched as HIP kernels on gfx950 (engine_fused_impl.h) and stepped by the CPU emulator in
// tests/emu (fibers; sync() yields).  No wave intrinsics: block reductions go through LDS.
//
// Decomposition of one length-N transform of a row (N = N1*N2, n = n1*N2 + n2,
// k = k1 + N1*k2), all buffers in the natural [row][n1][n2] matrix layout:
//
//   column kernel (C columns x N1 per workgroup, both polarisations of a pair):
//        [G -> inverse column FFT (DIF, natural k1 in, digit-reversed n1 in registers)]
//        -> time-domain work on registers (power, Kerr rotation, convergence sums, E_hd / field I/O)
//        -> [forward column FFT (DIT, digit-reversed in, natural k1 out) -> G]
//   row kernel (one contiguous row of N2 per transform):
//        G -> forward row FFT (DIF) -> * linear operator / N (from the bin index)
//        -> inverse row FFT (DIT) -> G
//   (the inter-pass twiddle W_N^{n2 k1} is applied by the column kernel on its G loads/stores)
//
// so FFT . H . IFFT costs two HBM round trips and no transposes.  Reference semantics:
// optic/models/channels.py:387-441 (Manakov step), :219-229 (NLSE step).
#pragma once
#include "fused_core.h"
#include "mixed_fft.h"
#include "ssf_rng.h"

namespace ssf {
namespace fused {

// ------------------------------------------------------------------------------- control
// Launch sequence per span: Col, then [Row, Col] pairs.  Every launch reads the state, does the
// matching stage (or nothing) and forwards the state; the host never looks inside a chunk.
//
// Convergence test without the previous iterate.  The reference compares consecutive iterates
// after the second linear step, lim = |E_fd(i) - E_fd(i-1)| / |E_fd(i-1)| (channels.py:424,
// 517-519).  Both iterates are Lin(E_hd * rot), Lin = ifft(fft(.) * linOperator) with
// |linOperator| = exp(-alpha hz/4) on every bin, so by Parseval
//     lim_i = sqrt(sum |E_hd|^2 |rot_i - rot_{i-1}|^2) / sqrt(sum |E_hd|^2),   i >= 1,
// which the column stage that BUILDS iterate i can evaluate from E_hd and the two phase arrays
// (|rot_i - rot_{i-1}|^2 = 4 sin^2((theta_i - theta_{i-1})/2)) -- one whole iteration before
// E_fd(i) exists.  The pipeline therefore knows in advance which iterate is the last one: no
// iterate is ever written to or re-read from HBM just for the test, only the final one is stored.
// lim_0 (against the field at the step start, channels.py:381-382) is evaluated directly; if it
// ever signals convergence the iterate is rebuilt as final (ST_REDO0).  Only that decision needs lim_0,
// and lim_0 is normally orders of magnitude above tol: unless a trace is recorded the numerator is taken
// over one sample in sixteen (a rigorous lower bound with the exact denominator sum Pch), which spares
// re-reading the step-start field (32 MiB per step at N = 2^20).  A bound below tol proves nothing: iterate 0
// is then rebuilt and lim_0 evaluated on all samples (exact0) before anything is decided.
enum {
    ST_NEED_S = 0,     // Col: span start: Pch of T[cur], forward column FFT                      -> AFTER_S
    ST_AFTER_S = 1,    // Row: new step: step size / operator, first half linear step             -> NEED_H
    ST_NEED_H = 2,     // Col: E_hd out, first rotation, forward column FFT                       -> ROW_ITER
    ST_ROW_ITER = 3,   // Row: evaluate pending convergence sums, second linear step of iterate it -> NEED_I
    ST_NEED_I = 4,     // Col: E_fd(it).  Not final: next iterate + sums of lim_{it+1}            -> ROW_ITER
                       //      final: field out, next step's Pch + forward FFT         -> AFTER_S | SPAN_DONE
    ST_SPAN_DONE = 6,
    ST_REDO0 = 7,      // Col: rebuild iterate 0 as the final one (lim_0 < tol)                   -> ROW_ITER
    // The final stage of a step that another step follows stores the field at ONE sample in sixteen -- all that the next step's
    // bound of lim_0 reads (Ctrl::t_sparse).  If that bound cannot exclude convergence at iterate 0 (the rounding-sized last step
    // of a span: lim_0 ~ 1e-12), lim_0 has to be measured on every sample of the field at the step start, which is then
    // recovered first: E(z) = Lin^-1(E_hd), one more column / row / column round, the last one also rebuilding iterate 0.
    ST_RECOVER_A = 8,  // Col: E_hd -> forward column FFT                                         -> RECOVER_ROW
    ST_RECOVER_ROW = 9,// Row: inverse half linear step                                           -> RECOVER_B
    ST_RECOVER_B = 10  // Col: field at the step start out (all samples); then as REDO0           -> ROW_ITER
};

struct LinOp {          // exp(argLimOp * hz/2) / N evaluated from the bin index (row kernel)
    double cth;         // phase = cth * kk^2, kk = signed bin index
    double mag;         // exp(lin_a * hzh) / N
    double Cre[9], Cim[9];   // cis(cth * (N/16)^2 * m^2), m = 0..8
};

struct Ctrl {           // device-resident step state, double-buffered by launch parity
    int state, it, cur, hz_valid;
    int final_;         // the I stage of iterate `it` is the last one of this step
    int pend0, pendn;   // partial sums of lim_0 / lim_it are waiting for the next Row launch
    int cap0;           // iterate 0 ended the step only because maxIter == 1 (lim_0 decides non-convergence)
    int pcur, redo_;    // current Pch buffer; iterate 0 is being rebuilt as final
    int bound0;         // the pending lim_0 sums cover a sixteenth of the samples: a lower bound of lim_0
    int exact0;         // the next I stage of iterate 0 evaluates lim_0 on all samples (bound was inconclusive)
    int t_sparse;       // the field at the step start (T[cur]) holds one sample in sixteen only (see ST_RECOVER_A)
    int dense;          // this step needed the exact lim_0: its final stage stores every sample
    int gscale;         // G holds the column spectrum of the field DIVIDED by N1: the final I stage of the last step did not
                        // rewrite it (see mk_col_stage); the next row stage multiplies its operator by N1
    int last_nit;       // iterations of the latest finished step (the host predicts the stage sequence from it: FusedCore::run_span)
    long long pend0_idx;// trace row lim_0 belongs to
    double z, hz;
    long long steps, iterations, nonconv, trace_n;
    long long n_ahead, n_rebuilt;   // iterations decided in advance / iterates rebuilt
    long long n_recovered;          // step-start fields recovered (ST_RECOVER_A)
    LinOp lin;
};

struct MkConst {        // per-execute constants (by value)
    double Lspan, hz_fixed, tol, maxRot, c8g, sgn, lin_a, lin_b, w2, invN;
    int maxIter, adaptive, log2N;
    int exact_lim0;     // lim_0 always on all samples (a trace is recorded, or maxIter == 1: lim_0 decides convergence)
    long long trace_cap;
    double *tr_hz;
    int *tr_it;
    double *tr_lim;
};

SSF_HD LinOp make_linop(double hzh, double lin_a, double lin_b, double w2, double invN, int log2N) {
    LinOp l;
    l.cth = lin_b * w2 * hzh;
    l.mag = exp(lin_a * hzh) * invN;
    const double d = (double)(1ll << (log2N - 4));
    for (int m = 0; m < 9; ++m) {
        double s, c;
        cis_rad_d(l.cth * d * d * (double)(m * m), c, s);     // (not the full-range sincos: nine inlined copies of
                                                              //  its reduction are a fifth of the row kernel's code)
        l.Cre[m] = c;
        l.Cim[m] = s;
    }
    return l;
}

// channels.py:392-403
SSF_HD double pick_hz(const MkConst &k, double z, double maxphi) {
    if (k.adaptive) {
        const double cand = k.maxRot / maxphi;
        return (k.Lspan - z >= cand) ? cand : k.Lspan - z;
    }
    return (k.Lspan - z < k.hz_fixed) ? k.Lspan - z : k.hz_fixed;
}

// ------------------------------------------------------------------------ thread-level FFT
// (mul_by_d -- factors applied as hi + lo pairs in single precision -- lives in fused_core.h: the mixed-radix rows use it too)
// w[s] = cis(sign * 2 pi j s / 2^lgL), s = 0..R-1.  The power tree always runs in double and is
// rounded once, where it is applied: in single precision a float tree gives every twiddle a magnitude error
// that is the same at every step (w^s inherits s times the rounding of w), and those errors add up
// coherently over thousands of steps (measured -0.26 % power after 2000 steps with a float tree).
SSF_HD cx<double> tw_base(int sign, int j, int lgL) {
    double c, s;
    cis2pi_d(scale_pow2((double)(sign * j), lgL), c, s);
    return mk<double>(c, s);
}
template <int R> SSF_HD void tw_powers(cx<double> w1, cx<double> *p) {
    p[0] = mk<double>(1.0, 0.0);
    p[1] = w1;
    if (R > 2) {
        p[2] = w1 * w1;
        p[3] = p[2] * w1;
    }
    if (R > 4) {
        p[4] = p[2] * p[2];
        p[5] = p[4] * w1;
        p[6] = p[3] * p[3];
        p[7] = p[4] * p[3];
    }
    if (R > 8) {
        p[8] = p[4] * p[4];
#pragma unroll
        for (int s2 = 9; s2 < 16; ++s2) p[s2 < R ? s2 : 0] = p[8] * p[s2 - 8];
    }
}

// Where the twiddles of a transform come from.
//  * Generated in registers (the default): one sincospi per pass for the base w = cis(2 pi j / L_i) + a double-precision
//    power tree of depth <= 4, rounded once.  The library sincospi is ~100 issue slots (a Horner scheme whose 24
//    coefficients are moved into registers one by one), as much as the tree and the fifteen products it feeds, and a
//    kernel's forward and inverse transforms use conjugate bases: SSF_TW_REUSE keeps the bases of the first transform (4
//    registers per pass; j is the same for every butterfly a thread carries in a pass >= 1, and in pass 0 there is one)
//    and the second one conjugates them -- half the sincospi calls of a launch.
//  * SSF_TW_TAB bit 0: the next-to-last pass (pass 0 of a two-pass transform, pass 1 of a three-pass one) reads its factors
//    from a table in LDS that the workgroup builds when it starts (fused_core.h: tw_entry_t; j < r_last, s < r: at most 256
//    entries = 4 KiB).  Measured (profiles/r4_ab_twiddle_tables.txt, same box): it pays only where a factor is expensive to
//    make -- the hi + lo float quadruples of the single-precision rows (row launch 49.2 -> 47.2 us at config 3) -- and costs
//    where the LDS is the busy unit already: double precision rows 20.8 -> 21.4 us, columns 25.5 -> 26.3 us, packed columns
//    60.8 -> 62.1 us.  The host therefore offers the table (tw_off > 0) to the single-precision row stage only.  A second
//    table for pass 0 of three-pass transforms in global memory (L2-resident, 64 KiB) gained nothing on top (rows 47.6 us)
//    and cost double precision another 1.3 us per row launch: removed again.
#ifndef SSF_TW_TAB
#define SSF_TW_TAB 1
#endif
#ifndef SSF_TW_REUSE
#define SSF_TW_REUSE 1
#endif
constexpr int kTwMaxPass = 5;
template <typename T> struct TwSrc {
    const tw_entry_t<T> *lds = nullptr;      // table of pass npass - 2: entry [s * r_last + j]
    cx<double> base[kTwMaxPass];             // SSF_TW_REUSE: cis(+2 pi j / L_i) of the passes generated so far
    bool have[kTwMaxPass] = {false, false, false, false, false};
};
constexpr int kTwLdsBytes = 4096;

// the workgroup's LDS table (no barrier inside: the caller's next barrier -- the first exchange of a three-pass transform, an
// explicit one for two passes -- publishes it)
template <typename T, class Ctx> SSF_HD void tw_lds_build(Ctx &ctx, const PassPlan &p, tw_entry_t<T> *tab) {
    if (!(SSF_TW_TAB & 1) || p.npass < 2) return;
    const int it = p.npass - 2, lgr = p.lg(it), lgl = p.lg(p.npass - 1);
    for (int t = ctx.tid; t < (1 << (lgr + lgl)); t += ctx.nthreads) {
        const int s = t >> lgl, j = t & ((1 << lgl) - 1);
        double c, sn;
        cis2pi_d(scale_pow2((double)(-(j * s)), lgr + lgl), c, sn);
        tab[t] = tw_make<T>(mk<double>(c, sn));
    }
}

// twiddles of the R values of one butterfly (bb) of pass i: v[s] *= cis(SIGN 2 pi j s / L_i), s = 1 .. R-1
// (first: this is the first butterfly of the pass this thread carries -- the only one that may set / use the kept base)
template <int SIGN, int R, bool TAB, typename T>
SSF_HD void apply_tw(const PassPlan &p, int i, int bb, cx<T> *v, TwSrc<T> &src, bool first) {
    if (TAB && i == p.npass - 2) {
        const int lgl = p.lg(p.npass - 1);
        const tw_entry_t<T> *e = src.lds + (bb & ((1 << lgl) - 1));
#pragma unroll
        for (int s = 1; s < R; ++s) v[s] = tw_mul<(SIGN > 0)>(v[s], e[s << lgl]);
        return;
    }
    cx<double> w1;
    const bool keep = SSF_TW_REUSE && i < kTwMaxPass && (first || i > 0);      // (pass >= 1: every butterfly of the thread has this j)
    if (keep && src.have[i]) {
        w1 = SIGN > 0 ? src.base[i] : conj(src.base[i]);
    } else {
        w1 = tw_base(SIGN, pass_j(p, i, bb), pass_lgLi(p, i));
        if (keep) {
            src.base[i] = SIGN > 0 ? w1 : conj(w1);
            src.have[i] = true;
        }
    }
    cx<double> w[R];
    tw_powers<R>(w1, w);
#pragma unroll
    for (int s = 1; s < R; ++s) v[s] = mul_by_d(v[s], w[s]);
}

// butterflies of pass i for thread b (values v[u*r + q]); DIF: DFT then twiddle w^s.  V = values per thread (16, or 8 for
// the 128-register kernels): a thread carries V / r butterflies of a radix-r pass.
template <int SIGN, int V = 16, bool TAB = false, typename T>
SSF_HD void dif_pass(const PassPlan &p, int i, int b, cx<T> *v, TwSrc<T> &src) {
    const bool tw = p.lgLn(i) > 0;
    switch (p.lg(i)) {
    case 4:
        if constexpr (V >= 16) {
            dft16<SIGN>(v);
            if (tw) apply_tw<SIGN, 16, TAB>(p, i, b, v, src, true);
        }
        break;
    case 3:
#pragma unroll
        for (int u = 0; u < V / 8; ++u) {
            dft8<SIGN>(v + 8 * u);
            if (tw) apply_tw<SIGN, 8, TAB>(p, i, b + p.tpf * u, v + 8 * u, src, u == 0);
        }
        break;
    case 2:
#pragma unroll
        for (int u = 0; u < V / 4; ++u) {
            dft4<SIGN>(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
            if (tw) apply_tw<SIGN, 4, TAB>(p, i, b + p.tpf * u, v + 4 * u, src, u == 0);
        }
        break;
    default:
#pragma unroll
        for (int u = 0; u < V / 2; ++u) {
            dft2<SIGN>(v[2 * u], v[2 * u + 1]);
            if (tw) apply_tw<SIGN, 2, TAB>(p, i, b + p.tpf * u, v + 2 * u, src, u == 0);
        }
        break;
    }
}

// DIT: twiddle w^s then DFT (exact mirror of dif_pass with the opposite SIGN)
template <int SIGN, int V = 16, bool TAB = false, typename T>
SSF_HD void dit_pass(const PassPlan &p, int i, int b, cx<T> *v, TwSrc<T> &src) {
    const bool tw = p.lgLn(i) > 0;
    switch (p.lg(i)) {
    case 4:
        if constexpr (V >= 16) {
            if (tw) apply_tw<SIGN, 16, TAB>(p, i, b, v, src, true);
            dft16<SIGN>(v);
        }
        break;
    case 3:
#pragma unroll
        for (int u = 0; u < V / 8; ++u) {
            if (tw) apply_tw<SIGN, 8, TAB>(p, i, b + p.tpf * u, v + 8 * u, src, u == 0);
            dft8<SIGN>(v + 8 * u);
        }
        break;
    case 2:
#pragma unroll
        for (int u = 0; u < V / 4; ++u) {
            if (tw) apply_tw<SIGN, 4, TAB>(p, i, b + p.tpf * u, v + 4 * u, src, u == 0);
            dft4<SIGN>(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
        }
        break;
    default:
#pragma unroll
        for (int u = 0; u < V / 2; ++u) {
            if (tw) apply_tw<SIGN, 2, TAB>(p, i, b + p.tpf * u, v + 2 * u, src, u == 0);
            dft2<SIGN>(v[2 * u], v[2 * u + 1]);
        }
        break;
    }
}

// transform-local position of register idx (0..V-1) of thread b in pass i
SSF_HD int reg_pos(const PassPlan &p, int i, int b, int idx) {
    const int lg = p.lg(i);
    return pass_pos(p, i, b + p.tpf * (idx >> lg), idx & ((1 << lg) - 1));
}

// CI = slot multiplier: 1 = the transform's slots are contiguous (rows; columns side by side, one after the other); CI = C > 1:
// the C columns of a workgroup are INTERLEAVED, slot lds_slot(pos) * C + c (the caller's `lds` points at column c's slot 0).
// Lane l of a wave works on column l % C and butterfly l / C, so the lanes of one LDS instruction then cover C consecutive
// 16-byte slots per butterfly, and lds_slot() makes consecutive butterflies follow each other in every pass: the 8-lane
// groups of ds_write_b128 (bank = slot mod 8) and the four 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...; bank =
// slot mod 16) are conflict-free for C = 4, 8, 16 in every pass (tools/exp/lds_conflicts.py, MI355X_MICROARCH.md LDS table).
// Side by side (slot = c * stride + lds_slot(pos)) no stride frees both: C = 8 reads were three-way conflicted (SQ_LDS_BANK_CONFLICT
// 0.72 x SQ_ACTIVE_INST_LDS in k_col<double,8,3>), C = 4 writes two-way (1.18 x in k_col_pk<10>).
template <int V = 16, int CI = 1, typename T> SSF_HD void lds_put(const PassPlan &p, int i, int b, const cx<T> *v, cx<T> *lds) {
#pragma unroll
    for (int idx = 0; idx < V; ++idx) lds[lds_slot(reg_pos(p, i, b, idx)) * CI] = v[idx];
}
template <int V = 16, int CI = 1, typename T> SSF_HD void lds_get(const PassPlan &p, int i, int b, cx<T> *v, const cx<T> *lds) {
#pragma unroll
    for (int idx = 0; idx < V; ++idx) v[idx] = lds[lds_slot(reg_pos(p, i, b, idx)) * CI];
}

// DIF transform: v holds pass-0 positions on entry, pass-(p-1) positions (digit-reversed) on exit
template <int SIGN, int V = 16, bool TAB = false, int CI = 1, typename T, class Ctx>
SSF_HD void fft_dif(Ctx &ctx, const PassPlan &p, int b, cx<T> *v, cx<T> *lds, TwSrc<T> &src) {
    dif_pass<SIGN, V, TAB>(p, 0, b, v, src);
#pragma unroll
    for (int i = 1; i < p.npass; ++i) {
        lds_put<V, CI>(p, i - 1, b, v, lds);
        ctx.sync();
        lds_get<V, CI>(p, i, b, v, lds);
        dif_pass<SIGN, V, TAB>(p, i, b, v, src);
    }
}
// DIT transform: v holds pass-(p-1) positions on entry, pass-0 positions (natural) on exit
template <int SIGN, int V = 16, bool TAB = false, int CI = 1, typename T, class Ctx>
SSF_HD void fft_dit(Ctx &ctx, const PassPlan &p, int b, cx<T> *v, cx<T> *lds, TwSrc<T> &src) {
#pragma unroll
    for (int i = p.npass - 1; i >= 1; --i) {
        dit_pass<SIGN, V, TAB>(p, i, b, v, src);
        lds_put<V, CI>(p, i, b, v, lds);
        ctx.sync();
        lds_get<V, CI>(p, i - 1, b, v, lds);
    }
    dit_pass<SIGN, V, TAB>(p, 0, b, v, src);
}

template <int SIGN, int V = 16, typename T, class Ctx> SSF_HD void fft_dif(Ctx &ctx, const PassPlan &p, int b, cx<T> *v, cx<T> *lds) {
    TwSrc<T> none;
    fft_dif<SIGN, V, false>(ctx, p, b, v, lds, none);
}
template <int SIGN, int V = 16, typename T, class Ctx> SSF_HD void fft_dit(Ctx &ctx, const PassPlan &p, int b, cx<T> *v, cx<T> *lds) {
    TwSrc<T> none;
    fft_dit<SIGN, V, false>(ctx, p, b, v, lds, none);
}

// ------------------------------------------------------------------------ block reductions
// Every thread gets the result; `red` is LDS scratch of nthreads doubles.  Two stages (<= 64
// partial sums), fixed order: deterministic and identical in every workgroup.
template <bool MAX, class Ctx> SSF_HD double block_reduce(Ctx &ctx, double v, double *red) {
    if constexpr (Ctx::kWaveOps) {
        if (ctx.nthreads >= 64) {             // wave butterfly + one LDS hop (device only; fixed order)
            v = MAX ? ctx.wave_max(v) : ctx.wave_sum(v);
            const int w = ctx.tid >> 6, nw = ctx.nthreads >> 6;
            ctx.sync();
            if ((ctx.tid & 63) == 0) red[w] = v;
            ctx.sync();
            double s = red[0];
            for (int i = 1; i < nw; ++i) s = MAX ? (red[i] > s ? red[i] : s) : s + red[i];
            return s;
        }
    }
    ctx.sync();
    red[ctx.tid] = v;
    ctx.sync();
    const int nt = ctx.nthreads, ng = nt < 64 ? nt : 64, per = nt / ng;     // nt is a power of two
    if (ctx.tid < ng) {
        double s = red[ctx.tid * per];
        for (int i = 1; i < per; ++i) {
            const double x = red[ctx.tid * per + i];
            s = MAX ? (x > s ? x : s) : s + x;
        }
        v = s;
    }
    ctx.sync();
    if (ctx.tid < ng) red[ctx.tid] = v;
    ctx.sync();
    double s = red[0];
    for (int i = 1; i < ng; ++i) s = MAX ? (red[i] > s ? red[i] : s) : s + red[i];
    return s;
}
// sum of two values at once (one barrier sequence): results in s0, s1
template <class Ctx> SSF_HD void block_sum2(Ctx &ctx, double &s0, double &s1, double *red) {
    if constexpr (Ctx::kWaveOps) {
        if (ctx.nthreads >= 64) {
            s0 = ctx.wave_sum(s0);
            s1 = ctx.wave_sum(s1);
            const int w = ctx.tid >> 6, nw = ctx.nthreads >> 6;
            ctx.sync();
            if ((ctx.tid & 63) == 0) {
                red[2 * w] = s0;
                red[2 * w + 1] = s1;
            }
            ctx.sync();
            double a = 0, b = 0;
            for (int i = 0; i < nw; ++i) {
                a += red[2 * i];
                b += red[2 * i + 1];
            }
            s0 = a;
            s1 = b;
            return;
        }
    }
    ctx.sync();
    red[2 * ctx.tid] = s0;
    red[2 * ctx.tid + 1] = s1;
    ctx.sync();
    const int nt = ctx.nthreads, ng = nt < 64 ? nt : 64, per = nt / ng;
    if (ctx.tid < ng) {
        double a = 0, b = 0;
        for (int i = 0; i < per; ++i) {
            a += red[2 * (ctx.tid * per + i)];
            b += red[2 * (ctx.tid * per + i) + 1];
        }
        s0 = a;
        s1 = b;
    }
    ctx.sync();
    if (ctx.tid < ng) {
        red[2 * ctx.tid] = s0;
        red[2 * ctx.tid + 1] = s1;
    }
    ctx.sync();
    double a = 0, b = 0;
    for (int i = 0; i < ng; ++i) {
        a += red[2 * i];
        b += red[2 * i + 1];
    }
    s0 = a;
    s1 = b;
}
template <class Ctx> SSF_HD double block_sum(Ctx &ctx, double v, double *red) { return block_reduce<false>(ctx, v, red); }
template <class Ctx> SSF_HD double block_max(Ctx &ctx, double v, double *red) { return block_reduce<true>(ctx, v, red); }
// deterministic reduction of a global partial array by the whole block (same order in every block)
template <class Ctx> SSF_HD double global_sum(Ctx &ctx, const double *a, int n, double *red) {
    double s = 0;
    for (int i = ctx.tid; i < n; i += ctx.nthreads) s += a[i];
    return block_sum(ctx, s, red);
}
template <class Ctx> SSF_HD double global_max(Ctx &ctx, const double *a, int n, double *red) {
    double s = -INFINITY;
    for (int i = ctx.tid; i < n; i += ctx.nthreads) s = a[i] > s ? a[i] : s;
    return block_max(ctx, s, red);
}

// ------------------------------------------------------------------------------ row kernel
template <typename T> struct RowArgs {
    cx<T> *G;                 // (nrows, N1, N2)
    int log2N1, log2N2, nfft; // nfft = nrows * N1 row transforms
    int use_ctrl;             // 1: Manakov state machine (ctrl), 0: explicit LinOp (NLSE / linear channel)
    const LinOp *lin;         // use_ctrl == 0
    const Ctrl *cin;          // ctrl[seq & 1]
    Ctrl *cout;               // ctrl[(seq + 1) & 1]
    MkConst k;
    const double *pmax;       // adaptive: block maxima of phi written by the step-start stage
    const double *pnum, *pden;   // partial sums of lim_it written by the I stage
    const double *pnum0, *pden0; // partial sums of lim_0
    int npart;
    int N2;                   // row length (mixed-radix rows: any 2^a 3^b 5^c; radix-2^n rows: 1 << log2N2)
    long long N;              // N1 * N2
    int mixed;                // 1: row_mixed_body (N2 has factors 3 / 5)
    MixPlan plan;             // its pass plan
    const cx<double> *wtab;   // cis(-2 pi k / N2), k < N2
    int rows_per_wg;          // rows a workgroup handles side by side (nthreads / rows_per_wg threads each)
    // convolution with a fixed kernel (Bluestein transforms of the general-length path, conv_engine): the multiplier is
    // an array in the row kernel's own spectrum order, [k1][position after the forward passes], the same for every field row
    const cx<T> *harr;        // use_ctrl == 0 and lin == nullptr: spectrum *= harr[(rr mod N1) * N2 + position]
    int fwd_only;             // 1: stop after the forward row transform and store the spectrum in that order (makes harr)
    const cx<T> *src;         // mixed-radix rows only: read the rows from here instead of G (out-of-place; nullptr = in place)
    int vpt;                  // values per thread of the radix-2^n row kernel: 16 (0 = 16) or 8
    // independent units (grid.y = number of units): unit u works on G + u * u_elems with control blocks cin[u] / cout[u]
    // and partial sums at + u * u_part of every array (unit_view below); 0 / unused for a single unit
    long long u_elems;
    int u_part;
    int prio;                 // 1: issue priority by phase (s_setprio); 0 when several plans share the GPU (lanes)
    int tw_off;               // single precision: byte offset of the workgroup's twiddle table in LDS (kTwLdsBytes behind the transform area)
    int N1mix;                // > 0: the column length when it is not 1 << log2N1 (col_mixed_body: N = N1mix x N2, both 2^a 3^b 5^c)
};
// number of rows (column length) of the N1 x N2 matrix a row stage works on
template <typename T> SSF_HD long long row_n1(const RowArgs<T> &a) { return a.N1mix ? (long long)a.N1mix : 1ll << a.log2N1; }

// linear operator for the V registers of a last-radix-V butterfly (V = 16 | 8): bins k0 + (N/V) q, signed q' = q or q - V;
// phase cth (k0 + dk q')^2 = A * B^q' * C_|q'| with C_m = cis(cth dk^2 m^2) = the control block's table at 16/V * m
// (the two sines / cosines of it by themselves: they depend on the thread's bins only, not on the data, so the row stage
//  evaluates them while its row is still in flight -- SSF_EARLY_BASES)
struct Lin16Bases {
    cx<double> A, B1;       // A: the unit phasor cis(cth k0^2) (the magnitude is applied with the operator), B1 = cis(2 cth k0 dk)
    double cth;             // the operator constant they were made for
};
template <int V = 16> SSF_HD Lin16Bases lin16_bases(double cth, long long k0, int log2N) {
    constexpr int lgV = V == 16 ? 4 : 3;
    const double dk = (double)(1ll << (log2N - lgV));
    const double k0d = (double)k0;
    double s, c;
    Lin16Bases r;
    cis_rad_d(cth * k0d * k0d, c, s);
    r.A = mk<double>(c, s);
    cis_rad_d(2.0 * cth * k0d * dk, c, s);
    r.B1 = mk<double>(c, s);
    r.cth = cth;
    return r;
}
template <int V = 16, typename T> SSF_HD void apply_lin16(const LinOp &lo, const Lin16Bases &lb, cx<T> *v);
template <int V = 16, typename T>
SSF_HD void apply_lin16(const LinOp &lo, long long k0, int log2N, cx<T> *v) {
    apply_lin16<V>(lo, lin16_bases<V>(lo.cth, k0, log2N), v);
}
template <int V, typename T>
SSF_HD void apply_lin16(const LinOp &lo, const Lin16Bases &lb, cx<T> *v) {
    constexpr int H = V / 2, CS = 16 / V;
    const cx<double> A = mk<double>(lo.mag * lb.A.re, lo.mag * lb.A.im);
    cx<double> Bp[H + 1];
    Bp[0] = mk<double>(1.0, 0.0);
    Bp[1] = lb.B1;
    Bp[2] = Bp[1] * Bp[1];
    Bp[3] = Bp[2] * Bp[1];
    Bp[4] = Bp[2] * Bp[2];
    if constexpr (V == 16) {
        Bp[5] = Bp[4] * Bp[1];
        Bp[6] = Bp[3] * Bp[3];
        Bp[7] = Bp[4] * Bp[3];
        Bp[8] = Bp[4] * Bp[4];
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
        const int m = q < H ? q : V - q;                        // |q'|, q' = q (q < V/2) or q - V
        const cx<double> bq = q < H ? Bp[m] : conj(Bp[m]);
        const cx<double> h = A * bq * mk<double>(lo.Cre[CS * m], lo.Cim[CS * m]);
        v[q] = mul_by_d(v[q], h);
    }
}
// general (slow) form for short rows whose last radix is not 16
SSF_HD cx<double> lin_at(const LinOp &lo, long long k, int log2N) {
    const long long N = 1ll << log2N;
    const double kk = (double)(k < N / 2 ? k : k - N);
    double s, c;
    cis_rad_d(lo.cth * kk * kk, c, s);
    return mk<double>(lo.mag * c, lo.mag * s);
}

// Forward nwords 8-byte words of the control block (lead thread).  ALL words are fetched before any is stored: copied one by
// one, every load waits for the store before it (the compiler must assume the blocks alias), i.e. 35 memory round trips in a row
// -- 3.5 us that the lead workgroup finishes late, a quarter of a launch at small N (round 2: groups of eight).  Round 5: the 38
// words of the block were still four groups and six single words = ten dependent round trips in front of the lead workgroup's own
// work, in EVERY launch (the generated code shows a load + s_waitcnt vmcnt(0) + store per word of the tail); now one round trip.
#ifndef SSF_CTRL_WIDE
#define SSF_CTRL_WIDE 1
#endif
SSF_HD void ctrl_forward(const Ctrl *cin, Ctrl *cout, int nwords) {
    const unsigned long long *src = (const unsigned long long *)cin;
    unsigned long long *dst = (unsigned long long *)cout;
#if SSF_CTRL_WIDE
    constexpr int kMax = (int)(sizeof(Ctrl) / 8);
    unsigned long long w[kMax];
#pragma unroll
    for (int q = 0; q < kMax; ++q)
        if (q < nwords) w[q] = src[q];
#pragma unroll
    for (int q = 0; q < kMax; ++q)
        if (q < nwords) dst[q] = w[q];
#else
    int i = 0;
    for (; i + 8 <= nwords; i += 8) {
        unsigned long long w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = src[i + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[i + q] = w[q];
    }
    for (; i < nwords; ++i) dst[i] = src[i];
#endif
}

// Control-block handling of a Manakov Row launch: evaluates the convergence sums the last column stage left
// (part: this thread's first two entries of each, fetched by the caller before its row), decides final / redo,
// derives a new step size and operator when needed, and lets the lead thread write the next control block.
// Returns false if this launch has nothing to transform; lo receives the operator.  Uses the first 4 KiB of LDS.
template <typename T, class Ctx>
SSF_HD bool row_ctrl(Ctx &ctx, const RowArgs<T> &a, const double (&part)[2][4], LinOp &lo) {
    LinOp *lsh = (LinOp *)ctx.lds;
    // Only the fields used here are read (scalar loads); the lead thread forwards the block
    // word by word and patches what changed.  A private copy of the whole struct makes the
    // compiler fetch the pass-through fields with a vector load, and waiting for that one
    // means waiting for the row (in-order vmcnt).
    const Ctrl &c = *a.cin;
    const int c_state = c.state, c_it = c.it, c_pend0 = c.pend0, c_pendn = c.pendn, c_cap0 = c.cap0, c_bound0 = c.bound0;
    int n_exact0 = c.exact0;
    const int c_gscale = c.gscale, c_sparse = c.t_sparse;
    int n_state = c_state, n_final = c.final_, n_cap0 = c_cap0, n_hzv = c.hz_valid;
    int add_nonconv = 0, add_ahead = 0;
    const long long c_nonconv = c.nonconv, c_ahead = c.n_ahead;      // (read before the block is forwarded: see mk_col_stage)
    double n_hz = c.hz;
    double *red = (double *)(ctx.lds) + 64;
    const bool lead = ctx.bid == 0 && ctx.tid == 0;
    bool act = c_state == ST_AFTER_S || c_state == ST_ROW_ITER || c_state == ST_RECOVER_ROW;
    if (c_pend0 || c_pendn) {             // every block reduces the sums in the same order
        double s0 = part[0][0] + part[1][0], s1 = part[0][1] + part[1][1];
        double s2 = part[0][2] + part[1][2], s3 = part[0][3] + part[1][3];
        for (int i = ctx.tid + 2 * ctx.nthreads; i < a.npart; i += ctx.nthreads) {
            s0 += a.pnum0[i];
            s1 += a.pden0[i];
            s2 += a.pnum[i];
            s3 += a.pden[i];
        }
        if (c_pend0) block_sum2(ctx, s0, s1, red);
        if (c_pendn) block_sum2(ctx, s2, s3, red);
        ctx.sync();
        bool redo = false;
        if (c_pend0) {                                                // lim_0 (channels.py:424, 517-519)
            const double lim0 = sqrt(s0) / sqrt(s1);
            if (lead) {
                const long long idx = c.pend0_idx;
                if (idx < a.k.trace_cap && a.k.tr_lim) a.k.tr_lim[idx * a.k.maxIter] = lim0;
            }
            if (c_cap0) {
                if (!(lim0 < a.k.tol)) add_nonconv = 1;
                n_cap0 = 0;
            } else if (c_pendn && lim0 < a.k.tol) {
                redo = true;                                          // converged at iterate 0 after all ...
                if (c_bound0) n_exact0 = 1;                           // ... or the bound could not exclude it: measure it
            }
        }
        if (c_pendn) {                                                // lim_it, known before iterate it exists
            if (redo) {
                n_state = (n_exact0 && c_sparse) ? ST_RECOVER_A : ST_REDO0;   // (exact lim_0 needs the whole field at the step start)
                act = false;
            } else {
                const double lim = sqrt(s2) / sqrt(s3);
                if (lead) {
                    const long long tn = c.trace_n;
                    if (tn < a.k.trace_cap && a.k.tr_lim) a.k.tr_lim[tn * a.k.maxIter + c_it] = lim;
                }
                const bool conv = lim < a.k.tol;
                n_final = conv || c_it == a.k.maxIter - 1;            // channels.py:429-434
                if (n_final && !conv) add_nonconv += 1;
                add_ahead = 1;
            }
        }
    }
    const bool new_lin = act && !n_hzv;
    if (new_lin) {                        // new step size: every block derives the same hz / operator
        double mx = 0.0;
        if (a.k.adaptive) mx = global_max(ctx, a.pmax, a.npart, red);
        ctx.sync();
        if (ctx.tid == 0) {
            const double hz = pick_hz(a.k, c.z, mx);
            lsh[0] = make_linop(hz / 2, a.k.lin_a, a.k.lin_b, a.k.w2, a.k.invN, a.k.log2N);
            ((double *)(lsh + 1))[0] = hz;
        }
        ctx.sync();
        n_hz = ((double *)(lsh + 1))[0];
        lo = lsh[0];
        if (lead) a.cout->lin = lsh[0];   // (straight from LDS: callers that only need cth / mag keep no copy)
        n_hzv = 1;
        ctx.sync();
    } else if (act && c_state == ST_RECOVER_ROW) {                    // E(z) = Lin^-1(E_hd): the half step backwards (rare: once per span)
        ctx.sync();
        if (ctx.tid == 0) lsh[0] = make_linop(-c.hz / 2, a.k.lin_a, a.k.lin_b, a.k.w2, a.k.invN, a.k.log2N);
        ctx.sync();
        lo = lsh[0];
        ctx.sync();
    } else if (act) {
        lo = c.lin;
    }
    if (act && c_state == ST_AFTER_S && c_gscale) lo.mag *= (double)row_n1(a);           // (G = spectrum / N1: mk_col_stage)
    if (act) n_state = c_state == ST_AFTER_S ? ST_NEED_H : c_state == ST_RECOVER_ROW ? ST_RECOVER_B : ST_NEED_I;
    if (lead) {
        ctrl_forward(a.cin, a.cout, (int)((new_lin ? offsetof(Ctrl, lin) : sizeof(Ctrl)) / 8));
        Ctrl *n = a.cout;
        n->state = n_state;
        n->final_ = n_final;
        n->pend0 = 0;
        n->pendn = 0;
        n->cap0 = n_cap0;
        n->bound0 = 0;
        n->exact0 = n_exact0;
        if (n_exact0) n->dense = 1;
        if (act && c_state == ST_AFTER_S) n->gscale = 0;
        n->hz_valid = n_hzv;
        n->hz = n_hz;
        n->nonconv = c_nonconv + add_nonconv;
        n->n_ahead = c_ahead + add_ahead;
    }
    return act;
}

// same for any N (fftfreq sign convention: bins 0 .. (N+1)/2 - 1 are non-negative)
template <typename T> SSF_HD cx<T> lin_at_n(const LinOp &lo, long long k, long long N) {
    const double kk = (double)(k < (N + 1) / 2 ? k : k - N);
    double s, c;
    cis_rad_d(lo.cth * kk * kk, c, s);
    return mk<T>((T)(lo.mag * c), (T)(lo.mag * s));
}

// Row stage for row lengths with factors 3 and 5 (mixed_fft.h): one row per workgroup, the row lives in LDS
// (after the 4 KiB the control logic uses), G -> forward transform -> x linear operator -> inverse -> G.
// p: the pass plan, indexed at run time (p.r[i], p.S[i] ...).  It must be the plan IN THE KERNEL ARGUMENTS (scalar loads with a computed
// offset): `a` is usually unit_view()'s modified COPY of the arguments, and a run-time index into a local copy puts the whole copy
// into scratch memory -- 420 bytes per lane and a memory round trip for every p.S[i] (round 6: SQ_INSTS_VMEM_RD 139 per wave
// against 34 in the radix-2^n rows; profiles/r6_mixed_rows.txt).  The wrappers pass the kernel parameter's own member.
template <typename T, class Ctx> SSF_HD void row_mixed_body(Ctx &ctx, const RowArgs<T> &a, const MixPlan &p) {
    const int L = a.N2, R_ = a.rows_per_wg, T_ = ctx.nthreads / R_;
    const int f = ctx.tid / T_, t = ctx.tid - f * T_;      // row within the workgroup, thread within the row
    cx<T> *x = (cx<T> *)(ctx.lds + 4096) + (size_t)f * L;
    const long long rr = (long long)ctx.bid * R_ + f;      // global row index (grid is exact)
    cx<T> *g = a.G + rr * L;
    const cx<T> *gin = a.src ? a.src + rr * L : g;
    LinOp lo;
    double part[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#if SSF_EARLY_BASES
    // the operator constant as the previous launch left it (a scalar load, ahead of everything: the vector loads of the whole
    // operator inside row_ctrl queue up behind the row): the thread's operator bases are made from it while the row is in flight,
    // and made again in the (rare) launch that derives a new operator
    const double cth_pre = a.use_ctrl ? a.cin->lin.cth : (a.lin ? a.lin->cth : 0.0);
#endif
    if (a.use_ctrl) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = ctx.tid + u * ctx.nthreads;
            if (i < a.npart) {
                part[u][0] = a.pnum0[i];
                part[u][1] = a.pden0[i];
                part[u][2] = a.pnum[i];
                part[u][3] = a.pden[i];
            }
        }
        ctx.issue_fence();
    }
    ctx.mark(0);
    const int N1 = (int)row_n1(a);
    const int k1 = a.N1mix ? (int)(rr % N1) : (int)(rr & (N1 - 1));
    // x linear operator: applied between the two butterflies of the stride-1 pass (mixed_fft.h: mix_pass_mid, mix_apply_op), which
    // holds the spectrum in runs of bins N / R apart.  (As a pass of its own over the row in LDS it was 6 of the 30 us of
    // a launch at rows of 3750: one read-modify-write per bin, each waiting for the one before it.)
    MixRowOp op;
    // The row is staged into LDS with the control logic behind its loads, and goes from the last pass's butterflies straight back to
    // global memory (mix_dif_op_dit; SSF_MIX_IO).
    auto ctrl = [&]() {
        if (a.use_ctrl) {
            ctx.issue_fence();
            if (!row_ctrl(ctx, a, part, lo)) return false;
        } else {
            lo = *a.lin;
        }
        const int Rl = p.r[p.npass - 1];
        op.cth = lo.cth;
        op.mag = lo.mag;
        op.D = (double)(a.N / Rl);
        double c, s;
        cis_rad_d(2.0 * lo.cth * op.D * op.D, c, s);
        op.c2 = mk<double>(c, s);
        op.k1 = k1;
        op.N1 = N1;
        op.N = a.N;
        return true;
    };
    // (Butterflies are dealt to the threads of a row from number 0 up, so a pass with a large radix works on the first waves only --
    //  3125 = 25 x 25 x 5: 125 butterflies for 256 threads.  Dealing them from the middle of the row in the second workgroup of a CU
    //  changes nothing, measured: the hardware already puts wave k of the two workgroups of a CU on different SIMDs -- appendix #54.)
    if (!mix_dif_op_dit(ctx, p, t, T_, x, a.wtab, op, gin, g, ctrl)) return;
    ctx.mark(4);
    ctx.mark(5);
    ctx.flush(0);
}

// The row stage's G stores.  A launch ends with up to 32 MiB of dirty lines in the eight L2s, which the end-of-kernel release
// writes back before the next launch can start; stored write-through (sc0 sc1) the lines leave the L2 while the other workgroups
// still compute: row launch 23.2 -> 21.4 us at config 2, +4 % steps/s (profiles/r3_ab_runs.txt; the column stage's G and field
// stores gain nothing that way).  The stores are buffer stores with the cache policy in the instruction's aux bits
// (__builtin_amdgcn_raw_buffer_store_*), so the compiler sees them: it counts them in vmcnt and keeps the store-data hazard
// of a 128-bit store (two wait states before a VALU instruction may overwrite the data registers on gfx940+).  Round 3 issued
// them by inline assembly, which the hazard recogniser does not look into: where the data sat in a temporary register tuple
// that the next value's moves overwrote at once -- the packed complex64 kernel -- the store sent the NEXT element's data
// (rel-L2 0.67 at every size); the double-precision kernel's values happened to live in aligned tuples of their own and
// passed every test.  SSF_WT_ROWS: bit 0 double, bit 1 packed pairs, bit 2 float.  Round 4, compiler-visible stores, same box
// (profiles/r4_ab_twiddle_tables.txt): double 24.1 -> 22.5 us per row launch, packed pairs 47.7 -> 45.5 us with correct results.
#ifndef SSF_WT_ROWS
#define SSF_WT_ROWS 3
#endif
#ifndef SSF_LOAD_ORDER
#define SSF_LOAD_ORDER 1      // loads of a row in the order its first butterfly consumes them: + 0.5 ... 1.3 %, 8 of 8 (profiles/r5_ab_load_order.txt)
#endif
#ifndef SSF_EARLY_BASES
#define SSF_EARLY_BASES 0
#endif
#ifndef SSF_SPEC_G
#define SSF_SPEC_G 0
#endif
// Experiments of round 6 on the double-precision Manakov column stage (appendix #48, #49):
//  SSF_PAIR_DPP   the x / y partner threads of a sample sit in NEIGHBOURING lanes (pol = tid & 1) and swap powers, rotations and
//                 |d rot|^2 with DPP moves (ctx.xchg) instead of through LDS: no LDS traffic and no barrier for the pair exchange
//  SSF_THETA_F32  the phase array of the latest rotation is kept in single precision: it only enters the convergence measure
//                 (4 sin^2((theta_new - theta_old) / 2), relative error ~1e-8 of lim), never the field -- 4 instead of 8 bytes
//                 read and written per sample and iteration
#ifndef SSF_PAIR_DPP
#define SSF_PAIR_DPP 0
#endif
#ifndef SSF_THETA_F32
#define SSF_THETA_F32 0
#endif
template <typename T> constexpr bool wt_rows() {
    return sizeof(scalar_t<T>) == 8 ? (SSF_WT_ROWS & 1) != 0 : sizeof(T) == 8 ? (SSF_WT_ROWS & 2) != 0 : (SSF_WT_ROWS & 4) != 0;
}
template <int V, typename T, class Ctx>
SSF_HD void row_store(Ctx &ctx, const RowArgs<T> &a, const PassPlan &p, int f, int b, const cx<T> *v) {
    const int fpw = ctx.nthreads / p.tpf;
    cx<T> *base = a.G + (((long long)ctx.bid * fpw) << a.log2N2);       // first row of this workgroup (wave-uniform)
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (wt_rows<T>()) {
        const unsigned bytes = (unsigned)(((size_t)fpw << a.log2N2) * sizeof(cx<T>));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes, 0x00020000);
        constexpr int kWT = 0x11;                                         // aux: sc0 | sc1 (system-scope write-through)
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const unsigned off = (unsigned)(((f << a.log2N2) + b + p.tpf * q) * (int)sizeof(cx<T>));
            if constexpr (sizeof(cx<T>) == 16) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 w;
                __builtin_memcpy(&w, &v[q], 16);
                __builtin_amdgcn_raw_buffer_store_b128(w, rs, off, 0, kWT);
            } else {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                u2 w;
                __builtin_memcpy(&w, &v[q], 8);
                __builtin_amdgcn_raw_buffer_store_b64(w, rs, off, 0, kWT);
            }
        }
        return;
    }
#endif
    cx<T> *g = base + ((long long)f << a.log2N2);
#pragma unroll
    for (int q = 0; q < V; ++q) g[b + p.tpf * q] = v[q];
}

// LG > 0: row length fixed at compile time (index math folds to immediates); 0: runtime
// V = values per thread: 16 (256 registers, two waves per SIMD) or 8 (128 registers, four waves per SIMD: while the waves
// of one workgroup wait for their row or for their stores to drain, the other workgroup of the CU has the SIMDs)
template <typename T, int LG, int V = 16, class Ctx> SSF_HD void row_body(Ctx &ctx, const RowArgs<T> &a) {
    constexpr int lgV = V == 16 ? 4 : 3;
    cx<T> *lds = (cx<T> *)ctx.lds;
    LinOp lo;
    const PassPlan p = make_plan(LG > 0 ? LG : a.log2N2, lgV);
    const int fpw = ctx.nthreads / p.tpf;                  // row transforms per workgroup
    const int f = ctx.tid / p.tpf, b = ctx.tid % p.tpf;
    const long long rr = (long long)ctx.bid * fpw + f;     // global row-transform index (grid is exact)
    cx<T> *g = a.G + (rr << a.log2N2);
    cx<T> v[V];
    ctx.mark(0);
    // Issue order matters (vmcnt retires in order): first the convergence sums the last column
    // stage may have left (fetched unconditionally, they are only used if the control block says
    // they are pending), then the row itself.  The control block (scalar loads) and the
    // reduction of the sums are then evaluated while the row is still in flight; a launch that
    // turns out to have nothing to do just drops the row.  (Fetching the sums after the row
    // makes the decision wait for the row: +2.5 us per launch.  Starting half of the workgroups
    // late, so that co-resident workgroups are out of phase, was measured and does not help.)
    double part[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#if SSF_EARLY_BASES
    // the operator constant as the previous launch left it (a scalar load, ahead of everything: the vector loads of the whole
    // operator inside row_ctrl queue up behind the row): the thread's operator bases are made from it while the row is in flight,
    // and made again in the (rare) launch that derives a new operator
    const double cth_pre = a.use_ctrl ? a.cin->lin.cth : (a.lin ? a.lin->cth : 0.0);
#endif
    if (a.use_ctrl) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = ctx.tid + u * ctx.nthreads;
            if (i < a.npart) {
                part[u][0] = a.pnum0[i];
                part[u][1] = a.pden0[i];
                part[u][2] = a.pnum[i];
                part[u][3] = a.pden[i];
            }
        }
        ctx.issue_fence();
    }
#if SSF_LOAD_ORDER
    // Experiment (appendix #43): the row's loads issued in the order the first butterfly consumes them (bit-reversed: dft16 starts
    // with v[0] + v[8], v[4] + v[12], ...), in groups the scheduler may not merge, so that the waits can be per group and the first
    // additions start while the rest of the row is still in flight.
    {
        constexpr int ord16[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
        constexpr int ord8[8] = {0, 4, 2, 6, 1, 5, 3, 7};
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int q = V == 16 ? ord16[i] : ord8[i & 7];
            v[q] = g[b + p.tpf * q];
            if ((i & 3) == 3) ctx.issue_fence();
        }
    }
#else
#pragma unroll
    for (int q = 0; q < V; ++q) v[q] = g[b + p.tpf * q];
#endif
    // single precision: the next-to-last pass takes its hi + lo factors from a table in LDS, built while the row is in flight (TwSrc)
    constexpr bool kTab = (SSF_TW_TAB & 1) && sizeof(scalar_t<T>) == 4;
    TwSrc<T> tws;
    if (kTab) {
        tw_entry_t<T> *tab = (tw_entry_t<T> *)(ctx.lds + a.tw_off);
        tw_lds_build<T>(ctx, p, tab);
        tws.lds = tab;
    }
    if (a.use_ctrl) {
        ctx.issue_fence();
        if (!row_ctrl(ctx, a, part, lo)) return;
    } else if (a.lin) {
        lo = *a.lin;
    }
    if (kTab && p.npass == 2) ctx.sync();                   // (three passes: the first exchange's barrier publishes the table)
    const int N1 = 1 << a.log2N1, log2N = a.log2N1 + a.log2N2;
    const int k1 = (int)(rr & (N1 - 1));
    cx<T> *l = lds + (size_t)f * lds_slots_per_fft(p.L);
    ctx.mark(1);
    // Issue priority by phase (s_setprio): a wave that is further behind in its launch gets the VALU first.  The two workgroups
    // of a CU start together, but the older one wins every arbitration, runs through at full speed and leaves the younger one
    // to finish alone, one wave per SIMD (phase stamps: first half of the row grid done at 13.3 us, second half at 17.5 us).
    // With the forward transform above the inverse one (rows) and inverse > time domain > forward (columns) the late
    // workgroup catches up while the early one is in a later phase: +3 % steps/s at config 2 (profiles/r3_ab_runs.txt);
    // off (a.prio = 0) when several plans share the GPU, where it costs 4 % (profiles/r3_lanes_prio_wt.txt).
    if (a.prio) ctx.template setprio<2>();
    const int last = p.npass - 1;
#if SSF_EARLY_BASES
    // Everything that depends on the thread's position only -- the twiddle bases of all passes (one sincospi each; the inverse
    // transform conjugates them) and the two sines / cosines of the linear operator -- is evaluated HERE, while the row is still
    // in flight and the VALU has nothing else to do (appendix #44).
    Lin16Bases lb{};
    const bool lin16 = (a.use_ctrl || a.lin) && p.lg(last) == lgV;
    {
#pragma unroll
        for (int i = 0; i < kTwMaxPass; ++i) {
            if (i < p.npass && p.lgLn(i) > 0 && !(kTab && i == p.npass - 2)) {
                tws.base[i] = tw_base(+1, pass_j(p, i, b), pass_lgLi(p, i));
                tws.have[i] = true;
            }
        }
        if (lin16) lb = lin16_bases<V>(cth_pre, k1 + ((long long)rev_pos(p, reg_pos(p, last, b, 0)) << a.log2N1), log2N);
        ctx.issue_fence();
    }
#endif
    fft_dif<-1, V, kTab>(ctx, p, b, v, l, tws);
    ctx.mark(2);
    if (a.prio) ctx.template setprio<1>();
    // registers now hold pass-(p-1) positions; bin k = k1 + N1 * rev(pos)
    if (!a.use_ctrl && !a.lin) {                 // fixed-kernel convolution: multiplier array in this kernel's spectrum order
        if (a.fwd_only) {
#pragma unroll
            for (int idx = 0; idx < V; ++idx) g[reg_pos(p, last, b, idx)] = v[idx];
            return;
        }
        const cx<T> *h = a.harr + ((size_t)k1 << a.log2N2);
#pragma unroll
        for (int idx = 0; idx < V; ++idx) v[idx] = v[idx] * h[reg_pos(p, last, b, idx)];
    } else if (p.lg(last) == lgV) {
#if SSF_EARLY_BASES
        if (lo.cth != lb.cth) lb = lin16_bases<V>(lo.cth, k1 + ((long long)rev_pos(p, reg_pos(p, last, b, 0)) << a.log2N1), log2N);
        apply_lin16<V>(lo, lb, v);
#else
        const long long k0 = k1 + ((long long)rev_pos(p, reg_pos(p, last, b, 0)) << a.log2N1);
        apply_lin16<V>(lo, k0, log2N, v);
#endif
    } else {
#pragma unroll
        for (int idx = 0; idx < V; ++idx) {
            const long long k = k1 + ((long long)rev_pos(p, reg_pos(p, last, b, idx)) << a.log2N1);
            v[idx] = mul_by_d(v[idx], lin_at(lo, k, log2N));
        }
    }
    ctx.mark(3);
    fft_dit<+1, V, kTab>(ctx, p, b, v, l, tws);
    if (a.prio) ctx.template setprio<0>();
    ctx.mark(4);
    row_store<V>(ctx, a, p, f, b, v);
    ctx.mark(5);
    ctx.flush(0);
}

// --------------------------------------------------------------------------- column kernel
enum {
    CM_NLSE_FIRST = 0,   // time -> forward -> G
    CM_NLSE_STEP = 1,    // G -> inverse -> E *= exp(j g_hz |E|^2) -> forward -> G
    CM_NLSE_LAST = 2,    // G -> inverse -> time
    CM_MK = 3,           // Manakov column stage picked by the Ctrl state (S | H | I | rebuild of iterate 0)
    CM_PLAIN_FWD = 5,    // time -> forward -> G            (linear channel)
    CM_PLAIN_INV = 6     // G -> inverse -> time            (linear channel)
};

template <typename T> struct ColArgs {
    cx<T> *G;                 // (nrows, N1, N2)
    cx<T> *T0, *T1;           // time-domain fields, (nrows, N); Manakov: field at the step start / end (ping-pong)
    cx<T> *Ehd;               // (nrows, N)
    scalar_t<T> *P;           // (2, K, N): Pch of the current / next step
    scalar_t<T> *Theta;       // (K, N): phase shz * phi of the latest rotation
    int log2N1, log2N2, npol, mode;
    int ngroups;              // Manakov: polarisation pairs K (P holds 2 x K x N values: two buffers)
    scalar_t<T> g_hz;         // NLSE: gamma * hz
    const Ctrl *cin;
    Ctrl *cout;
    MkConst k;
    double *pmax, *pnum, *pden, *pnum0, *pden0;
    int npart;                // number of column workgroups (partials per array)
    int N2;                   // row length when it is not 1 << log2N2 (mixed-radix rows), else 0
    long long N;              // N1 * N2 in that case
    int vpt;                  // values per thread: 16 (0 = 16) or 8 (128-register kernels, four waves per SIMD)
    long long u_elems;        // independent units (see RowArgs): field elements per unit; P / Theta advance by 2 / 1 x ngroups x N
    int u_part;
    int prio;                 // see RowArgs
    int sg;                   // Manakov: stage groups the launched kernel carries (0 = SG_ALL); read by the launcher only
    // column lengths with factors 3 / 5 (col_mixed_body): N = N1mix x N2, the tile (mix_cols columns x N1mix, both polarisations)
    // lives in LDS and is transformed by the mixed-radix passes of mixed_fft.h
    int N1mix, mix_cols;      // column length (0: 1 << log2N1), columns per workgroup (a power of two)
    MixPlan plan1;            // pass plan of a column
    const cx<double> *wtab1;  // cis(-2 pi k / N1mix), k < N1mix
};

// Arguments of unit u of a batch of independent units: every pointer moved to the unit's block.  The kernel bodies never
// know: a unit has its own rows, control blocks, partial sums, step sizes and convergence decisions.
template <typename T> SSF_HD RowArgs<T> unit_view(const RowArgs<T> &a, int u) {
    // No `if (u > 0)` around this: a branch here cuts the kernel's argument loads into two dependent round trips in front of
    // the first row load, and the launch's start-up is latency, chip-wide (every workgroup starts at once, scalar cache cold):
    // with the branch and the (experiment-only) stagger test gone the row stage's prologue is 51 instructions and ONE wait
    // instead of 119 and three: + 1 - 2 % steps/s for fields of 2^12 ... 2^18 samples, within the noise at 2^20
    // (profiles/r3_chained_launches_and_stagger.txt).
    RowArgs<T> b = a;
    b.G += (long long)u * a.u_elems;
    if (a.cin) b.cin += u;
    if (a.cout) b.cout += u;
    const long long po = (long long)u * a.u_part;
    if (a.pmax) b.pmax += po;
    if (a.pnum) b.pnum += po;
    if (a.pden) b.pden += po;
    if (a.pnum0) b.pnum0 += po;
    if (a.pden0) b.pden0 += po;
    return b;
}
template <typename T> SSF_HD ColArgs<T> unit_view(const ColArgs<T> &a, int u) {
    // (Kept with its branch, unlike the row stage's: branch-free, the 128-register column kernels spill 12 registers instead
    // of 4 and the small fields lose 1 - 2 %; nothing measurable at 2^20.)
    ColArgs<T> b = a;
    if (u > 0) {
        const long long fo = (long long)u * a.u_elems;
        b.G += fo;
        if (a.T0) b.T0 += fo;
        if (a.T1) b.T1 += fo;
        if (a.Ehd) b.Ehd += fo;
        const long long N = a.N2 ? a.N : 1ll << (a.log2N1 + a.log2N2);
        if (a.P) b.P += 2 * N * a.ngroups * (long long)u;
        if (a.Theta) b.Theta += N * a.ngroups * (long long)u;
        if (a.cin) b.cin += u;
        if (a.cout) b.cout += u;
        const long long po = (long long)u * a.u_part;
        if (a.pmax) b.pmax += po;
        if (a.pnum) b.pnum += po;
        if (a.pden) b.pden += po;
        if (a.pnum0) b.pnum0 += po;
        if (a.pden0) b.pden0 += po;
    }
    return b;
}

// Thread geometry of the column kernel.  A workgroup owns C adjacent columns of one field
// group (Manakov: one polarisation pair; the x row is handled by the first half of the
// threads, the y row by the second half; NLSE: a single row, no split).
// RAGGED: the row length N2 is not a power of two (mixed-radix rows); the last tile of a row is then
// only partly filled, and its surplus threads (valid == false) load zeros and store nothing.
template <typename T, int LG, class Ctx, bool RAGGED = false, int V = 16> struct ColGeom {
    static constexpr int kV = V;
    static constexpr bool kDppT = SSF_PAIR_DPP && V == 16 && sizeof(T) == 8;    // (the polarisation-split double-precision kernels)
    bool dpp;                 // the partner thread is the neighbouring lane (Manakov stage only)
    PassPlan p;
    int half, pol, t, C, c, b, n2, N2;
    bool valid;
    long long N;
    long long rowbase;        // element offset of this thread's row
    long long pbase;          // element offset of the pair's row in P
    SSF_HD ColGeom(Ctx &ctx, const ColArgs<T> &a) {
        p = make_plan(LG > 0 ? LG : a.log2N1, V == 16 ? 4 : 3);
        half = ctx.nthreads / a.npol;
        dpp = kDppT && a.npol == 2;
        if (kDppT && a.npol == 2) {
            pol = ctx.tid & 1;
            t = ctx.tid >> 1;
        } else {
            pol = ctx.tid / half;
            t = ctx.tid - pol * half;
        }
        C = half / p.tpf;
        c = t % C;
        b = t / C;
        N2 = RAGGED ? a.N2 : 1 << a.log2N2;
        const int tpp = RAGGED ? (N2 + C - 1) / C : N2 / C;      // tiles per field group
        const int grp = ctx.bid / tpp;
        int tile = ctx.bid - grp * tpp;
        // Workgroups go round-robin over the 8 XCDs (each with its own L2).  Neighbouring tiles share the
        // 128-B lines of the real-valued P / Theta arrays (C = 8 columns are 64 B), so tiles are dealt out
        // in contiguous runs per XCD: the second half of such a line is then an L2 hit, not a second fetch
        // (+1.2 %).  (Laying P / Theta out tile-major instead, one contiguous 16 KiB block per workgroup,
        // is slower: -1.5 %, the block sits in one memory channel.)
        if (tpp % 8 == 0) tile = (tile & 7) * (tpp >> 3) + (tile >> 3);
        else if (RAGGED && tpp > 8) {
            // The same dealing for any number of tiles (row lengths that are not powers of two: 3125 -> 391 tiles): XCD x gets the
            // run of q + (x < r) tiles that starts at x q + min(x, r).  Here it matters more than for P / Theta: a row of such a length
            // does not start on a 128-byte boundary, so EVERY 128-byte segment of a tile straddles two cache lines, each shared with a
            // neighbouring tile -- dealt round-robin, two XCDs fetched every line (800 000 = 256 x 3125: 91 MB read per launch where
            // the stage needs 46; profiles/r6_traffic_lengths.txt).
            const int q = tpp >> 3, r = tpp & 7, x = tile & 7;
            tile = x * q + (x < r ? x : r) + (tile >> 3);
        }
        n2 = tile * C + c;
        valid = !RAGGED || n2 < N2;
        N = RAGGED ? a.N : 1ll << (a.log2N1 + a.log2N2);
        rowbase = (long long)(grp * a.npol + pol) * N;
        pbase = (long long)grp * N;
    }
    // frequency side: register q <-> k1 = b + tpf*q (pass-0 positions)
    SSF_HD long long freq_off(int q) const { return (long long)(b + p.tpf * q) * N2 + n2; }
    // time side: register idx <-> n1 = rev(position in the last pass)
    SSF_HD long long time_off(int idx) const {
        return (long long)rev_pos(p, reg_pos(p, p.npass - 1, b, idx)) * N2 + n2;
    }
    // guarded accesses (the guard disappears when the row length is a power of two)
    template <typename W> SSF_HD W ld(const W *ptr, long long i) const {
        if (RAGGED && !valid) return W{};
        return ptr[i];
    }
    template <typename W> SSF_HD void st(W *ptr, long long i, W x) const {
        if (!RAGGED || valid) ptr[i] = x;
    }
};

// inter-pass twiddle of the N = N1*N2 decomposition, applied on the frequency side of the
// column kernel (which is HBM-bound and has VALU head-room; the row kernel is VALU-bound):
// register q holds k1 = b + tpf*q of column n2  ->  v[q] *= cis(SIGN * 2 pi n2 k1 / N)
struct GTw {                      // the two bases of a thread's inter-pass twiddles, cis(+2 pi n2 b / N) and cis(+2 pi n2 tpf / N)
    cx<double> w0, ws;
    bool have = false;            // SSF_TW_REUSE: the inverse transform's bases are kept, the forward one conjugates them
};
template <int SIGN, bool RAGGED, typename T, class G> SSF_HD void global_twiddle(const G &g, int log2N, cx<T> *v, GTw &gt) {
    cx<double> w0, ws;
    if (SSF_TW_REUSE && gt.have) {
        w0 = SIGN > 0 ? gt.w0 : conj(gt.w0);
        ws = SIGN > 0 ? gt.ws : conj(gt.ws);
    } else {
        if (RAGGED) {                        // N = N1 * N2 with N2 not a power of two: the fraction is rounded once
            w0 = cis2pi<double>((double)SIGN * ((double)(((long long)g.n2 * g.b) % g.N) / (double)g.N));
            ws = cis2pi<double>((double)SIGN * ((double)(((long long)g.n2 * g.p.tpf) % g.N) / (double)g.N));
        } else {
            const long long N = 1ll << log2N;
            w0 = cis2pi<double>((double)SIGN * scale_pow2((double)(((long long)g.n2 * g.b) & (N - 1)), log2N));
            ws = cis2pi<double>((double)SIGN * scale_pow2((double)(((long long)g.n2 * g.p.tpf) & (N - 1)), log2N));
        }
        if (SSF_TW_REUSE) {
            gt.w0 = SIGN > 0 ? w0 : conj(w0);
            gt.ws = SIGN > 0 ? ws : conj(ws);
            gt.have = true;
        }
    }
    // t[q] = w0 * ws^q by doubling: ws^2, ws^4, ws^8, then t[q + 2^k] = t[q] * ws^(2^k) -- 18 complex products of depth <= 7
    // (a power tree of ws followed by w0 * ws^q costs 30); rounded once, where it is applied (see tw_powers)
    constexpr int V = G::kV;
    cx<double> t[V];
    const cx<double> s2 = ws * ws, s4 = s2 * s2;
    t[0] = w0;
    t[1] = w0 * ws;
    t[2] = t[0] * s2;
    t[3] = t[1] * s2;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[4 + q] = t[q] * s4;
    if constexpr (V == 16) {
        const cx<double> s8 = s4 * s4;
#pragma unroll
        for (int q = 0; q < 8; ++q) t[8 + q] = t[q] * s8;
    }
#pragma unroll
    for (int q = 0; q < V; ++q) v[q] = mul_by_d(v[q], t[q]);
}

// exchange V per-thread values with the partner thread (same column/butterfly, other
// polarisation) through LDS scratch `sh` (2*16*half values); one barrier inside.
// The caller guarantees (barrier) that `sh` is free on entry.
template <typename U, class Ctx, class G> SSF_HD void pair_swap(Ctx &ctx, const G &g, const U *mine, U *other, U *sh) {
    constexpr int V = G::kV;
    if constexpr (G::kDppT) {
        if (g.dpp) {
#pragma unroll
            for (int idx = 0; idx < V; ++idx) other[idx] = (U)ctx.xchg((double)mine[idx]);
            return;
        }
    }
#pragma unroll
    for (int idx = 0; idx < V; ++idx) sh[(size_t)(g.pol * V + idx) * g.half + g.t] = mine[idx];
    ctx.sync();
#pragma unroll
    for (int idx = 0; idx < V; ++idx) other[idx] = sh[(size_t)((g.pol ^ 1) * V + idx) * g.half + g.t];
}

// The two threads of a polarisation pair hold the same 16 time samples (x row | y row).  Every
// per-sample quantity that is common to both polarisations (phase, rotation, |d rot|^2) is
// evaluated once, by the sample's owner: the x thread owns registers 0 .. V/2-1, the y thread V/2 .. V-1.
// own_idx(j) = register of this thread's j-th owned sample, oth_idx(j) = the partner's.
template <class G> SSF_HD int own_idx(const G &g, int j) { return g.pol ? j + G::kV / 2 : j; }
template <class G> SSF_HD int oth_idx(const G &g, int j) { return g.pol ? j : j + G::kV / 2; }
template <class G> SSF_HD long long own_time_off(const G &g, int j) { return g.pol ? g.time_off(j + G::kV / 2) : g.time_off(j); }
// pick, for register idx (compile-time), the owner's or the partner's value
template <typename U, class G> SSF_HD U pick16(const G &g, int idx, const U *own, const U *oth) {
    constexpr int H = G::kV / 2;
    return ((idx / H) == g.pol) ? own[idx % H] : oth[idx % H];
}

// rot[idx] = cis(ang) for all 16 registers from the 8 phases this thread owns; the partners
// swap their halves through LDS scratch `sh` (16*half complex).  One barrier inside; the caller
// guarantees `sh` is free on entry.
template <typename T, class Ctx, class G>
SSF_HD void pair_cis(Ctx &ctx, const G &g, const T *ang_own, cx<T> *rot, cx<T> *sh) {
    constexpr int V = G::kV, H = V / 2;
    cx<T> own[H], oth[H];
    if constexpr (G::kDppT) {
        if (g.dpp) {
#pragma unroll
            for (int j = 0; j < H; ++j) {
                own[j] = cis_t<T>(ang_own[j]);
                oth[j] = mk<T>((T)ctx.xchg((double)own[j].re), (T)ctx.xchg((double)own[j].im));
            }
#pragma unroll
            for (int idx = 0; idx < V; ++idx) rot[idx] = pick16(g, idx, own, oth);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
        own[j] = cis_t<T>(ang_own[j]);
        sh[(size_t)own_idx(g, j) * g.half + g.t] = own[j];
    }
    ctx.sync();
#pragma unroll
    for (int j = 0; j < H; ++j) oth[j] = sh[(size_t)oth_idx(g, j) * g.half + g.t];
#pragma unroll
    for (int idx = 0; idx < V; ++idx) rot[idx] = pick16(g, idx, own, oth);
}

// ---- Manakov time-domain building blocks (registers v = this thread's 16 samples of its row) ----
// step start (channels.py:388-395): Pch = |Ex|^2 + |Ey|^2 -> Pbuf, block max of phi -> pmax
template <typename T, class Ctx, class G>
SSF_HD void mk_step_start(Ctx &ctx, const G &g, const ColArgs<T> &a, const cx<T> *v, T *Pbuf, bool lds_busy) {
    constexpr int V = G::kV;
    T *shT = (T *)ctx.lds;
    T mine[V], oth[V];
#pragma unroll
    for (int idx = 0; idx < V; ++idx) mine[idx] = norm2(v[idx]);
    if (lds_busy && !(G::kDppT && g.dpp)) ctx.sync();
    pair_swap(ctx, g, mine, oth, shT);
    const T c8g = (T)a.k.c8g;
    double m = -INFINITY;
#pragma unroll
    for (int idx = 0; idx < V; ++idx) {
        const T ax = g.pol ? oth[idx] : mine[idx], ay = g.pol ? mine[idx] : oth[idx];
        const T pw = ax + ay;
        if (g.pol == 0) g.st(Pbuf, g.pbase + g.time_off(idx), pw);
        const T phi = c8g * (pw + ax + ay) / (T)2;
        m = (double)phi > m ? (double)phi : m;
    }
    if (a.k.adaptive) {
        m = block_max(ctx, m, (double *)ctx.lds);
        if (ctx.tid == 0) a.pmax[ctx.bid] = m;
    }
    ctx.sync();                                  // scratch reads done before the FFT reuses the LDS
}
// the phase array of the latest rotation (see SSF_THETA_F32)
template <typename T, class G> SSF_HD T theta_ld(const G &g, const T *th, long long i) {
    if constexpr (SSF_THETA_F32 && sizeof(T) == 8) return (T)g.ld((const float *)th, i);
    else return g.ld(th, i);
}
template <typename T, class G> SSF_HD void theta_st(const G &g, T *th, long long i, T x) {
    if constexpr (SSF_THETA_F32 && sizeof(T) == 8) g.st((float *)th, i, (float)x);
    else g.st(th, i, x);
}
// mk_advance with the partner in the neighbouring lane (SSF_PAIR_DPP): the same arithmetic, every exchange a DPP move -- no LDS, no
// barrier; rotation and |d rot|^2 of a sample are applied as soon as its owner has them
template <typename T, class Ctx, class G>
SSF_HD void mk_advance_dpp(Ctx &ctx, const G &g, const ColArgs<T> &a, cx<T> *v, const T *Pbuf, T shz, bool first,
                           double &num, double &den, double &pch_sum) {
    constexpr int V = G::kV, H = V / 2;
    T pw8[H], prev8[H], ang8[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const long long t = own_time_off(g, j);
        pw8[j] = g.ld(Pbuf, g.pbase + t);
        prev8[j] = first ? (T)0 : theta_ld(g, a.Theta, g.pbase + t);
    }
    const T c8g = (T)a.k.c8g;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const T lo = norm2(v[j]), hi = norm2(v[j + H]);
        const T nown = g.pol ? hi : lo, noth = g.pol ? lo : hi;
        const T po = (T)ctx.xchg((double)noth);                  // the partner's power of the sample I own
        const T ax = g.pol ? po : nown, ay = g.pol ? nown : po;
        const T pw = pw8[j];
        pch_sum += (double)pw;
        ang8[j] = shz * (c8g * (pw + ax + ay) / (T)2);
        if (first) prev8[j] = shz * (c8g * (pw + pw) / (T)2);
    }
    ctx.mark(6);
#pragma unroll
    for (int idx = 0; idx < V; ++idx) v[idx] = g.ld(a.Ehd, g.rowbase + g.time_off(idx));
    ctx.mark(7);
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const T ang = ang8[j];
        const double s = sin_half_angle((double)ang - (double)prev8[j]);
        theta_st(g, a.Theta, g.pbase + own_time_off(g, j), ang);
        const cx<T> ro = cis_t<T>(ang);
        const T dd = (T)(4.0 * s * s);
        const cx<T> rp = mk<T>((T)ctx.xchg((double)ro.re), (T)ctx.xchg((double)ro.im));
        const T dp = (T)ctx.xchg((double)dd);
        // register j is owned by the x thread, register j + H by the y thread
        const cx<T> r_lo = g.pol ? rp : ro, r_hi = g.pol ? ro : rp;
        const T d_lo = g.pol ? dp : dd, d_hi = g.pol ? dd : dp;
        const double w0 = (double)norm2(v[j]), w1 = (double)norm2(v[j + H]);
        num += w0 * (double)d_lo + w1 * (double)d_hi;
        den += w0 + w1;
        v[j] = v[j] * r_lo;
        v[j + H] = v[j + H] * r_hi;
    }
}
// next iterate (channels.py:436, 414-417): v holds E_fd(it) on entry and E_hd * rot_{it+1} on exit;
// accumulates the sums of lim_{it+1} = |E_hd (rot_{it+1} - rot_it)| / |E_hd| (see the header note).
// Every phase / rotation / |d rot|^2 is evaluated by the sample's owner and swapped through LDS.
template <typename T, class Ctx, class G>
SSF_HD void mk_advance(Ctx &ctx, const G &g, const ColArgs<T> &a, cx<T> *v, const T *Pbuf, T shz, bool first,
                       double &num, double &den, double &pch_sum) {
    constexpr int V = G::kV, H = V / 2;
    T *shN = (T *)ctx.lds;                                   // V*half powers for the owners
    cx<T> *shC = (cx<T> *)(shN + V * (size_t)g.half);        // V*half rotations
    T *shD = (T *)(shC + V * (size_t)g.half);                // V*half |d rot|^2
    T nown[H], noth[H];
    // The step-start powers and the last phases of the owned samples are fetched here, all at once, and arrive
    // while the partners swap their powers.  (Fetched inside the phase loop, each pair of loads waits for the
    // Theta store of the sample before it - the compiler has to assume they alias - and the loop pays eight
    // memory round trips in a row.)
    T pw8[H], prev8[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const long long t = own_time_off(g, j);
        pw8[j] = g.ld(Pbuf, g.pbase + t);
        prev8[j] = first ? (T)0 : theta_ld(g, a.Theta, g.pbase + t);
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const T lo = norm2(v[j]), hi = norm2(v[j + H]);
        nown[j] = g.pol ? hi : lo;
        noth[j] = g.pol ? lo : hi;                           // goes to the partner, who owns that sample
    }
    ctx.sync();                                              // the inverse transform's LDS reads are done
#pragma unroll
    for (int j = 0; j < H; ++j) shN[(size_t)oth_idx(g, j) * g.half + g.t] = noth[j];
    ctx.sync();
    const T c8g = (T)a.k.c8g;
    T ang8[H];                                               // the new phases; prev8 <- the old ones
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const T po = shN[(size_t)own_idx(g, j) * g.half + g.t];
        const T ax = g.pol ? po : nown[j], ay = g.pol ? nown[j] : po;
        const T pw = pw8[j];
        pch_sum += (double)pw;                               // sum |E(step start)|^2 of the owned samples (lim_0 bound)
        ang8[j] = shz * (c8g * (pw + ax + ay) / (T)2);
        if (first) prev8[j] = shz * (c8g * (pw + pw) / (T)2);
    }
    ctx.mark(6);
    // E_hd goes into the dead iterate's registers and is fetched before the phase loop, so that it arrives while
    // the sines and cosines are evaluated.  To make room the loop keeps nothing but the two phases per sample:
    // rotation and |d rot|^2 go to LDS and all sixteen come back from there (own and partner's alike) once the
    // partners have met.
#pragma unroll
    for (int idx = 0; idx < H; ++idx) v[idx] = g.ld(a.Ehd, g.rowbase + g.time_off(idx));
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const long long t = own_time_off(g, j);
        const T ang = ang8[j], prev = prev8[j];
        // |rot_new - rot_old|^2 = 4 sin^2((theta_new - theta_old) / 2)
        const double s = sin_half_angle((double)ang - (double)prev);
        theta_st(g, a.Theta, g.pbase + t, ang);                   // read and written by the owner only
        shC[(size_t)own_idx(g, j) * g.half + g.t] = cis_t<T>(ang);
        shD[(size_t)own_idx(g, j) * g.half + g.t] = (T)(4.0 * s * s);
    }
    ctx.mark(7);
#pragma unroll
    for (int idx = H; idx < V; ++idx) v[idx] = g.ld(a.Ehd, g.rowbase + g.time_off(idx));
    ctx.sync();
#pragma unroll
    for (int idx = 0; idx < V; ++idx) {
        const cx<T> e = v[idx];
        const double w = (double)norm2(e);
        num += w * (double)shD[(size_t)idx * g.half + g.t];
        den += w;
        v[idx] = e * shC[(size_t)idx * g.half + g.t];
    }
    ctx.sync();
}

// What a Manakov column launch does, decided from the control block; the lead thread forwards the block with the
// state advanced.  op: 0 = S (span start), 1 = H, 2 = I, 3 = rebuild of iterate 0, 4 / 5 = recover the field at the step
// start (ST_RECOVER_A / _B; 5 goes on as 3), -1 = nothing to do.
struct MkColStage {
    bool do_inv = false, do_fwd = false, final_ = false, more = false, exact0 = true;
    bool sparse = false;      // final stage: store the field at the samples the next step's bound of lim_0 reads only
    int op = -1;
    struct { int state, it, cur, pcur; double z, hz; } c{};
};
// the samples of a thread's V that the bound of lim_0 reads: one in sixteen = one cache line in sixteen
template <int V> SSF_HD bool lim0_bound_sample(int idx, int b) { return idx == 0 && (V == 16 || !(b & 1)); }
// Stage groups: a column kernel instantiated for a subset SG of them carries only that subset's code and does NOTHING (forwards
// the control block unchanged) when the state asks for another stage -- the host enqueues such kernels along the sequence it
// predicts (H, ADV x (iterations - 1), FIN per step; FusedCore::run_span) and a wrong guess only costs idle launches: the next
// kernel of the right group picks the state up.  SG_ALL = the one kernel that does whatever the state asks for.
enum { SG_H = 1, SG_ADV = 2, SG_FIN = 4, SG_RARE = 8, SG_ALL = 15 };
SSF_HD int stage_group(int op, bool final_) { return op == 1 ? SG_H : op == 2 ? (final_ ? SG_FIN : SG_ADV) : SG_RARE; }
template <class Ctx, class Args> SSF_HD void mk_col_stage(Ctx &ctx, const Args &a, MkColStage &st, int sg_mask = SG_ALL) {
    bool &do_inv = st.do_inv, &do_fwd = st.do_fwd, &final_ = st.final_, &more = st.more, &exact0 = st.exact0;
    int &op = st.op;
    auto &c = st.c;
    // only scalars are taken from the control block (a private copy of the struct would live
    // in scratch memory and cost real HBM traffic on every launch)
    c.state = a.cin->state;
    c.it = a.cin->it;
    c.cur = a.cin->cur;
    c.pcur = a.cin->pcur;
    c.z = a.cin->z;
    c.hz = a.cin->hz;
    final_ = a.cin->final_ != 0;
    exact0 = a.k.exact_lim0 || a.cin->exact0 != 0;
    // everything the lead thread's patches of the forwarded block need, read HERE (scalar loads next to the ones above): read
    // after the block's stores they are vector loads the compiler must wait for one by one (cin and cout may alias for all it
    // knows) -- up to five more memory round trips in front of the lead workgroup's own work
    const int c_exact0 = a.cin->exact0, c_redo = a.cin->redo_;
    const long long c_trace_n = a.cin->trace_n, c_steps = a.cin->steps, c_iters = a.cin->iterations;
    const long long c_rebuilt = a.cin->n_rebuilt, c_recovered = a.cin->n_recovered;
    if (c.state == ST_NEED_S) {
        op = 0;
        do_fwd = true;
    } else if (c.state == ST_NEED_H) {
        op = 1;
        do_inv = do_fwd = true;
    } else if (c.state == ST_NEED_I) {
        op = 2;
        do_inv = true;
        more = c.z + c.hz < a.k.Lspan;                                // channels.py:441, 387
        // The final stage only OBSERVES the field (stores it, takes the next step's Pch): its forward transform would rebuild
        // the column spectrum it has just read -- G as the row stage left it is that spectrum / N1 (unnormalised transforms; N1
        // is a power of two, so the factor is exact) -- and so it neither transforms nor rewrites G (one forward column
        // transform and one write of the field less per step); the control block tells the next row stage (gscale).
        do_fwd = final_ ? false : true;
        // ... and the next step reads the stored field at one sample in sixteen (the bound of lim_0), unless lim_0 is always
        // exact, the last step needed the exact one (weak nonlinearity: lim_0 < tol as a rule -- keep storing everything) or
        // a fixed-step run is about to take its short last step (whose lim_0 is rounding-sized).
        st.sparse = final_ && more && !a.k.exact_lim0 && !a.cin->dense &&
                    (a.k.adaptive || pick_hz(a.k, c.z + c.hz, 0.0) == c.hz);
    } else if (c.state == ST_REDO0) {
        op = 3;
        do_fwd = true;
    } else if (c.state == ST_RECOVER_A) {
        op = 4;
        do_fwd = true;
    } else if (c.state == ST_RECOVER_B) {
        op = 5;
        do_inv = do_fwd = true;
    }
    if (op >= 0 && !(stage_group(op, final_) & sg_mask)) {            // not this kernel's stage: nothing happens, the state stays
        op = -1;
        do_inv = do_fwd = false;
    }
    if (ctx.bid == 0 && ctx.tid == 0) {                               // forward the control block
        ctrl_forward(a.cin, a.cout, (int)(sizeof(Ctrl) / 8));
        Ctrl *n = a.cout;
        if (op == 0) {
            n->state = ST_AFTER_S;
            n->gscale = 0;
            n->t_sparse = n->dense = 0;
            n->hz_valid = 0;
        } else if (op == 1) {
            n->state = ST_ROW_ITER;
            n->it = 0;
            n->final_ = a.k.maxIter == 1;
            n->pend0 = n->pendn = 0;
        } else if (op == 4) {
            n->state = ST_RECOVER_ROW;
        } else if (op == 3 || op == 5) {
            if (op == 5) n->t_sparse = 0;
            n->state = ST_ROW_ITER;
            n->it = 0;
            n->final_ = c_exact0 ? 0 : 1;                        // (exact0: rebuilt to measure lim_0, not as final)
            n->redo_ = c_exact0 ? 0 : 1;
            n->pend0 = n->pendn = 0;
            n->n_rebuilt = c_rebuilt + 1;
            if (op == 5) n->n_recovered = c_recovered + 1;
        } else if (op == 2 && !final_) {
            n->state = ST_ROW_ITER;
            n->it = c.it + 1;
            n->pendn = 1;
            if (c.it == 0) {
                n->pend0 = 1;
                n->pend0_idx = c_trace_n;
                n->bound0 = exact0 ? 0 : 1;
                n->exact0 = 0;
            }
        } else if (op == 2) {                                         // the step ends here
            const long long tn = c_trace_n;
            if (tn < a.k.trace_cap) {
                if (a.k.tr_hz) a.k.tr_hz[tn] = c.hz;
                if (a.k.tr_it) a.k.tr_it[tn] = c.it + 1;
            }
            n->trace_n = tn + 1;
            n->steps = c_steps + 1;
            n->iterations = c_iters + c.it + 1;
            n->last_nit = c.it + 1;
            n->z = c.z + c.hz;
            n->cur = c.cur ^ 1;
            n->it = 0;
            n->final_ = 0;
            if (c.it == 0) {
                n->pend0 = 1;
                n->pend0_idx = tn;
                n->cap0 = c_redo ? 0 : 1;
                n->bound0 = exact0 ? 0 : 1;      // (only recorded in a trace, and a trace makes it exact)
                n->exact0 = 0;
            }
            n->redo_ = 0;
            n->t_sparse = st.sparse ? 1 : 0;
            n->dense = 0;
            if (more) {
                n->gscale = 1;
                n->state = ST_AFTER_S;
                n->pcur = c.pcur ^ 1;
                if (a.k.adaptive) n->hz_valid = 0;
                else {
                    const double hz = pick_hz(a.k, c.z + c.hz, 0.0);
                    if (hz != c.hz) n->hz_valid = 0;
                    n->hz = hz;
                }
            } else n->state = ST_SPAN_DONE;
        }
    }
}

// (the DPP variant only exists where the geometry can have it: a run-time test of a compile-time false costs nothing)
#define G_DPP(g) (std::remove_reference<decltype(g)>::type::kDppT && (g).dpp)
// MODE is one of CM_*; the Manakov mode picks its stage from the Ctrl state.
// CI > 0: the workgroup's CI columns (per polarisation row) are interleaved in LDS (lds_put); the launch must have exactly CI of them.
// SG: the stage groups this instantiation carries (Manakov mode; see stage_group).
template <typename T, int LG, int MODE, bool RAGGED, int V = 16, int CI = 0, int SG = SG_ALL, class Ctx>
SSF_HD void col_body(Ctx &ctx, const ColArgs<T> &a) {
    constexpr int H = V / 2;
    constexpr int kCI = CI > 0 ? CI : 1;
    constexpr bool kMk = MODE == CM_MK;
    constexpr bool kgH = (SG & SG_H) != 0, kgADV = (SG & SG_ADV) != 0, kgFIN = (SG & SG_FIN) != 0, kgRARE = (SG & SG_RARE) != 0;
    constexpr bool kFwd = !kMk || kgH || kgADV || kgRARE;             // (the final stage never transforms forward)
    // ---- what does this launch do? ------------------------------------------------------
    bool do_inv = false, do_fwd = false;
    int op = -1;     // Manakov: 0 = S (span start), 1 = H, 2 = I, 3 = rebuild iterate 0
    bool final_ = false, more = false, exact0 = true, sparse = false;
    struct { int state, it, cur, pcur; double z, hz; } c{};
    double *red = (double *)ctx.lds;
    ctx.mark(0);
    ColGeom<T, LG, Ctx, RAGGED, V> g(ctx, a);
    cx<T> v[V];
#if SSF_SPEC_G
    // The spectrum is fetched BEFORE the control block is looked at (its addresses do not depend on the state): the block was
    // written by the previous launch on another XCD, so its first read at the start of a launch is a miss that 512 workgroups
    // wait for before they know their stage -- one memory latency in front of every column launch's loads.  A launch that
    // turns out to have nothing to do (or a rare stage that starts from the time-domain field) drops the values.
    if (kMk) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
            v[q] = g.ld(a.G, g.rowbase + g.freq_off(q));
            if ((q & 3) == 3) ctx.issue_fence();
        }
    }
#endif
    if (kMk) {
        MkColStage st;
        mk_col_stage(ctx, a, st, SG);
        do_inv = st.do_inv;
        do_fwd = kFwd && st.do_fwd;
        final_ = kgFIN && (!kgADV || st.final_);
        more = st.more;
        exact0 = st.exact0;
        sparse = st.sparse;
        op = st.op;
        c.state = st.c.state;
        c.it = st.c.it;
        c.cur = st.c.cur;
        c.pcur = st.c.pcur;
        c.z = st.c.z;
        c.hz = st.c.hz;
        if (op < 0) return;
    } else {
        do_inv = MODE == CM_NLSE_STEP || MODE == CM_NLSE_LAST || MODE == CM_PLAIN_INV;
        do_fwd = MODE == CM_NLSE_STEP || MODE == CM_NLSE_FIRST || MODE == CM_PLAIN_FWD;
    }

    const PassPlan &p = g.p;
    cx<T> *lds = CI > 0 ? (cx<T> *)ctx.lds + (size_t)g.pol * CI * lds_slots_per_fft(p.L) + g.c
                        : (cx<T> *)ctx.lds + (size_t)(g.pol * g.C + g.c) * lds_col_stride(p.L, g.C, (int)sizeof(cx<T>));

    // buffers by role (Manakov)
    cx<T> *Tcur = a.T0, *Tnew = a.T1;
    T *Pcur = a.P, *Palt = a.P;
    if (kMk) {
        Tcur = c.cur ? a.T1 : a.T0;                  // field at the step start
        Tnew = c.cur ? a.T0 : a.T1;                  // receives the field at the step end
        const long long psz = g.N * a.ngroups;
        Pcur = a.P + (c.pcur ? psz : 0);
        Palt = a.P + (c.pcur ? 0 : psz);
    }

    // ---- inverse column transform: G -> time samples in registers -------------------------
    TwSrc<T> tws;
    GTw gtw;
    cx<T> e_pre{};
    if (do_inv) {
        if (!(kMk && SSF_SPEC_G)) {
#pragma unroll
            for (int q = 0; q < V; ++q) {
                v[q] = g.ld(a.G, g.rowbase + g.freq_off(q));
#if SSF_LOAD_ORDER
                if ((q & 3) == 3) ctx.issue_fence();         // (appendix #43: the inter-pass twiddles are applied in this order, group by group)
#endif
            }
        }
        // the one sample per thread that the bound of lim_0 compares with the field at the step start: fetched here, behind the
        // spectrum, instead of where it is used -- there the load was issued and waited for on the spot (s_waitcnt vmcnt(0) right
        // behind it: a whole memory round trip exposed in the middle of the first iterate's launch of every step)
        if (kMk && (kgADV || kgFIN) && op == 2 && c.it == 0 && !exact0 && lim0_bound_sample<V>(0, g.b)) e_pre = g.ld(Tcur, g.rowbase + g.time_off(0));
        ctx.mark(1);
        if (a.prio) ctx.template setprio<3>();
        global_twiddle<+1, RAGGED>(g, a.log2N1 + a.log2N2, v, gtw);
        fft_dif<+1, V, false, kCI>(ctx, p, g.b, v, lds, tws);
        ctx.mark(2);
        if (a.prio) ctx.template setprio<2>();
    } else if (kMk && !kgRARE) {                     // (every other stage starts from the spectrum)
    } else if (!(kMk && op == 3)) {
        const cx<T> *src = kMk && op == 4 ? a.Ehd : Tcur;
#pragma unroll
        for (int idx = 0; idx < V; ++idx) v[idx] = g.ld(src, g.rowbase + g.time_off(idx));
    }

    // ---- time-domain work ---------------------------------------------------------------
    if (MODE == CM_NLSE_STEP) {                                          // channels.py:225
#pragma unroll
        for (int idx = 0; idx < V; ++idx) v[idx] = v[idx] * cis_t<T>(a.g_hz * norm2(v[idx]));
    } else if (MODE == CM_NLSE_LAST || MODE == CM_PLAIN_INV) {
#pragma unroll
        for (int idx = 0; idx < V; ++idx) g.st(a.T0, g.rowbase + g.time_off(idx), v[idx]);
    } else if (kMk) {
        const T shz = (T)(a.k.sgn * c.hz), c8g = (T)a.k.c8g;
        cx<T> *shC = (cx<T> *)(ctx.lds + 2 * V * (size_t)g.half * sizeof(T));
        if (kgRARE && op == 0) {                     // span start: Pch into the current buffer
            mk_step_start(ctx, g, a, v, Pcur, false);
        } else if (kgRARE && op == 4) {              // (E_hd goes through the forward transform as it is)
        } else if ((kgH && op == 1) || (kgRARE && (op == 3 || op == 5))) {  // H (channels.py:409-417) | rebuild of iterate 0 (5: after the recovered field is out)
            T ang[H];
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const T pw = g.ld(Pcur, g.pbase + own_time_off(g, j));
                ang[j] = shz * (c8g * (pw + pw) / (T)2);
            }
#pragma unroll
            for (int idx = 0; idx < V; ++idx) {
                const long long t = g.time_off(idx);
                if (!kgRARE || op == 1) g.st(a.Ehd, g.rowbase + t, v[idx]);
                else {
                    if (op == 5) g.st(Tcur, g.rowbase + t, v[idx]);
                    v[idx] = g.ld(a.Ehd, g.rowbase + t);
                }
            }
            cx<T> rot[V];
            if (!G_DPP(g)) ctx.sync();               // inverse transform's LDS reads are done
            pair_cis(ctx, g, ang, rot, shC);
#pragma unroll
            for (int idx = 0; idx < V; ++idx) v[idx] = v[idx] * rot[idx];
            ctx.sync();
        } else if (kgADV || kgFIN) {                 // I: iterate `it` is in registers
            double n0 = 0, d0 = 0, n1 = 0, d1 = 0, psum = 0;
            if (c.it == 0) {                         // lim_0 against the field at the step start
                // (exact: all samples, fetched together -- loaded where they are used, one by one, every load was waited for on
                //  the spot: sixteen memory round trips in a row in the traced / maxIter = 1 / weak-nonlinearity runs)
                cx<T> e[V];
                if (exact0) {
#pragma unroll
                    for (int idx = 0; idx < V; ++idx) e[idx] = g.ld(Tcur, g.rowbase + g.time_off(idx));
                } else e[0] = e_pre;
#pragma unroll
                for (int idx = 0; idx < V; ++idx) {
                    if (exact0 || lim0_bound_sample<V>(idx, g.b)) {
                        const double dr = (double)v[idx].re - (double)e[idx].re, di = (double)v[idx].im - (double)e[idx].im;
                        n0 += dr * dr + di * di;
                        d0 += (double)e[idx].re * e[idx].re + (double)e[idx].im * e[idx].im;
                    }
                }
            }
            if (final_) {                            // the field after this step (channels.py:438-439)
#pragma unroll
                for (int idx = 0; idx < V; ++idx)
                    if (!sparse || lim0_bound_sample<V>(idx, g.b)) g.st(Tnew, g.rowbase + g.time_off(idx), v[idx]);
            } else if (kgADV) {
                if (G_DPP(g)) mk_advance_dpp(ctx, g, a, v, Pcur, shz, c.it == 0, n1, d1, psum);
                else mk_advance(ctx, g, a, v, Pcur, shz, c.it == 0, n1, d1, psum);
                if (!exact0) d0 = psum;              // exact denominator of the bound: sum Pch over the tile
            }
            if (c.it == 0) {
                block_sum2(ctx, n0, d0, red);
                if (ctx.tid == 0) {
                    a.pnum0[ctx.bid] = n0;
                    a.pden0[ctx.bid] = d0;
                }
            }
            if (!final_) {
                block_sum2(ctx, n1, d1, red);
                if (ctx.tid == 0) {
                    a.pnum[ctx.bid] = n1;
                    a.pden[ctx.bid] = d1;
                }
                ctx.sync();
            } else if (more) {
                mk_step_start(ctx, g, a, v, Palt, true);     // next step: Pch + forward transform
            }
        }
    }

    // ---- forward column transform: registers -> G -------------------------------------------
    ctx.mark(3);
    if (a.prio) ctx.template setprio<1>();
    if (kFwd && do_fwd) {
        if (!kMk && do_inv) ctx.sync();              // (Manakov paths synchronised above)
        fft_dit<-1, V, false, kCI>(ctx, p, g.b, v, lds, tws);
        global_twiddle<-1, RAGGED>(g, a.log2N1 + a.log2N2, v, gtw);
        ctx.mark(4);
        if (a.prio) ctx.template setprio<0>();
#pragma unroll
        for (int q = 0; q < V; ++q) g.st(a.G, g.rowbase + g.freq_off(q), v[q]);
        ctx.mark(5);
        ctx.flush(do_inv ? 0 : 1);
    }
}

// ------------------------------------------------------------- column stage for ANY 2^a 3^b 5^c column length
// The radix-2^n column kernels above want a power-of-two column length, which leaves lengths whose power-of-two part is too
// small for the rest to fit a row (2 000 000 = 2^7 x 5^6: the top size of the reference's own benchmark,
// examples/benchmarck_GPU_processing.ipynb: rows of 15 625) to the host-driven Bluestein path.  This kernel takes any
// N = N1 x N2 with both factors 2^a 3^b 5^c: a tile of C adjacent columns x N1 of BOTH polarisations lives in LDS (columns
// interleaved: slot ((pol N1 + pos) C + c), so a global row segment of C samples is C consecutive slots), the transforms are the
// in-place mixed-radix passes of mixed_fft.h over elements C slots apart, and the time-domain work is a loop over the tile's
// samples -- x and y of a sample are both in LDS, so nothing is exchanged between threads.  Same stages, same control block, same
// partial sums as col_body (mk_col_stage); the sample set of the lim_0 bound / sparse field store is every sixteenth time row.
// A general-purpose kernel: far from the roofline of the specialised ones, but device-resident and one launch per stage.
constexpr int kColMixScratch = 8192;            // bytes of LDS in front of the tile: block reductions (lower half), bin table (upper half)
constexpr int kMix2MaxCol = 1024;               // longest column of this stage
// (plan1: the column pass plan in the kernel arguments themselves -- see row_mixed_body)
template <typename T, int MODE, class Ctx> SSF_HD void col_mixed_body(Ctx &ctx, const ColArgs<T> &a, const MixPlan &plan1) {
    static_assert(sizeof(T) == sizeof(scalar_t<T>), "one row per polarisation (no packed pairs)");
    constexpr bool kMk = MODE == CM_MK;
    const int N1 = a.N1mix, C = a.mix_cols, N2 = a.N2, npol = a.npol;
    const long long N = a.N;
    int lgC = 0;
    while ((1 << lgC) < C) ++lgC;
    const int tpp = (N2 + C - 1) / C;                         // tiles per field group
    const int grp = ctx.bid / tpp;
    int tile = ctx.bid - grp * tpp;
    if (tpp % 8 == 0) tile = (tile & 7) * (tpp >> 3) + (tile >> 3);        // contiguous runs of tiles per XCD (see ColGeom)
    else if (tpp > 8) {                                                    // ... for any number of tiles (4 columns are half a cache line)
        const int q = tpp >> 3, r = tpp & 7, x = tile & 7;
        tile = x * q + (x < r ? x : r) + (tile >> 3);
    }
    const int n2base = tile * C;
    const long long rowbase0 = (long long)grp * npol * N, pbase = (long long)grp * N;
    double *red = (double *)ctx.lds;
    cx<T> *X = (cx<T> *)(ctx.lds + kColMixScratch);
    // time row n1 of tile position pos (digit reversal): a table in the scratch area's upper half instead of mix_bin()'s run-time
    // divisions per element and loop
    unsigned short *binlut = (unsigned short *)(ctx.lds + kColMixScratch / 2);
    static_assert(kMix2MaxCol * sizeof(unsigned short) <= kColMixScratch / 2, "bin table fits the scratch area");
    for (int pos = ctx.tid; pos < N1; pos += ctx.nthreads) binlut[pos] = (unsigned short)mix_bin(plan1, pos);
    ctx.sync();

    bool do_inv, do_fwd;
    MkColStage st;
    if (kMk) {
        mk_col_stage(ctx, a, st, SG_ALL);
        if (st.op < 0) return;
        do_inv = st.do_inv;
        do_fwd = st.do_fwd;
    } else {
        do_inv = MODE == CM_NLSE_STEP || MODE == CM_NLSE_LAST || MODE == CM_PLAIN_INV;
        do_fwd = MODE == CM_NLSE_STEP || MODE == CM_NLSE_FIRST || MODE == CM_PLAIN_FWD;
    }
    const int op = st.op;
    cx<T> *Tcur = a.T0, *Tnew = a.T1;
    T *Pcur = a.P, *Palt = a.P;
    if (kMk) {
        Tcur = st.c.cur ? a.T1 : a.T0;
        Tnew = st.c.cur ? a.T0 : a.T1;
        const long long psz = N * a.ngroups;
        Pcur = a.P + (st.c.pcur ? psz : 0);
        Palt = a.P + (st.c.pcur ? 0 : psz);
    }
    // frequency side: thread (c, t) walks k1 = t, t + Tt, ... of column c; the inter-pass twiddle cis(sign 2 pi n2 k1 / N) follows
    // by a chain of products from two bases (N1 / Tt <= a few dozen products: 1e-15)
    const int fc = ctx.tid & (C - 1), ft = ctx.tid >> lgC, Tt = ctx.nthreads >> lgC;
    const int fn2 = n2base + fc;
    const bool fvalid = fn2 < N2;
    cx<double> w0 = mk<double>(1.0, 0.0), ws = w0;
    if (do_inv || do_fwd) {
        w0 = cis2pi<double>((double)(((long long)fn2 * ft) % N) / (double)N);
        ws = cis2pi<double>((double)(((long long)fn2 * Tt) % N) / (double)N);
    }
    // time side: element e of the tile = (position pos in the transform's digit-reversed order, column c); time row n1 = mix_bin(pos)
    const int ne = N1 << lgC;
    // one transform per (polarisation, column): nthreads / (npol C) threads each
    const int ntr = npol << lgC, tr = ctx.tid % ntr, tt = ctx.tid / ntr, tpt = ctx.nthreads / ntr;
    cx<T> *xt = X + ((size_t)(tr >> lgC) * N1 << lgC) + (tr & (C - 1));

    // Global loads go out in batches of U per thread, all of a batch issued before the first is consumed: the trip counts are run-time
    // values, the compiler does not pipeline such loops by itself, and an iteration that waits for its own loads costs a memory
    // latency each (round 6, 2 000 000 = 500 x 4000: G loads 6.9 us and time-domain work 12.7 us of a workgroup's 36 us,
    // profiles/r6_mixed_phases.txt).  Addresses of elements a thread does not have are clamped to valid ones, their values dropped.
    constexpr int U = 4;
    ctx.mark(0);
    if (do_inv) {                                                     // G -> x cis(+...) -> inverse column transform
        cx<double> w = w0;
        const int fn2c = fvalid ? fn2 : N2 - 1;
        for (int k0 = ft; k0 < N1; k0 += U * Tt) {
            cx<T> v[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k1 = k0 + u * Tt < N1 ? k0 + u * Tt : ft;
#pragma unroll
                for (int pol = 0; pol < 2; ++pol)
                    if (pol < npol) v[u][pol] = a.G[rowbase0 + (long long)pol * N + (long long)k1 * N2 + fn2c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k1 = k0 + u * Tt;
                if (k1 < N1) {
#pragma unroll
                    for (int pol = 0; pol < 2; ++pol)
                        if (pol < npol) X[(((size_t)pol * N1 + k1) << lgC) + fc] = fvalid ? mul_by_d(v[u][pol], w) : mk<T>((T)0, (T)0);
                    w = w * ws;
                }
            }
        }
        ctx.sync();
        ctx.mark(1);
        mix_dif_strided<+1>(ctx, plan1, tt, tpt, xt, a.wtab1, C);
        ctx.mark(2);
    }
    // ---- time-domain work on the tile ----------------------------------------------------------------------------
    // element e of the tile: position e >> lgC of the transform's digit-reversed order, column e & (C - 1); el() gives its offset in a
    // time-domain row buffer (0 for an element outside the tile or the field: a valid address whose value is not used)
    struct El {
        int pos, c;
        long long toff;
        bool in, ok;      // inside the tile; ... and inside the field (ragged last tile)
    };
    auto el = [&](int e) {
        El r;
        r.in = e < ne;
        r.pos = r.in ? e >> lgC : 0;
        r.c = e & (C - 1);
        r.ok = r.in && n2base + r.c < N2;
        r.toff = r.ok ? (long long)binlut[r.pos] * N2 + n2base + r.c : 0;
        return r;
    };
    // src -> X (time order), for the stages that start from a time-domain buffer
    auto load_time = [&](const cx<T> *src) {
        for (int e0 = ctx.tid; e0 < ne; e0 += U * ctx.nthreads) {
            El q[U];
            cx<T> v[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                q[u] = el(e0 + u * ctx.nthreads);
#pragma unroll
                for (int pol = 0; pol < 2; ++pol)
                    if (pol < npol) v[u][pol] = src[rowbase0 + (long long)pol * N + q[u].toff];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q[u].in) {
#pragma unroll
                    for (int pol = 0; pol < 2; ++pol)
                        if (pol < npol) X[(((size_t)pol * N1 + q[u].pos) << lgC) + q[u].c] = q[u].ok ? v[u][pol] : mk<T>((T)0, (T)0);
                }
        }
    };
    auto store_time = [&](cx<T> *dst, bool sparse) {
        for (int e = ctx.tid; e < ne; e += ctx.nthreads) {
            const int pos = e >> lgC, c = e & (C - 1), n2 = n2base + c;
            if (n2 >= N2) continue;
            const int n1 = binlut[pos];
            if (sparse && (n1 & 15)) continue;
            for (int pol = 0; pol < npol; ++pol) dst[rowbase0 + (long long)pol * N + (long long)n1 * N2 + n2] = X[(((size_t)pol * N1 + pos) << lgC) + c];
        }
    };
    if (MODE == CM_NLSE_FIRST || MODE == CM_PLAIN_FWD) {
        load_time(a.T0);
    } else if (MODE == CM_NLSE_STEP) {                                // channels.py:225
        for (int e = ctx.tid; e < ne; e += ctx.nthreads) X[e] = X[e] * cis_t<T>(a.g_hz * norm2(X[e]));
    } else if (MODE == CM_NLSE_LAST || MODE == CM_PLAIN_INV) {
        store_time(a.T0, false);
    } else if (kMk) {
        const T shz = (T)(a.k.sgn * st.c.hz), c8g = (T)a.k.c8g;
        const size_t yoff = (size_t)N1 << lgC;                        // the y polarisation's half of the tile
        // step start (channels.py:388-395): Pch -> Pbuf, block max of phi -> pmax
        auto step_start = [&](T *Pbuf) {
            double m = -INFINITY;
            for (int e = ctx.tid; e < ne; e += ctx.nthreads) {
                const int pos = e >> lgC, c = e & (C - 1), n2 = n2base + c;
                if (n2 >= N2) continue;
                const T ax = norm2(X[e]), ay = norm2(X[yoff + e]);
                const T pw = ax + ay;
                Pbuf[pbase + (long long)binlut[pos] * N2 + n2] = pw;
                const T phi = c8g * (pw + ax + ay) / (T)2;
                m = (double)phi > m ? (double)phi : m;
            }
            if (a.k.adaptive) {
                m = block_max(ctx, m, red);
                if (ctx.tid == 0) a.pmax[ctx.bid] = m;
            }
        };
        // first rotation of a step (channels.py:409-417): X <- X cis(shz phi(Pch, Pch))
        auto rotate0 = [&]() {
            for (int e0 = ctx.tid; e0 < ne; e0 += U * ctx.nthreads) {
                El q[U];
                T pw[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    q[u] = el(e0 + u * ctx.nthreads);
                    pw[u] = Pcur[pbase + q[u].toff];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (q[u].ok) {
                        const int e = e0 + u * ctx.nthreads;
                        const cx<T> rot = cis_t<T>(shz * (c8g * (pw[u] + pw[u]) / (T)2));
                        X[e] = X[e] * rot;
                        X[yoff + e] = X[yoff + e] * rot;
                    }
            }
        };
        if (op == 0) {
            load_time(Tcur);
            ctx.sync();
            step_start(Pcur);
        } else if (op == 4) {
            load_time(a.Ehd);
        } else if (op == 1) {
            store_time(a.Ehd, false);
            rotate0();
        } else if (op == 3 || op == 5) {
            if (op == 5) store_time(Tcur, false);
            ctx.sync();
            load_time(a.Ehd);
            ctx.sync();
            rotate0();
        } else {                                                      // op 2: iterate `it` is in the tile
            double n0 = 0, d0 = 0, n1s = 0, d1 = 0, psum = 0;
            const bool first = st.c.it == 0, final_ = st.final_;
            if (first) {                                              // lim_0 against the field at the step start
                for (int e = ctx.tid; e < ne; e += ctx.nthreads) {
                    const int pos = e >> lgC, c = e & (C - 1), n2 = n2base + c;
                    if (n2 >= N2) continue;
                    const int n1 = binlut[pos];
                    if (!st.exact0 && (n1 & 15)) continue;
                    for (int pol = 0; pol < npol; ++pol) {
                        const cx<T> e0 = Tcur[rowbase0 + (long long)pol * N + (long long)n1 * N2 + n2], v = X[(((size_t)pol * N1 + pos) << lgC) + c];
                        const double dr = (double)v.re - (double)e0.re, di = (double)v.im - (double)e0.im;
                        n0 += dr * dr + di * di;
                        d0 += (double)e0.re * e0.re + (double)e0.im * e0.im;
                    }
                }
            }
            if (final_) {                                             // the field after this step (channels.py:438-439)
                store_time(Tnew, st.sparse);
            } else {
                // next iterate (channels.py:436, 414-417): X <- E_hd rot_{it+1}; sums of lim_{it+1} (see the note at the top of this file)
                for (int e0 = ctx.tid; e0 < ne; e0 += U * ctx.nthreads) {
                    El q[U];
                    T pw[U], prev[U];
                    cx<T> hx[U], hy[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        q[u] = el(e0 + u * ctx.nthreads);
                        pw[u] = Pcur[pbase + q[u].toff];
                        prev[u] = first ? (T)0 : a.Theta[pbase + q[u].toff];
                        hx[u] = a.Ehd[rowbase0 + q[u].toff];
                        hy[u] = a.Ehd[rowbase0 + N + q[u].toff];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (q[u].ok) {
                            const int e = e0 + u * ctx.nthreads;
                            const T ax = norm2(X[e]), ay = norm2(X[yoff + e]);
                            const T ang = shz * (c8g * (pw[u] + ax + ay) / (T)2);
                            const T pv = first ? shz * (c8g * (pw[u] + pw[u]) / (T)2) : prev[u];
                            psum += (double)pw[u];
                            a.Theta[pbase + q[u].toff] = ang;
                            const double sh = sin_half_angle((double)ang - (double)pv);
                            const cx<T> rot = cis_t<T>(ang);
                            const double w = (double)norm2(hx[u]) + (double)norm2(hy[u]);
                            n1s += w * (4.0 * sh * sh);
                            d1 += w;
                            X[e] = hx[u] * rot;
                            X[yoff + e] = hy[u] * rot;
                        }
                }
                if (!st.exact0) d0 = psum;                            // exact denominator of the bound: sum Pch over the tile
            }
            if (first) {
                block_sum2(ctx, n0, d0, red);
                if (ctx.tid == 0) {
                    a.pnum0[ctx.bid] = n0;
                    a.pden0[ctx.bid] = d0;
                }
            }
            if (!final_) {
                block_sum2(ctx, n1s, d1, red);
                if (ctx.tid == 0) {
                    a.pnum[ctx.bid] = n1s;
                    a.pden[ctx.bid] = d1;
                }
            } else if (st.more) {
                ctx.sync();
                step_start(Palt);                                     // next step: Pch (the field's spectrum is in G already)
            }
        }
    }
    // ---- forward column transform -> x cis(-...) -> G ---------------------------------------------------------------
    ctx.mark(3);
    if (do_fwd) {
        ctx.sync();
        mix_dit_strided<-1>(ctx, plan1, tt, tpt, xt, a.wtab1, C);
        ctx.mark(4);
        cx<double> w = conj(w0);
        const cx<double> wsc = conj(ws);
        for (int k1 = ft; k1 < N1; k1 += Tt) {
            if (fvalid)
                for (int pol = 0; pol < npol; ++pol)
                    a.G[rowbase0 + (long long)pol * N + (long long)k1 * N2 + fn2] = mul_by_d(X[(((size_t)pol * N1 + k1) << lgC) + fc], w);
            w = w * wsc;
        }
        ctx.mark(5);
        ctx.flush(do_inv ? 0 : 1);
    }
}

// ------------------------------------------------------------- packed pair: complex64 Manakov
// The complex64 Manakov path keeps both polarisations of a sample in one 16-byte element (fused_core.h: pf2), in
// memory as (x.re, y.re, x.im, y.im).  The transforms are the ones above with T = pf2 (one packed instruction per
// butterfly operation for both polarisations, twiddles and operator shared); the time-domain work below needs no
// exchange between threads: |Ex|^2 + |Ey|^2, the phase and the rotation of a sample are local to its thread.
// Same stages, same control block, same sums as the polarisation-split double-precision kernel (col_body).
struct RepackArgs {
    cx<float> *soa;       // (2 K, N): rows x0, y0, x1, y1, ...
    cx<pf2> *pk;          // (K, N) packed pairs
    long long N;
    int npairs, to_pk;    // to_pk: soa -> pk, else pk -> soa
};
template <class Ctx> SSF_HD void repack_body(Ctx &ctx, const RepackArgs &a) {
    const long long total = a.N * a.npairs;
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < total; i += (long long)ctx.nblocks * ctx.nthreads) {
        const long long g = i / a.N, n = i - g * a.N;
        cx<float> *x = a.soa + (2 * g) * a.N + n, *y = a.soa + (2 * g + 1) * a.N + n;
        if (a.to_pk) {
            const cx<float> ex = *x, ey = *y;
            a.pk[i] = mk<pf2>(mk2(ex.re, ey.re), mk2(ex.im, ey.im));
        } else {
            const cx<pf2> e = a.pk[i];
            *x = mk<float>(e.re[0], e.im[0]);
            *y = mk<float>(e.re[1], e.im[1]);
        }
    }
}

// P (step-start powers) and Theta (last phases) of the PACKED complex64 column stage are written and read by that stage only,
// always by the thread that holds the sample (same kernel, same launch geometry in every launch of a plan): they are laid out by
// owner instead of by sample, [16-byte chunk of the thread's values][workgroup][thread][chunk], so that a thread moves 16 bytes
// and a wave 1 KiB per instruction.  (By sample, the four adjacent columns of a workgroup are 16 B: a wave instruction touches 16
// cache lines for 256 bytes, and the three real-valued accesses per sample cost the memory pipeline as many line look-ups as the
// three complex ones: +1.5 % at config 3.  The polarisation-split double-precision stage moves 64-byte segments by sample and
// loses 0.7 % with this layout -- both in profiles/r4_ab_pk_owner_layout.txt -- so it keeps the by-sample one.)
template <typename T, int NV, class Ctx> struct Owned {          // NV values of type T per thread
    static constexpr int K = 16 / (int)sizeof(T), NQ = NV / K;
    static_assert(NV % K == 0, "a thread's values are whole 16-byte chunks");
    typedef T vec __attribute__((vector_size(16)));
    long long me, stride;                                    // this thread's slot / slots per chunk index
    SSF_HD Owned(const Ctx &ctx) : me((long long)ctx.bid * ctx.nthreads + ctx.tid), stride((long long)ctx.nblocks * ctx.nthreads) {}
    SSF_HD void load(const T *buf, T *x) const {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const vec w = ((const vec *)buf)[q * stride + me];
#pragma unroll
            for (int l = 0; l < K; ++l) x[K * q + l] = w[l];
        }
    }
    SSF_HD void store(T *buf, const T *x) const {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            vec w;
#pragma unroll
            for (int l = 0; l < K; ++l) w[l] = x[K * q + l];
            ((vec *)buf)[q * stride + me] = w;
        }
    }
};
template <int V, class Ctx> using PkOwned = Owned<float, V, Ctx>;
// |Ex|^2 and |Ey|^2 of a packed sample
SSF_HD void pair_pow(cx<pf2> e, float &ax, float &ay) {
    const pf2 n = e.re * e.re + e.im * e.im;
    ax = n[0];
    ay = n[1];
}
// step start (channels.py:388-395): Pch = |Ex|^2 + |Ey|^2 -> Pbuf, block max of phi -> pmax.  Ends with a barrier.
template <class Ctx, class G>
SSF_HD void pk_step_start(Ctx &ctx, const G &g, const ColArgs<pf2> &a, const cx<pf2> *v, float *Pbuf) {
    constexpr int V = G::kV;
    const float c8g = (float)a.k.c8g;
    double m = -INFINITY;
    float pws[V];
#pragma unroll
    for (int idx = 0; idx < V; ++idx) {
        float ax, ay;
        pair_pow(v[idx], ax, ay);
        const float pw = ax + ay;
        pws[idx] = pw;
        const float phi = c8g * (pw + ax + ay) / 2.0f;
        m = (double)phi > m ? (double)phi : m;
    }
    PkOwned<V, Ctx>(ctx).store(Pbuf, pws);
    if (a.k.adaptive) {
        m = block_max(ctx, m, (double *)ctx.lds);
        if (ctx.tid == 0) a.pmax[ctx.bid] = m;
    }
    ctx.sync();
}

template <int LG, int V = 16, int CI = 0, int SG = SG_ALL, class Ctx> SSF_HD void col_pk_body(Ctx &ctx, const ColArgs<pf2> &a) {
    using T = pf2;
    constexpr int kCI = CI > 0 ? CI : 1;
    constexpr bool kgH = (SG & SG_H) != 0, kgADV = (SG & SG_ADV) != 0, kgFIN = (SG & SG_FIN) != 0, kgRARE = (SG & SG_RARE) != 0;
    constexpr bool kFwd = kgH || kgADV || kgRARE;                     // (the final stage never transforms forward)
    ctx.mark(0);
    ColGeom<T, LG, Ctx, false, V> g(ctx, a);
    cx<T> v[V];
#if SSF_SPEC_G
#pragma unroll
    for (int q = 0; q < V; ++q) {                            // (the spectrum ahead of the control block: see col_body)
        v[q] = g.ld(a.G, g.rowbase + g.freq_off(q));
        if ((q & 3) == 3) ctx.issue_fence();
    }
#endif
    MkColStage st;
    mk_col_stage(ctx, a, st, SG);
    if (st.op < 0) return;
    const int op = st.op;
    const bool final_ = kgFIN && (!kgADV || st.final_);
    const PassPlan &p = g.p;
    cx<T> *lds = CI > 0 ? (cx<T> *)ctx.lds + g.c : (cx<T> *)ctx.lds + (size_t)g.c * lds_col_stride(p.L, g.C, (int)sizeof(cx<T>));
    double *red = (double *)ctx.lds;
    cx<T> *Tcur = st.c.cur ? a.T1 : a.T0;                    // field at the step start
    cx<T> *Tnew = st.c.cur ? a.T0 : a.T1;                    // receives the field at the step end
    const long long psz = g.N * a.ngroups;
    float *Pcur = a.P + (st.c.pcur ? psz : 0), *Palt = a.P + (st.c.pcur ? 0 : psz);

    TwSrc<T> tws;
    GTw gtw;
    if (st.do_inv) {
#if !SSF_SPEC_G
#pragma unroll
        for (int q = 0; q < V; ++q) {
            v[q] = g.ld(a.G, g.rowbase + g.freq_off(q));
#if SSF_LOAD_ORDER
            if ((q & 3) == 3) ctx.issue_fence();             // (appendix #43: the inter-pass twiddles are applied in this order, group by group)
#endif
        }
#endif
        // (the sample of the lim_0 bound is NOT fetched ahead here as col_body does: in this kernel it costs -0.3 ... -2.1 %, 4 of 4,
        //  where the double-precision stage gains +0.1 ... +1.3 %, 4 of 4: profiles/r5_ab_lim0_prefetch.txt)
        ctx.mark(1);
        global_twiddle<+1, false>(g, a.log2N1 + a.log2N2, v, gtw);
        fft_dif<+1, V, false, kCI>(ctx, p, g.b, v, lds, tws);
        ctx.mark(2);
    } else if (!kgRARE) {                                    // (every other stage starts from the spectrum)
    } else if (op != 3) {
        const cx<T> *src = op == 4 ? a.Ehd : Tcur;
#pragma unroll
        for (int idx = 0; idx < V; ++idx) v[idx] = g.ld(src, g.rowbase + g.time_off(idx));
    }

    const float shz = (float)(a.k.sgn * st.c.hz), c8g = (float)a.k.c8g;
    const PkOwned<V, Ctx> own(ctx);
    if (kgRARE && op == 0) {                                 // span start: Pch into the current buffer
        pk_step_start(ctx, g, a, v, Pcur);
    } else if (kgRARE && op == 4) {                          // (E_hd goes through the forward transform as it is)
    } else if ((kgH && op == 1) || (kgRARE && (op == 3 || op == 5))) {   // H (channels.py:409-417) | rebuild of iterate 0 (5: recovered field out first)
        float pw[V];
        own.load(Pcur, pw);
#pragma unroll
        for (int idx = 0; idx < V; ++idx) {
            const long long t = g.time_off(idx);
            if (!kgRARE || op == 1) g.st(a.Ehd, g.rowbase + t, v[idx]);
            else {
                if (op == 5) g.st(Tcur, g.rowbase + t, v[idx]);
                v[idx] = g.ld(a.Ehd, g.rowbase + t);
            }
        }
#pragma unroll
        for (int idx = 0; idx < V; ++idx) v[idx] = tmul(v[idx], cis_t<float>(shz * (c8g * (pw[idx] + pw[idx]) / 2.0f)));
        ctx.sync();                                          // the inverse transform's LDS reads are done
    } else if (kgADV || kgFIN) {                             // I: iterate `it` is in registers
        double n0 = 0, d0 = 0, n1 = 0, d1 = 0, psum = 0;
        const bool first = st.c.it == 0;
        if (first) {                                         // lim_0 against the field at the step start
            cx<T> e[V];                                      // (exact: all samples fetched together, see col_body)
            if (st.exact0) {
#pragma unroll
                for (int idx = 0; idx < V; ++idx) e[idx] = g.ld(Tcur, g.rowbase + g.time_off(idx));
            } else if (lim0_bound_sample<V>(0, g.b)) e[0] = g.ld(Tcur, g.rowbase + g.time_off(0));
#pragma unroll
            for (int idx = 0; idx < V; ++idx) {
                if (st.exact0 || lim0_bound_sample<V>(idx, g.b)) {
#pragma unroll
                    for (int l = 0; l < 2; ++l) {
                        const double dr = (double)v[idx].re[l] - (double)e[idx].re[l], di = (double)v[idx].im[l] - (double)e[idx].im[l];
                        n0 += dr * dr + di * di;
                        d0 += (double)e[idx].re[l] * e[idx].re[l] + (double)e[idx].im[l] * e[idx].im[l];
                    }
                }
            }
        }
        if (final_) {                                        // the field after this step (channels.py:438-439)
#pragma unroll
            for (int idx = 0; idx < V; ++idx)
                if (!st.sparse || lim0_bound_sample<V>(idx, g.b)) g.st(Tnew, g.rowbase + g.time_off(idx), v[idx]);
        } else if (kgADV) {
            // next iterate (channels.py:436, 414-417): v holds E_fd(it) on entry and E_hd * rot_{it+1} on exit;
            // sums of lim_{it+1} = |E_hd (rot_{it+1} - rot_it)| / |E_hd| (see the note at the top of this file)
            float pw[V], prev[V], pn[V];
            own.load(Pcur, pw);
            if (!first) own.load(a.Theta, prev);
#pragma unroll
            for (int idx = 0; idx < V; ++idx) {
                float ax, ay;
                pair_pow(v[idx], ax, ay);
                pn[idx] = shz * (c8g * (pw[idx] + ax + ay) / 2.0f);                    // the new phase
                if (first) prev[idx] = shz * (c8g * (pw[idx] + pw[idx]) / 2.0f);
                psum += (double)pw[idx];
            }
            ctx.mark(6);
#pragma unroll
            for (int idx = 0; idx < V; ++idx) v[idx] = g.ld(a.Ehd, g.rowbase + g.time_off(idx));
            own.store(a.Theta, pn);
            ctx.mark(7);
#pragma unroll
            for (int idx = 0; idx < V; ++idx) {
                // |rot_new - rot_old|^2 = 4 sin^2((theta_new - theta_old) / 2); the two phases agree to a few digits, so
                // their single-precision difference is exact (Sterbenz) and small: the float kernel is enough
                const float dth = pn[idx] - prev[idx];
                const float sn = fabsf(dth) <= 1.5f ? ksin_f(0.5f * dth) : (float)sin_half_angle((double)dth);
                float ex, ey;
                pair_pow(v[idx], ex, ey);
                const double w = (double)ex + (double)ey;
                n1 += w * (double)(4.0f * sn * sn);
                d1 += w;
                v[idx] = tmul(v[idx], cis_t<float>(pn[idx]));
            }
            if (!st.exact0) d0 = psum;                       // exact denominator of the bound: sum Pch over the tile
        }
        if (first) {
            block_sum2(ctx, n0, d0, red);
            if (ctx.tid == 0) {
                a.pnum0[ctx.bid] = n0;
                a.pden0[ctx.bid] = d0;
            }
        }
        if (!final_) {
            block_sum2(ctx, n1, d1, red);
            if (ctx.tid == 0) {
                a.pnum[ctx.bid] = n1;
                a.pden[ctx.bid] = d1;
            }
            ctx.sync();
        } else if (st.more) {
            ctx.sync();
            pk_step_start(ctx, g, a, v, Palt);               // next step: Pch + forward transform
        }
    }

    ctx.mark(3);
    if (kFwd && st.do_fwd) {
        fft_dit<-1, V, false, kCI>(ctx, p, g.b, v, lds, tws);
        global_twiddle<-1, false>(g, a.log2N1 + a.log2N2, v, gtw);
        ctx.mark(4);
#pragma unroll
        for (int q = 0; q < V; ++q) g.st(a.G, g.rowbase + g.freq_off(q), v[q]);
        ctx.mark(5);
        ctx.flush(st.do_inv ? 0 : 1);
    }
}

// arguments of the persistent span kernel (engine_fused_impl.h: k_nlse_span): every stage of a scalar-NLSE span in one launch
template <typename T> struct SpanNlseArgs {
    RowArgs<T> row;
    ColArgs<T> col;
    const LinOp *lin_half, *lin_full;
    int nsteps, row_grid, col_grid;
    unsigned *bar;            // [0] arrivals, [1] generation, [2] abort
};

// arguments of the persistent Manakov span kernel (engine_fused_impl.h: k_mk_span): the launch sequence Col, [Row, Col]* of a
// span inside ONE launch, the device-resident control block deciding what every stage does exactly as between launches
template <typename T> struct SpanMkArgs {
    RowArgs<T> row;           // use_ctrl = 1; cin / cout are set per stage by the kernel
    ColArgs<T> col;           // mode CM_MK
    Ctrl *ctrl;               // [2], double-buffered by stage parity
    unsigned seq0;            // parity of the block the first stage reads
    int row_grid, col_grid;   // (virtual) workgroups of the two stages
    int max_stages;
    int nworkers;             // workgroups that take part
    int xcd;                  // >= 0: only workgroups that run on this XCD take part (one coherent L2: no write-back per barrier)
    size_t ctrl_lds;          // byte offset of the control block's LDS copy (behind the stages' own LDS)
    unsigned *bar;            // [0] arrivals, [1] generation, [2] abort, [3] tickets
};

// --------------------------------------------------------------------- elementwise helpers
template <typename T> struct AmpArgs {
    cx<T> *E;
    const cx<T> *noise;   // host-supplied noise (may be null)
    long long total, N;
    T gain;
    double sigma;         // > 0: add device-generated ASE, sigma per quadrature
    unsigned long long seed;
    unsigned span;
    unsigned row0;        // stream row of row 0 (ssf_params::rng_row_offset)
};
template <typename T, class Ctx> SSF_HD void amp_body(Ctx &ctx, const AmpArgs<T> &a) {
    for (long long i = (long long)ctx.bid * ctx.nthreads + ctx.tid; i < a.total; i += (long long)ctx.nblocks * ctx.nthreads) {
        cx<T> e = a.E[i] * a.gain;
        if (a.noise) e = e + a.noise[i];
        if (a.sigma > 0) {
            double re, im;
            gauss_pair((unsigned long long)(i % a.N), a.row0 + (unsigned)(i / a.N), a.span, a.seed, a.sigma, re, im);
            e = e + mk<T>((T)re, (T)im);
        }
        a.E[i] = e;
    }
}

// ------------------------------------------------------------------- overlap-save convolution
// One block of the overlap-and-save FFT convolution of optic/dsp/core.py:1032-1041 per transform:
//   X = fft(xpad[blk*d : blk*d + NFFT]);  y_blk = ifft(X * H);  y[blk*d : (blk+1)*d] = y_blk[discard:]
// followed by out = y[D : D + sigLen] (core.py:1043-1046), all fused: the padded signal and y are
// never materialised, the block is transformed in LDS with the same DIF/DIT pair as the row kernel.
template <typename T> struct OlsArgs {
    const cx<T> *in;      // (inLen, in_ld) row-major, as the reference's sigIn; this launch filters columns [0, nrows)
    cx<T> *out;           // (keep, out_ld)
    const cx<T> *H;       // NFFT values per filter: fft(zero-padded impulse response) / NFFT
    long long sigLen, njobs;   // njobs = numBlocks * nrows
    int nrows, log2nfft, d, discard, D;
    // extensions used by the receiver pipeline (all neutral when zero-initialised through ols_defaults):
    long long inLen;      // samples actually present in `in` (the rest of sigLen is the reference's zero padding)
    long long keep;       // only outputs n < keep are stored
    int in_ld, out_ld;    // leading dimensions (columns per sample) of in / out
    int Hstride;          // 0: one filter for all columns; NFFT: column m uses H + m * NFFT
    int roll;             // np.roll(y, -roll) before the [:keep] cut (optic/dsp/core.py:920-922)
    int in_up;            // > 1: `in` holds every in_up-th sample, the others are zero (upsample, core.py:395-432)
    // filters of any length, one segment of the impulse response per launch (RxCore::fir_long): output n takes the full
    // convolution's sample n + D + Dx (Dx of either sign), the launch's first block is blk0, and the results are ADDED
    long long Dx, blk0;
    int acc;
};
template <typename T> SSF_HD void ols_defaults(OlsArgs<T> &a) {
    a.inLen = a.sigLen;
    a.keep = a.sigLen;
    a.in_ld = a.out_ld = a.nrows;
    a.Hstride = 0;
    a.roll = 0;
    a.in_up = 1;
    a.Dx = a.blk0 = 0;
    a.acc = 0;
}
// pre(src, m): input sample `src` (an index into the unpadded, un-stuffed signal, 0 <= src < inLen) of column m;
// post(n, m, v): output sample n (after roll and cut) of column m.  The receiver pipeline plugs the stages in front of and behind
// a filter in here (rx_kernels.h: PBS rotation / detection in the loads, IQ mixing in the stores): one pass over the signal less each.
//
// LG > 0: the transform size is a compile-time constant (the pass plan, every register index and the twiddle bases fold; with
// the plan read from the arguments -- LG = 0, kept for transforms below 256 points and the single-precision host path -- the
// value array is indexed at run time and lives in scratch memory).  C: columns of ONE block handled side by side by neighbouring
// lanes (lane = butterfly * C + column): the 16-byte samples of the C columns of a row of the (N, ld) signal are neighbours in
// memory, so a wave's loads and stores cover whole rows instead of every C-th 16-byte piece; the C transforms are interleaved
// in LDS (lds_put / lds_get's CI).  H is stored in REGISTER order (ols_permute_filter below): thread b's value idx reads
// H[idx * tpf + b], consecutive lanes consecutive entries -- in natural order the digit-reversed positions of neighbouring
// lanes are a transform-stride apart (sixteen 64-line gathers per thread).
template <typename T, int LG, int C, class Ctx, class Pre, class Post>
SSF_HD void ols_body_x(Ctx &ctx, const OlsArgs<T> &a, const Pre &pre, const Post &post) {
    const PassPlan p = make_plan(LG > 0 ? LG : a.log2nfft);
    const int tpg = p.tpf * C;                               // threads per group = the C columns of one block
    const int gpw = ctx.nthreads / tpg;
    const int g = ctx.tid / tpg, r = ctx.tid - g * tpg, b = r / C, c = r - b * C;
    const int ncg = a.nrows / C;                             // column groups per block (launchers: nrows % C == 0)
    const long long job = (long long)ctx.bid * gpw + g;
    const bool live = job < a.njobs / C;                     // idle threads still take part in the barriers
    const long long jb = !live ? 0 : ncg == 1 ? job : job / ncg;
    const int m = (live ? (int)(job - jb * ncg) * C : 0) + c;
    const long long blk = jb + a.blk0;
    cx<T> *l = (cx<T> *)ctx.lds + (size_t)g * C * lds_slots_per_fft(p.L) + c;
    const cx<T> *H = a.H + (size_t)m * a.Hstride + b;
    const long long i0 = blk * a.d - a.discard;              // index of the block's first sample in the unpadded signal
    cx<T> v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const long long i = i0 + (b + p.tpf * q);
        bool have = live && i >= 0 && i < a.inLen;
        long long src = i;
        if (a.in_up > 1) {                                                // zero-stuffed input, never materialised
            src = i / a.in_up;
            have = have && src * a.in_up == i;
        }
        v[q] = have ? pre(src, m) : mk<T>((T)0, (T)0);
    }
    TwSrc<T> tws;                                            // the inverse transform conjugates the forward one's bases
    fft_dif<-1, 16, false, C>(ctx, p, b, v, l, tws);
#pragma unroll
    for (int idx = 0; idx < 16; ++idx) v[idx] = v[idx] * H[idx * p.tpf];
    fft_dit<+1, 16, false, C>(ctx, p, b, v, l, tws);
    const long long n0 = i0 - a.D - a.Dx;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int pos = b + p.tpf * q;
        long long n = n0 + pos;
        if (live && pos >= a.discard && n >= 0 && n < a.sigLen) {
            n -= a.roll;
            if (n < 0) n += a.sigLen;
            if (n < a.keep) post(n, m, v[q]);
        }
    }
}
template <typename T, int LG, int C, class Ctx> SSF_HD void ols_body(Ctx &ctx, const OlsArgs<T> &a) {
    ols_body_x<T, LG, C>(
        ctx, a, [&](long long src, int m) { return a.in[src * a.in_ld + m]; },
        [&](long long n, int m, cx<T> v) {
            cx<T> *o = a.out + n * a.out_ld + m;
            *o = a.acc ? *o + v : v;
        });
}

// natural order -> the order ols_body_x reads (host side, where a filter is built): out[idx * tpf + b] = H[k(b, idx)], k the
// frequency index that value idx of thread b holds after the forward (DIF) transform
template <typename Z> inline void ols_permute_filter(Z *H, int log2nfft) {
    const PassPlan p = make_plan(log2nfft);
    const int last = p.npass - 1;
    Z *tmp = new Z[(size_t)p.L];
    for (int b = 0; b < p.tpf; ++b)
        for (int idx = 0; idx < 16; ++idx) tmp[(size_t)idx * p.tpf + b] = H[rev_pos(p, reg_pos(p, last, b, idx))];
    for (int i = 0; i < p.L; ++i) H[i] = tmp[i];
    delete[] tmp;
}

// Launch geometry of the complex128 overlap-save kernels, shared by the HIP backend and the test emulator.  Transforms of
// 256 ... 8192 points get their own instantiation; an even number of columns is taken two at a time while two transforms fit the
// LDS (4096 points: 2 x 68 KiB).
constexpr int kOlsMinLg = 8, kOlsMaxLg = 13, kOlsMaxPairLg = 12;
struct OlsLaunch {
    int lg, C, threads;          // lg = 0: the run-time plan
    long long grid;
    size_t lds_bytes;
};
inline OlsLaunch ols_launch(int log2nfft, int nrows, long long njobs) {
    OlsLaunch o;
    const int nfft = 1 << log2nfft, tpf = nfft / 16;
    o.lg = log2nfft >= kOlsMinLg && log2nfft <= kOlsMaxLg ? log2nfft : 0;
    o.C = o.lg && o.lg <= kOlsMaxPairLg && nrows % 2 == 0 ? 2 : 1;
    o.threads = o.C * tpf > 256 ? o.C * tpf : 256;
    const int gpw = o.threads / (o.C * tpf);
    o.grid = (njobs / o.C + gpw - 1) / gpw;
    o.lds_bytes = (size_t)gpw * o.C * lds_slots_per_fft(nfft) * sizeof(cx<double>);
    return o;
}
constexpr int ols_threads(int LG, int C) { return LG == 0 ? 256 : (C << (LG - 4)) > 256 ? (C << (LG - 4)) : 256; }
// f(integral_constant<LG>, integral_constant<C>) for the instantiation a launch needs
template <class F> inline void ols_dispatch(const OlsLaunch &o, F &&f) {
    using std::integral_constant;
#define SSF_OLS_CASE(L)                                                       \
    case L:                                                                   \
        if (o.C == 2) f(integral_constant<int, L>{}, integral_constant<int, 2>{}); \
        else f(integral_constant<int, L>{}, integral_constant<int, 1>{});     \
        break;
    switch (o.lg) {
        SSF_OLS_CASE(8)
        SSF_OLS_CASE(9)
        SSF_OLS_CASE(10)
        SSF_OLS_CASE(11)
        SSF_OLS_CASE(12)
    case 13: f(integral_constant<int, 13>{}, integral_constant<int, 1>{}); break;
    default: f(integral_constant<int, 0>{}, integral_constant<int, 1>{}); break;
    }
#undef SSF_OLS_CASE
}

}  // namespace fused
}  // namespace ssf
