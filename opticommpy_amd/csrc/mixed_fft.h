// mixed_fft.h -- in-LDS mixed-radix FFT of one row (lengths with factors 2, 3 and 5), used by the row
// kernel for the lengths notebooks actually produce: N = SpS x Nsymbols = 2^a * 3^b * 5^c (e.g. 240 000 =
// 2^7 * 1875).  The power of two goes into the column length (existing radix-2^n column kernel), the rest is
// the row length L, transformed here.
//
// Same pairing as the radix-2^n kernels: the forward transform is decimation in frequency, in place,
// natural order in -> digit-reversed order out; the inverse is decimation in time and consumes exactly that
// order.  Nothing is ever reordered; the linear operator is applied to the digit-reversed spectrum through
// mix_bin().  A pass of radix r over blocks of length M: butterfly (blk, j) gathers x[blk*M + j + (M/r) q],
// q = 0..r-1, from LDS, transforms the r values in registers and puts them back in the same places --
// butterflies of a pass touch disjoint positions, so only passes are separated by barriers.
#pragma once
#include "fused_core.h"

namespace ssf {
namespace fused {

constexpr int kMixMaxPass = 6;
struct MixPlan {
    int L, npass;
    int r[kMixMaxPass];       // radix of pass i
    int M[kMixMaxPass];       // block length of pass i: L / (r[0] * ... * r[i-1])
    int S[kMixMaxPass];       // butterfly stride of pass i: M[i] / r[i]   (kept here: the kernels never divide by a runtime value)
    int W[kMixMaxPass];       // twiddle-table step of pass i: L / M[i]
};

// Radices the butterflies below implement, and what one butterfly costs one thread (relative units: butterfly
// arithmetic + twiddle chain + the LDS round trip of its r values).  A pass over a row of L values with T threads
// takes ceil((L / r) / T) butterflies in a row per thread, so large radices waste threads on short rows: the plan is
// the ordered factorisation with the smallest sum over passes (+ one barrier each).  T = 0: largest radix first.
constexpr int kMixRadices[] = {25, 20, 16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2};
constexpr double kMixRadixCost[] = {440, 335, 256, 237, 180, 145, 125, 108, 80, 65, 48, 30, 20};   // (fitted to plan sweeps at L = 1875, 375: tools/exp/mix_plan_sweep.py)
constexpr double kMixBarrierCost = 100;
// The plan's last pass of a ROW is the stride-1 pass, which runs as one pass for both directions with the row operator between its two
// butterflies (mix_pass_mid): cost of one such butterfly, measured rather than modelled (round 6, after the operator's two starts
// moved out of the unrolled loop: plan sweeps at L = 4000, 3750, 3125 with 256 threads, profiles/r6_mix_plan_sweeps.txt -- radix 10
// beats 16 by 9 % of a row launch although it needs two rounds of butterflies, 8 and 15 sit in between).
constexpr double kMixMidCost[] = {0, 0, 270, 270, 270, 270, 290, 0, 480, 480, 470, 0, 700, 0, 0, 900, 1100};   // indexed by the radix
constexpr int kMixMaxOpRadix = 16;   // the plan's last pass also applies the row operator (mix_apply_op): no room for that at 20 / 25

inline bool mix_plan_from_radices(int L, const int *r, int n, MixPlan *p) {
    p->L = L;
    p->npass = 0;
    int rest = L;
    for (int i = 0; i < n; ++i) {
        if (r[i] < 2 || rest % r[i] || p->npass == kMixMaxPass) return false;
        p->r[p->npass] = r[i];
        p->M[p->npass] = rest;
        p->S[p->npass] = rest / r[i];
        p->W[p->npass] = L / rest;
        rest /= r[i];
        ++p->npass;
    }
    return rest == 1 && p->npass > 0;
}

// with_op: the plan is a ROW's (its last pass is mix_pass_mid: radix <= kMixMaxOpRadix, costed by kMixMidCost); a column stage's
// plan (col_mixed_body: no operator inside the transform) may end on any radix
inline bool mix_make_plan(int L, MixPlan *p, int T = 0, bool with_op = true, double *cost_out = nullptr) {
    // enumerate ordered factorisations; the cost of a pass depends on its radix and on L only
    int cur[kMixMaxPass], best_r[kMixMaxPass], best_n = 0;
    double best = 1e300;
    struct Rec {
        static void go(int L, int rest, int T, bool with_op, int depth, double cost, int *cur, double *best, int *best_r, int *best_n) {
            if (rest == 1) {
                if (depth > 0 && cost < *best) {
                    *best = cost;
                    *best_n = depth;
                    for (int i = 0; i < depth; ++i) best_r[i] = cur[i];
                }
                return;
            }
            if (depth == kMixMaxPass || cost >= *best) return;
            for (int c = 0; c < (int)(sizeof(kMixRadices) / sizeof(int)); ++c) {
                const int r = kMixRadices[c];
                if (rest % r) continue;
                const bool mid = with_op && rest == r;                 // the stride-1 pass of a row
                if (mid && r > kMixMaxOpRadix) continue;
                double pass;
                if (T > 0) pass = (double)(((L / r) + T - 1) / T) * (mid ? kMixMidCost[r] : kMixRadixCost[c]) + kMixBarrierCost;
                else pass = 1.0;                                       // fewest passes; ties: largest radix first (loop order)
                if (T > 0) pass += 1e-3 * r * (kMixMaxPass - depth);   // ties: the larger radix later (500 = 20 x 25: 63.0 us against 65.9 as 25 x 20)
                cur[depth] = r;
                go(L, rest / r, T, with_op, depth + 1, cost + pass, cur, best, best_r, best_n);
            }
        }
    };
    Rec::go(L, L, T, with_op, 0, 0.0, cur, &best, best_r, &best_n);
    if (!best_n) return false;
    if (cost_out) *cost_out = best;
    return mix_plan_from_radices(L, best_r, best_n, p);
}

// frequency bin held at LDS position pos after the forward transform (digit reversal)
SSF_HD int mix_bin(const MixPlan &p, int pos) {
    int k = 0, w = 1;
    for (int i = 0; i < p.npass; ++i) {
        const int s = p.S[i];
        const int q = pos / s;
        pos -= q * s;
        k += q * w;
        w *= p.r[i];
    }
    return k;
}

// ... and the LDS position that holds bin k
SSF_HD int mix_pos(const MixPlan &p, int k) {
    int pos = 0;
    for (int i = 0; i < p.npass; ++i) {
        const int q = k % p.r[i];
        k /= p.r[i];
        pos += q * p.S[i];
    }
    return pos;
}

// ---- small DFTs, natural order in and out, X[k] = sum_q x[q] cis(SIGN 2 pi q k / R) ------------------
// (the irrational constants go through mul_cd: hi + lo in single precision, see mul_by_d)
template <int SIGN, typename T> SSF_HD void dft3(cx<T> &a, cx<T> &b, cx<T> &c) {
    constexpr double s = SIGN * 0.86602540378443864676;
    const T h = (T)-0.5;
    const cx<T> t = b + c, d = b - c;
    const cx<T> m = mk<T>(a.re + h * t.re, a.im + h * t.im);
    const cx<T> r = mk<T>(-mul_cd(d.im, s), mul_cd(d.re, s));     // j s d
    a = a + t;
    b = m + r;
    c = m - r;
}
template <int SIGN, typename T> SSF_HD void dft5(cx<T> *v) {
    constexpr double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
    constexpr double s1 = SIGN * 0.95105651629515357212, s2 = SIGN * 0.58778525229247312917;
    const cx<T> t1 = v[1] + v[4], t2 = v[2] + v[3], d1 = v[1] - v[4], d2 = v[2] - v[3];
    const cx<T> a = v[0];
    const cx<T> m1 = mk<T>(a.re + mul_cd(t1.re, c1) + mul_cd(t2.re, c2), a.im + mul_cd(t1.im, c1) + mul_cd(t2.im, c2));
    const cx<T> m2 = mk<T>(a.re + mul_cd(t1.re, c2) + mul_cd(t2.re, c1), a.im + mul_cd(t1.im, c2) + mul_cd(t2.im, c1));
    const cx<T> r1 = mk<T>(-(mul_cd(d1.im, s1) + mul_cd(d2.im, s2)), mul_cd(d1.re, s1) + mul_cd(d2.re, s2));       // j (s1 d1 + s2 d2)
    const cx<T> r2 = mk<T>(-(mul_cd(d1.im, s2) - mul_cd(d2.im, s1)), mul_cd(d1.re, s2) - mul_cd(d2.re, s1));       // j (s2 d1 - s1 d2)
    v[0] = a + t1 + t2;
    v[1] = m1 + r1;
    v[4] = m1 - r1;
    v[2] = m2 + r2;
    v[3] = m2 - r2;
}

template <int SIGN, int R, typename T> SSF_HD void dft_prime(cx<T> *v) {
    if constexpr (R == 2) dft2<SIGN>(v[0], v[1]);
    else if constexpr (R == 3) dft3<SIGN>(v[0], v[1], v[2]);
    else if constexpr (R == 4) dft4<SIGN>(v[0], v[1], v[2], v[3]);
    else dft5<SIGN>(v);
}

// cos / sin (2 pi k / R) for the composite butterflies (literal constants: a device sincospi is not folded)
template <int R> struct SmallW;
template <> struct SmallW<9> {
    static constexpr double c[9] = {1, 0.76604444311897803519, 0.173648177666930348805, -0.500000000000000000163, -0.939692620785908384103, -0.939692620785908384049, -0.499999999999999999702, 0.173648177666930348696, 0.766044443118978035298};
    static constexpr double s[9] = {0, 0.642787609686539326327, 0.984807753012208059401, 0.866025403784438646678, 0.342020143325668732966, -0.342020143325668733047, -0.866025403784438646949, -0.984807753012208059401, -0.642787609686539326164};
};
template <> struct SmallW<15> {
    static constexpr double c[15] = {1, 0.913545457642600895493, 0.669130606358858213772, 0.309016994374947424076, -0.104528463267653471498, -0.500000000000000000163, -0.809016994374947424104, -0.97814760073380563793, -0.978147600733805637875, -0.809016994374947423941, -0.499999999999999999702, -0.104528463267653471186, 0.309016994374947424185, 0.669130606358858213772, 0.913545457642600895601};
    static constexpr double s[15] = {0, 0.406736643075800207781, 0.743144825477394235065, 0.951056516295153572111, 0.994521895368273336916, 0.866025403784438646678, 0.587785252292473129135, 0.207911690817759336992, -0.20791169081775933729, -0.587785252292473129406, -0.866025403784438646949, -0.99452189536827333697, -0.951056516295153572111, -0.743144825477394235065, -0.406736643075800207537};
};
template <> struct SmallW<25> {
    static constexpr double c[25] = {1, 0.968583161128631119476, 0.876306680043863587301, 0.728968627421411523174, 0.535826794978996618279, 0.309016994374947424076, 0.062790519529313376117, -0.187381314585724630546, -0.425779291565072648802, -0.637423989748689710354, -0.809016994374947424104, -0.929776485888251403667, -0.992114701314477831018, -0.992114701314477831072, -0.929776485888251403667, -0.809016994374947423941, -0.637423989748689710246, -0.425779291565072648721, -0.187381314585724630125, 0.062790519529313375894, 0.309016994374947424185, 0.535826794978996618171, 0.728968627421411523282, 0.876306680043863587301, 0.968583161128631119476};
    static constexpr double s[25] = {0, 0.24868988716485478823, 0.481753674101715274988, 0.684547105928688673728, 0.844327925502015078508, 0.951056516295153572111, 0.99802672842827156195, 0.982287250728688681085, 0.904827052466019527712, 0.770513242775789230653, 0.587785252292473129135, 0.368124552684677959063, 0.125333233564304245448, -0.12533323356430424534, -0.368124552684677959171, -0.587785252292473129406, -0.770513242775789230707, -0.904827052466019527766, -0.982287250728688681139, -0.99802672842827156195, -0.951056516295153572111, -0.844327925502015078617, -0.68454710592868867362, -0.481753674101715274988, -0.248689887164854788406};
};
template <> struct SmallW<6> {
    static constexpr double c[6] = {1, 0.500000000000000000000, -0.500000000000000000000, -1.00000000000000000000, -0.500000000000000000000, 0.500000000000000000000};
    static constexpr double s[6] = {0, 0.866025403784438646764, 0.866025403784438646764, 0, -0.866025403784438646764, -0.866025403784438646764};
};
template <> struct SmallW<10> {
    static constexpr double c[10] = {1, 0.809016994374947424102, 0.309016994374947424102, -0.309016994374947424102, -0.809016994374947424102, -1.00000000000000000000, -0.809016994374947424102, -0.309016994374947424102, 0.309016994374947424102, 0.809016994374947424102};
    static constexpr double s[10] = {0, 0.587785252292473129169, 0.951056516295153572116, 0.951056516295153572116, 0.587785252292473129169, 0, -0.587785252292473129169, -0.951056516295153572116, -0.951056516295153572116, -0.587785252292473129169};
};
template <> struct SmallW<12> {
    static constexpr double c[12] = {1, 0.866025403784438646764, 0.500000000000000000000, 0, -0.500000000000000000000, -0.866025403784438646764, -1.00000000000000000000, -0.866025403784438646764, -0.500000000000000000000, 0, 0.500000000000000000000, 0.866025403784438646764};
    static constexpr double s[12] = {0, 0.500000000000000000000, 0.866025403784438646764, 1.00000000000000000000, 0.866025403784438646764, 0.500000000000000000000, 0, -0.500000000000000000000, -0.866025403784438646764, -1.00000000000000000000, -0.866025403784438646764, -0.500000000000000000000};
};
template <> struct SmallW<20> {
    static constexpr double c[20] = {1, 0.951056516295153572116, 0.809016994374947424102, 0.587785252292473129169, 0.309016994374947424102, 0, -0.309016994374947424102, -0.587785252292473129169, -0.809016994374947424102, -0.951056516295153572116, -1.00000000000000000000, -0.951056516295153572116, -0.809016994374947424102, -0.587785252292473129169, -0.309016994374947424102, 0, 0.309016994374947424102, 0.587785252292473129169, 0.809016994374947424102, 0.951056516295153572116};
    static constexpr double s[20] = {0, 0.309016994374947424102, 0.587785252292473129169, 0.809016994374947424102, 0.951056516295153572116, 1.00000000000000000000, 0.951056516295153572116, 0.809016994374947424102, 0.587785252292473129169, 0.309016994374947424102, 0, -0.309016994374947424102, -0.587785252292473129169, -0.809016994374947424102, -0.951056516295153572116, -1.00000000000000000000, -0.951056516295153572116, -0.809016994374947424102, -0.587785252292473129169, -0.309016994374947424102};
};

// R = A * B (Cooley-Tukey in registers, in place): B transforms of length A over x[b + B a], twiddle
// cis(2 pi b ka / R), A transforms of length B over the slots b + B ka.  Slot kb + B ka then holds X[ka + A kb]:
// the result is left in that permuted order (dft_slot_bin) and the caller stores each slot where it belongs,
// which keeps the butterfly at R live values instead of 2 R.
template <int SIGN, int A, int B, typename T> SSF_HD void dft_ab(cx<T> *v) {
    constexpr int R = A * B;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        cx<T> t[A];
#pragma unroll
        for (int a = 0; a < A; ++a) t[a] = v[b + B * a];
        dft_prime<SIGN, A>(t);
#pragma unroll
        for (int ka = 0; ka < A; ++ka) {
            if (b * ka == 0) {
                v[b + B * ka] = t[ka];
            } else {
                v[b + B * ka] = mul_by_d(t[ka], mk<double>(SmallW<R>::c[(b * ka) % R], SIGN * SmallW<R>::s[(b * ka) % R]));
            }
        }
    }
#pragma unroll
    for (int ka = 0; ka < A; ++ka) dft_prime<SIGN, B>(v + B * ka);
}
// the composite radices: R = A * B
template <int R> struct MixAB { static constexpr int A = 1, B = R; };
template <> struct MixAB<6> { static constexpr int A = 2, B = 3; };
template <> struct MixAB<9> { static constexpr int A = 3, B = 3; };
template <> struct MixAB<10> { static constexpr int A = 2, B = 5; };
template <> struct MixAB<12> { static constexpr int A = 3, B = 4; };
template <> struct MixAB<15> { static constexpr int A = 3, B = 5; };
template <> struct MixAB<20> { static constexpr int A = 4, B = 5; };
template <> struct MixAB<25> { static constexpr int A = 5, B = 5; };

// frequency index of register slot s after dft_small<R>  (dft_ab leaves X[ka + A kb] in slot kb + B ka)
template <int R> SSF_HD constexpr int dft_slot_bin(int s) {
    return MixAB<R>::A == 1 ? s : s / MixAB<R>::B + MixAB<R>::A * (s % MixAB<R>::B);
}

// ... and the slot that holds bin k
template <int R> SSF_HD constexpr int dft_bin_slot(int k) {
    return MixAB<R>::A == 1 ? k : k / MixAB<R>::A + MixAB<R>::B * (k % MixAB<R>::A);
}

template <int SIGN, int R, typename T> SSF_HD void dft_small(cx<T> *v) {
    if constexpr (R == 2 || R == 3 || R == 4 || R == 5) dft_prime<SIGN, R>(v);
    else if constexpr (R == 8) dft8<SIGN>(v);
    else if constexpr (R == 16) dft16<SIGN>(v);
    else dft_ab<SIGN, MixAB<R>::A, MixAB<R>::B>(v);
}

// The linear operator of the row stage, applied where the inverse transform picks the spectrum up (its first pass,
// the plan's last: stride 1).  The R values of one butterfly there are the bins k2 = kb + q L/R, q = 0 .. R-1, of the
// row, i.e. the bins k = k0 + q N/R of the whole transform with k0 = k1 + N1 kb < N/R: they span the spectrum once, so
// the signed bin kk wraps once inside the butterfly.  H(k) = mag cis(cth kk^2) follows a second-order recurrence along
// q within each of the two runs (two complex products per value, four sin/cos evaluations per butterfly).
struct MixRowOp {
    double cth, mag;        // LinOp
    double D;               // N / R as a double, R = radix of the plan's last pass
    cx<double> c2;          // cis(2 cth D^2)
    long long k1, N1, N;    // row (bin offset), bins between row elements, transform length
};
template <int R, typename T> SSF_HD void mix_apply_op(const MixPlan &p, const MixRowOp &op, int blk, cx<T> *v) {
    const long long k0 = op.k1 + op.N1 * (long long)mix_bin(p, blk * R);
    const long long D = op.N / R, npos = (op.N + 1) / 2;
    // the run of non-negative bins starts at q = 0, the run of negative ones at qs (the first q with k0 + q D >= npos; R: none).
    // Both starts are evaluated up front, side by side (four independent sin/cos evaluations), and the loop below only picks:
    // as a restart inside the unrolled loop the evaluation was emitted R times (code size) behind a divergent branch each.
    int qs = R;
#pragma unroll
    for (int q = R - 1; q >= 0; --q)
        if (k0 + q * D >= npos) qs = q;
    const double fa = (double)(qs == 0 ? k0 - op.N : k0);
    const double fb = (double)(k0 + (qs < R ? qs : 0) * D - op.N);
    double ca, sa, cva, sva, cb, sb, cvb, svb;
    cis_rad_d(op.cth * fa * fa, ca, sa);
    cis_rad_d(op.cth * (2.0 * fa * op.D + op.D * op.D), cva, sva);
    cis_rad_d(op.cth * fb * fb, cb, sb);
    cis_rad_d(op.cth * (2.0 * fb * op.D + op.D * op.D), cvb, svb);
    cx<double> u = mk<double>(op.mag * ca, op.mag * sa), vv = mk<double>(cva, sva);
    const cx<double> ub = mk<double>(op.mag * cb, op.mag * sb), vb = mk<double>(cvb, svb);
#pragma unroll
    for (int q = 0; q < R; ++q) {
        if (q > 0 && q == qs) {
            u = ub;
            vv = vb;
        }
        v[q] = mul_by_d(v[q], u);
        u = u * vv;
        vv = vv * op.c2;
    }
}

// one pass of radix R for this thread's butterflies; DIF: transform then twiddle, DIT: twiddle then transform.
// The twiddles cis(sign 2 pi j q / M) are a chain of products from the base, evaluated in double (see
// tw_powers for why single precision does not build it in float) and consumed as they are produced.
// STR: the transform's elements are `es` slots apart (the column stage keeps the C columns of a tile interleaved in LDS:
// col_mixed_body); rows are contiguous (STR = false: no multiplication by a run-time stride in their index arithmetic)
// IO (rows only, pass 0 -- the one whose block is the whole row): 1 = the butterflies read the row from global memory (`g`) instead
// of LDS, and `hook` runs once per thread after the first round's loads are issued (the row stage's control logic: it waits for
// other data and may end the launch -- it returns false, so does the pass); 2 = the butterflies write the row to global memory.
// Either way a row crosses LDS once less (round 6: no staging copy in front of the forward transform or behind the inverse one).
struct MixNoHook {
    SSF_HD bool operator()() const { return true; }
};
template <int SIGN, int R, bool DIF, bool STR = false, int IO = 0, typename T, class Ctx, class Hook = MixNoHook>
SSF_HD bool mix_pass(Ctx &ctx, const MixPlan &p, int i, int t, int nthreads, cx<T> *x, const cx<double> *wtab,
                     bool use_op, const MixRowOp &op, int es = 1, const cx<T> *gsrc = nullptr, cx<T> *gdst = nullptr, Hook hook = Hook()) {
    const int M = p.M[i], s = p.S[i], nbf = p.L / R, wstep = p.W[i];
    const int se = STR ? s * es : s;
    bool first = IO == 1;
    for (int bf = t; bf < nbf || first; bf += nthreads) {
        const bool act = bf < nbf;
        const int blk = act ? bf / s : 0, j = act ? bf - blk * s : 0;
        cx<T> *base = x + (STR ? (blk * M + j) * es : blk * M + j);
        cx<T> v[R];
        if constexpr (IO == 1) {
            const cx<T> *gb = gsrc + (blk * M + j);
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = gb[se * q];            // (an idle thread reads the first butterfly's values and drops them)
            if (first) {
                first = false;
                if (!hook()) return false;
            }
            if (!act) break;
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = base[se * q];
        }
        if constexpr (!DIF && R <= kMixMaxOpRadix) {               // (only given for the stride-1 pass: j = 0, bf = blk)
            if (use_op) mix_apply_op<R>(p, op, blk, v);
        }
        cx<double> w1 = mk<double>(1.0, 0.0);
        if (s > 1) {
            if (wtab) {                              // wtab[k] = cis(-2 pi k / L): one load instead of a sincospi
                const cx<double> w = wtab[wstep * j];
                w1 = mk<double>(w.re, SIGN < 0 ? w.im : -w.im);
            } else {
                double c, sn;
                cis2pi_d((double)(SIGN * j) / (double)M, c, sn);
                w1 = mk<double>(c, sn);
            }
        }
        // w^q, q = 1 .. R-1, as NCH interleaved chains w^(q + NCH) = w^q w^NCH (one chain of R - 1 dependent products is
        // pure latency: a thread has a single butterfly in flight); ch[q % NCH] holds the next power of its residue.
        // Four chains for the middle radices; the large ones (20, 25) leave no registers for more than one (scratch otherwise).
        constexpr int NCH = (R <= 5 || R >= 20) ? 1 : 4;
        cx<double> ch[4] = {w1, w1, w1, w1}, wn = w1;
        if (s > 1 && NCH > 1) {
            const cx<double> w2 = w1 * w1;
            if (NCH == 2) {
                wn = w2;
                ch[0] = w2;
            } else {
                ch[2] = w2;
                ch[3] = w2 * w1;
                wn = w2 * w2;
                ch[0] = wn;
            }
        }
        if (!DIF && s > 1) {                         // inputs are in natural order q
#pragma unroll
            for (int q = 1; q < R; ++q) {
                v[q] = mul_by_d(v[q], ch[q % NCH]);
                if (q + NCH < R) ch[q % NCH] = ch[q % NCH] * wn;
            }
        }
        dft_small<SIGN, R>(v);
        if (DIF && s > 1) {                          // outputs sit in slot order: walk the bins, pick the slot
#pragma unroll
            for (int kq = 1; kq < R; ++kq) {
                const int slot = dft_bin_slot<R>(kq);          // compile-time after unrolling
                v[slot] = mul_by_d(v[slot], ch[kq % NCH]);
                if (kq + NCH < R) ch[kq % NCH] = ch[kq % NCH] * wn;
            }
        }
        if constexpr (IO == 2) {
            cx<T> *gb = gdst + (blk * M + j);
#pragma unroll
            for (int sl = 0; sl < R; ++sl) gb[se * dft_slot_bin<R>(sl)] = v[sl];
        } else {
#pragma unroll
            for (int sl = 0; sl < R; ++sl) base[se * dft_slot_bin<R>(sl)] = v[sl];
        }
    }
    if constexpr (IO != 2) ctx.sync();
    return true;
}

// experiment builds: SSF_MIX_RADIX_SET = list of the radices the kernels carry (code-size experiments; a plan with another radix
// would fall into the smallest carried one: experiment libraries only)
#ifdef SSF_MIX_RADIX_SET
constexpr bool mix_has_radix(int r) {
    constexpr int set[] = {SSF_MIX_RADIX_SET};
    for (int q : set)
        if (q == r) return true;
    return false;
}
#else
constexpr bool mix_has_radix(int) { return true; }
#endif
#define SSF_MIX_CASE(R, ...) \
    case R:                  \
        if constexpr (mix_has_radix(R)) { __VA_ARGS__; break; }
template <int SIGN, bool DIF, bool STR = false, int IO = 0, typename T, class Ctx, class Hook = MixNoHook>
SSF_HD bool mix_pass_any(Ctx &ctx, const MixPlan &p, int i, int t, int nthreads, cx<T> *x, const cx<double> *wtab,
                         bool use_op, const MixRowOp &op, int es = 1, const cx<T> *gsrc = nullptr, cx<T> *gdst = nullptr, Hook hook = Hook()) {
    bool ok = true;
    switch (p.r[i]) {
    SSF_MIX_CASE(25, ok = mix_pass<SIGN, 25, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(20, ok = mix_pass<SIGN, 20, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(16, ok = mix_pass<SIGN, 16, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(15, ok = mix_pass<SIGN, 15, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(12, ok = mix_pass<SIGN, 12, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(10, ok = mix_pass<SIGN, 10, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(9, ok = mix_pass<SIGN, 9, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(8, ok = mix_pass<SIGN, 8, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(6, ok = mix_pass<SIGN, 6, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(5, ok = mix_pass<SIGN, 5, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(4, ok = mix_pass<SIGN, 4, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    SSF_MIX_CASE(3, ok = mix_pass<SIGN, 3, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook))
    default: ok = mix_pass<SIGN, 2, DIF, STR, IO>(ctx, p, i, t, nthreads, x, wtab, use_op, op, es, gsrc, gdst, hook); break;
    }
    return ok;
}

// x (L values in LDS, all threads of the transform past a barrier): forward, natural -> digit-reversed
template <int SIGN, typename T, class Ctx>
SSF_HD void mix_dif(Ctx &ctx, const MixPlan &p, int t, int nthreads, cx<T> *x, const cx<double> *wtab = nullptr) {
    const MixRowOp none{};
    for (int i = 0; i < p.npass; ++i) mix_pass_any<SIGN, true>(ctx, p, i, t, nthreads, x, wtab, false, none);
}
// inverse of mix_dif<-SIGN> (unscaled): digit-reversed -> natural; with_op: the spectrum is multiplied by the row
// operator `op` on the way in (first pass)
template <int SIGN, typename T, class Ctx>
SSF_HD void mix_dit(Ctx &ctx, const MixPlan &p, int t, int nthreads, cx<T> *x, const cx<double> *wtab, bool with_op,
                    const MixRowOp &op) {
    for (int i = p.npass - 1; i >= 0; --i)
        mix_pass_any<SIGN, false>(ctx, p, i, t, nthreads, x, wtab, with_op && i == p.npass - 1, op);
}
template <int SIGN, typename T, class Ctx>
SSF_HD void mix_dit(Ctx &ctx, const MixPlan &p, int t, int nthreads, cx<T> *x, const cx<double> *wtab = nullptr) {
    const MixRowOp none{};
    mix_dit<SIGN>(ctx, p, t, nthreads, x, wtab, false, none);
}
// Forward transform, row operator, inverse transform of a row in LDS.  The last forward pass and the first inverse pass (stride 1)
// work on the same R values of the same thread: they run as one pass, the values stay in registers between the two butterflies (one LDS
// round trip and one barrier less per row; same operations in the same order as mix_dif<-1> followed by mix_dit<+1> with the operator).
template <int R, typename T, class Ctx>
SSF_HD void mix_pass_mid(Ctx &ctx, const MixPlan &p, int t, int nthreads, cx<T> *x, const MixRowOp &op) {
    const int nbf = p.L / R;
    for (int bf = t; bf < nbf; bf += nthreads) {
        cx<T> *base = x + bf * R;
        cx<T> v[R], w[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = base[q];
        dft_small<-1, R>(v);
#pragma unroll
        for (int k = 0; k < R; ++k) w[k] = v[dft_bin_slot<R>(k)];      // slot order -> bin order (a renaming)
        mix_apply_op<R>(p, op, bf, w);
        dft_small<+1, R>(w);
#pragma unroll
        for (int sl = 0; sl < R; ++sl) base[dft_slot_bin<R>(sl)] = w[sl];
    }
    ctx.sync();
}
constexpr int kMixPerThread = 16;   // a row has at most 16 values per thread of its transform (fused_engine.h sizes the workgroups so)
#ifndef SSF_MIX_IO
#define SSF_MIX_IO 2      // bit 0: first forward pass reads the row from global memory; bit 1: last inverse pass writes it there
#endif
// (measured, round 6, row launch at 2 000 000 / 800 000: neither 66.7 / 35.8 us, bit 1 alone 65.2 / 35.4, bit 0 alone 79.2 / 43.2 -- the
//  control logic inlined behind the first butterflies' loads in every radix's body costs far more than the staging copy it saves)
// gin / gout: the row in global memory.  With at least two passes the last inverse pass writes it there, and -- experiment builds --
// the first forward pass reads it from there (mix_pass IO); `hook` as in mix_pass (it must leave `op` filled in: the stride-1 pass uses it).  Returns
// false when the hook ended the launch.  A single-pass row (shorter than any the engine asks for) is staged through LDS.
template <typename T, class Ctx, class Hook>
SSF_HD bool mix_dif_op_dit(Ctx &ctx, const MixPlan &p, int t, int nthreads, cx<T> *x, const cx<double> *wtab, const MixRowOp &op,
                           const cx<T> *gin, cx<T> *gout, Hook hook) {
    const MixRowOp none{};
    const int last = p.npass - 1;
    constexpr bool io_in = (SSF_MIX_IO & 1) != 0, io_out = (SSF_MIX_IO & 2) != 0;
    bool staged = true;
    if constexpr (io_in) {
        if (last > 0) {
            staged = false;
            if (!mix_pass_any<-1, true, false, 1>(ctx, p, 0, t, nthreads, x, wtab, false, none, 1, gin, (cx<T> *)nullptr, hook)) return false;
            for (int i = 1; i < last; ++i) mix_pass_any<-1, true>(ctx, p, i, t, nthreads, x, wtab, false, none);
        }
    }
    if (staged) {
        // (all of a thread's loads issued before the hook and before the first LDS store, straight-line: L <= kMixPerThread * nthreads is
        //  the caller's contract.  Measured, row launch at 2 000 000: a load-store loop with a run-time trip count 66.7 us, this 64.7 us,
        //  the same values in a chunk loop around the hook 83.6 us -- the array went to scratch memory.)
        cx<T> v[kMixPerThread];
#pragma unroll
        for (int m = 0; m < kMixPerThread; ++m) {
            const int i = t + nthreads * m;
            v[m] = i < p.L ? gin[i] : mk<T>((T)0, (T)0);
        }
        if (!hook()) return false;
#pragma unroll
        for (int m = 0; m < kMixPerThread; ++m) {
            const int i = t + nthreads * m;
            if (i < p.L) x[i] = v[m];
        }
        ctx.sync();
        ctx.mark(1);
        for (int i = 0; i < last; ++i) mix_pass_any<-1, true>(ctx, p, i, t, nthreads, x, wtab, false, none);
    }
    ctx.mark(2);
    switch (p.r[last]) {
    SSF_MIX_CASE(16, mix_pass_mid<16>(ctx, p, t, nthreads, x, op))
    SSF_MIX_CASE(15, mix_pass_mid<15>(ctx, p, t, nthreads, x, op))
    SSF_MIX_CASE(12, mix_pass_mid<12>(ctx, p, t, nthreads, x, op))
    SSF_MIX_CASE(10, mix_pass_mid<10>(ctx, p, t, nthreads, x, op))
    SSF_MIX_CASE(9, mix_pass_mid<9>(ctx, p, t, nthreads, x, op))
    SSF_MIX_CASE(8, mix_pass_mid<8>(ctx, p, t, nthreads, x, op))
    SSF_MIX_CASE(6, mix_pass_mid<6>(ctx, p, t, nthreads, x, op))
    SSF_MIX_CASE(5, mix_pass_mid<5>(ctx, p, t, nthreads, x, op))
    SSF_MIX_CASE(4, mix_pass_mid<4>(ctx, p, t, nthreads, x, op))
    SSF_MIX_CASE(3, mix_pass_mid<3>(ctx, p, t, nthreads, x, op))
    default: mix_pass_mid<2>(ctx, p, t, nthreads, x, op); break;
    }
    ctx.mark(3);
    bool direct = false;
    if constexpr (io_out) {
        if (last > 0) {
            direct = true;
            for (int i = last - 1; i >= 1; --i) mix_pass_any<+1, false>(ctx, p, i, t, nthreads, x, wtab, false, none);
            mix_pass_any<+1, false, false, 2>(ctx, p, 0, t, nthreads, x, wtab, false, none, 1, (const cx<T> *)nullptr, gout);
        }
    }
    if (!direct) {
        for (int i = last - 1; i >= 0; --i) mix_pass_any<+1, false>(ctx, p, i, t, nthreads, x, wtab, false, none);
        for (int i = t; i < p.L; i += nthreads) gout[i] = x[i];
    }
    return true;
}
// the same pair over elements `es` slots apart (column stage: the columns of a tile interleaved in LDS)
template <int SIGN, typename T, class Ctx>
SSF_HD void mix_dif_strided(Ctx &ctx, const MixPlan &p, int t, int nthreads, cx<T> *x, const cx<double> *wtab, int es) {
    const MixRowOp none{};
    for (int i = 0; i < p.npass; ++i) mix_pass_any<SIGN, true, true>(ctx, p, i, t, nthreads, x, wtab, false, none, es);
}
template <int SIGN, typename T, class Ctx>
SSF_HD void mix_dit_strided(Ctx &ctx, const MixPlan &p, int t, int nthreads, cx<T> *x, const cx<double> *wtab, int es) {
    const MixRowOp none{};
    for (int i = p.npass - 1; i >= 0; --i) mix_pass_any<SIGN, false, true>(ctx, p, i, t, nthreads, x, wtab, false, none, es);
}

}  // namespace fused
}  // namespace ssf
