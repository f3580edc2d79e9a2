// fused_core.h -- arithmetic core of the fused radix-2^n pipeline: complex POD, register
// butterflies (radix 2/4/8/16), the mixed-radix pass plan and its index maps.
//
// Everything here is `SSF_HD`: it compiles for gfx950 with hipcc AND as plain C++ with g++,
// so the very same source runs inside the CPU kernel emulator used by tests/ (tests/emu).
#pragma once
#include <cmath>
#include <cstdint>
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SSF_HD __host__ __device__ __forceinline__
#else
#define SSF_HD inline
#endif

namespace ssf {
namespace fused {

constexpr double kTwoPi = 6.28318530717958647692;

template <typename T> struct cx {
    T re, im;
};
template <typename T> SSF_HD cx<T> mk(T a, T b) {
    cx<T> r;
    r.re = a;
    r.im = b;
    return r;
}
template <typename T> SSF_HD cx<T> operator+(cx<T> a, cx<T> b) { return mk<T>(a.re + b.re, a.im + b.im); }
template <typename T> SSF_HD cx<T> operator-(cx<T> a, cx<T> b) { return mk<T>(a.re - b.re, a.im - b.im); }
template <typename T> SSF_HD cx<T> operator*(cx<T> a, cx<T> b) {
    return mk<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
template <typename T> SSF_HD cx<T> operator*(cx<T> a, T s) { return mk<T>(a.re * s, a.im * s); }
template <typename T> SSF_HD cx<T> conj(cx<T> a) { return mk<T>(a.re, -a.im); }
template <typename T> SSF_HD T norm2(cx<T> a) { return a.re * a.re + a.im * a.im; }
// a * (SIGN * j)
template <int SIGN, typename T> SSF_HD cx<T> mulj(cx<T> a) {
    return SIGN > 0 ? mk<T>(-a.im, a.re) : mk<T>(a.im, -a.re);
}

// ---------------------------------------------------------------------------------------
// Packed polarisation pair: single precision moves half the bytes of double precision for the same number of
// butterflies and twiddles, so the complex64 Manakov path carries BOTH polarisations of a sample in one
// element: re = (x.re, y.re), im = (x.im, y.im), two floats per register pair.  Every butterfly then is one
// packed instruction (v_pk_add_f32 / v_pk_fma_f32) for the two polarisations, twiddles and the linear operator
// are per-thread scalars shared by both, and |Ex|^2 + |Ey|^2 is local to the thread (no exchange through LDS).
// scalar_t<T> is the type of such a per-element scalar: T itself for float / double, float for the pair.
#if defined(__HIPCC__)
typedef float pf2 __attribute__((ext_vector_type(2)));
#else
typedef float pf2 __attribute__((vector_size(8)));
#endif
template <typename T> struct LaneOf { using S = T; };
template <> struct LaneOf<pf2> { using S = float; };
template <typename T> using scalar_t = typename LaneOf<T>::S;
template <typename T> SSF_HD T splat(scalar_t<T> a) { return a; }
template <> SSF_HD pf2 splat<pf2>(float a) {
    pf2 r;
    r[0] = a;
    r[1] = a;
    return r;
}
SSF_HD pf2 mk2(float a, float b) {
    pf2 r;
    r[0] = a;
    r[1] = b;
    return r;
}
// value (element type T) times a scalar complex factor (twiddle, operator, rotation)
template <typename T> SSF_HD cx<T> tmul(cx<T> v, cx<scalar_t<T>> w) {
    if constexpr (sizeof(T) == sizeof(scalar_t<T>)) return v * w;
    else {
        const T wr = splat<T>(w.re), wi = splat<T>(w.im);
        return mk<T>(v.re * wr - v.im * wi, v.re * wi + v.im * wr);
    }
}
// sum over the lanes of an element-wise quantity (the pair: both polarisations)
template <typename T> SSF_HD scalar_t<T> lane_sum(T a) { return a; }
template <> SSF_HD float lane_sum<pf2>(pf2 a) { return a[0] + a[1]; }
// fused multiply-add with a scalar factor, per lane
template <typename T> SSF_HD T fma_s(T x, scalar_t<T> a, T c);
template <> SSF_HD float fma_s<float>(float x, float a, float c) { return __builtin_fmaf(x, a, c); }
template <> SSF_HD double fma_s<double>(double x, double a, double c) { return __builtin_fma(x, a, c); }
template <> SSF_HD pf2 fma_s<pf2>(pf2 x, float a, pf2 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(x, splat<pf2>(a), c);
#else
    return mk2(__builtin_fmaf(x[0], a, c[0]), __builtin_fmaf(x[1], a, c[1]));
#endif
}

#ifndef SSF_CIS2PI_OWN
#define SSF_CIS2PI_OWN 0
#endif
SSF_HD double ksin_d(double x);
SSF_HD double kcos_d(double x);
// cis(2*pi*frac) evaluated in double (frac is exact: integer / power of two)
SSF_HD void cis2pi_d(double frac, double &c, double &s) {
#if defined(__HIP_DEVICE_COMPILE__) && SSF_CIS2PI_OWN
    // |frac| < 1 exact (integer / power of two): the quarter-turn reduction is exact too
    double t = frac - rint(frac);
    const double qd = rint(4.0 * t);
    const int q = (int)qd & 3;
    const double x = (t - 0.25 * qd) * kTwoPi;
    const double sx = ksin_d(x), cx_ = kcos_d(x);
    s = q == 0 ? sx : q == 1 ? cx_ : q == 2 ? -sx : -cx_;
    c = q == 0 ? cx_ : q == 1 ? -sx : q == 2 ? -cx_ : sx;
#elif defined(__HIP_DEVICE_COMPILE__)
    sincospi(2.0 * frac, &s, &c);
#else
    const double a = kTwoPi * frac;
    c = std::cos(a);
    s = std::sin(a);
#endif
}
SSF_HD void sincos_d(double a, double &s, double &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(a, &s, &c);
#else
    s = std::sin(a);
    c = std::cos(a);
#endif
}
SSF_HD void sincos_f(float a, float &s, float &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincosf(a, &s, &c);
#else
    s = std::sin(a);
    c = std::cos(a);
#endif
}
// x * 2^-k (exact)
SSF_HD double scale_pow2(double x, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ldexp(x, -k);
#else
    return std::ldexp(x, -k);
#endif
}
// Small-argument kernels: sin and cos on |x| <= pi/4 with no range reduction (the minimax
// polynomials of the freely distributable fdlibm __kernel_sin / __kernel_cos, < 1 ulp there).
// The nonlinear phase of a step and the phase increment between two iterates are small numbers
// (1e-3 ... 1e-1 rad), and the double-precision library sincos is ~10x the instructions: in the
// Manakov column stage it was half of the launch (phase timing: 6.5 us of 13 us).
SSF_HD double ksin_d(double x) {
    const double z = x * x;
    const double r = 8.33333333332248946124e-03 +
                     z * (-1.98412698298579493134e-04 +
                          z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    return x + x * (z * (-1.66666666666666324348e-01 + z * r));
}
SSF_HD double kcos_d(double x) {
    const double z = x * x;
    const double r = z * (4.16666666666666019037e-02 +
                          z * (-1.38888888888741095749e-03 +
                               z * (2.48015872894767294178e-05 +
                                    z * (-2.75573143513906633035e-07 +
                                         z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    return 1.0 - (0.5 * z - z * r);
}
constexpr double kQuarterPi = 0.78539816339744830962;
// cis(a) for an angle in radians.  Large |a|: reduced to one turn first instead of calling
// sincos (whose Payne-Hanek path bloats the kernel); the reduction error |a| * 2^-53 is the
// rounding a itself already carries
SSF_HD void cis_rad_d(double a, double &c, double &s) {
    if (fabs(a) <= kQuarterPi) {
        s = ksin_d(a);
        c = kcos_d(a);
        return;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    double t = a * 0.15915494309189533577;   // 1 / (2 pi)
    t -= rint(t);
    sincospi(2.0 * t, &s, &c);
#else
    c = std::cos(a);
    s = std::sin(a);
#endif
}
// sin(d / 2) without the generic sin()'s large-argument machinery
SSF_HD double sin_half_angle(double d) {
    if (fabs(d) <= 2.0 * kQuarterPi) return ksin_d(0.5 * d);
#if defined(__HIP_DEVICE_COMPILE__)
    double t = d * 0.15915494309189533577;    // d/2 = pi * (d / (2 pi))
    t -= 2.0 * rint(0.5 * t);                 // sinpi is 2-periodic
    return sinpi(t);
#else
    return std::sin(0.5 * d);
#endif
}
template <typename T> SSF_HD cx<T> cis2pi(double frac) {
    double c, s;
    cis2pi_d(frac, c, s);
    return mk<T>((T)c, (T)s);
}
// single precision: frac = m / 2^k with m < 2^24 is exact in float, so sincospif loses nothing
template <> SSF_HD cx<float> cis2pi<float>(double frac) {
#if defined(__HIP_DEVICE_COMPILE__)
    float c, s;
    sincospif(2.0f * (float)frac, &s, &c);
    return mk<float>(c, s);
#else
    double c, s;
    cis2pi_d(frac, c, s);
    return mk<float>((float)c, (float)s);
#endif
}
template <typename T> SSF_HD cx<T> cis_t(T a);
// The nonlinear rotation of the double-precision column stage: small angles as a rule (the kernels above), and for the rest an own
// quarter-turn reduction in front of the SAME kernels instead of cis_rad_d's library sincospi -- sixteen call sites per stage
// whose large-angle path practically never runs: k_col<double,8,3> is 10 % shorter and config 2 2.8 % faster for it
// (profiles/r4_ab_code_size.txt; code that never runs still costs instruction-cache lines).  The row stage's operator, whose two
// angles are always large, stays with cis_rad_d: there the library call is the faster one (config 3 - 1 % with this form).
template <> SSF_HD cx<double> cis_t<double>(double a) {
    double x = a;
    int q = 0;
    if (fabs(a) > kQuarterPi) {
        double t = a * 0.15915494309189533577;   // turns
        t -= rint(t);                            // [-1/2, 1/2]
        const double qd = rint(4.0 * t);         // nearest quarter turn
        q = (int)qd & 3;
        x = (t - 0.25 * qd) * kTwoPi;            // [-pi/4, pi/4]
    }
    const double sx = ksin_d(x), cx_ = kcos_d(x);
    double s = sx, c = cx_;
    if (q) {
        s = q == 1 ? cx_ : q == 2 ? -sx : -cx_;
        c = q == 1 ? -sx : q == 2 ? -cx_ : sx;
    }
    return mk<double>(c, s);
}
SSF_HD float ksin_f(float x) {      // |x| <= pi/4 (fdlibm __kernel_sinf / __kernel_cosf coefficients)
    const float z = x * x;
    return x + x * (z * (-1.6666667163e-01f + z * (8.3333337680e-03f + z * (-1.9841270114e-04f + z * 2.7557314297e-06f))));
}
SSF_HD float kcos_f(float x) {
    const float z = x * x;
    const float r = z * (4.1666667908e-02f + z * (-1.3888889225e-03f + z * (2.4801587642e-05f + z * -2.7557314297e-07f)));
    return 1.0f - (0.5f * z - z * r);
}
template <> SSF_HD cx<float> cis_t<float>(float a) {
    if (fabsf(a) <= (float)kQuarterPi) return mk<float>(kcos_f(a), ksin_f(a));
#if defined(__HIP_DEVICE_COMPILE__)
    // large |a| (rare: a nonlinear phase above 45 degrees per step): reduced to one turn in double, then sincospif -- sincosf's own
    // large-argument reduction is ~340 instructions per call site, 16 sites unrolled per stage: a third of the packed column
    // kernel's code for a path that never runs, and code that never runs still costs instruction-cache lines (config 3 + 3 %:
    // profiles/r4_ab_code_size.txt)
    double t = (double)a * 0.15915494309189533577;   // 1 / (2 pi)
    t -= rint(t);
    float s, c;
    sincospif((float)(2.0 * t), &s, &c);
    return mk<float>(c, s);
#else
    float s, c;
    sincos_f(a, s, c);
    return mk<float>(c, s);
#endif
}

// ---------------------------------------------------------------------------------------
// register butterflies: in-place DFT of R values, natural order in and out,
// X[s] = sum_q x[q] * cis(SIGN * 2 pi q s / R).   SIGN = -1 forward, +1 inverse (unscaled).
// ---------------------------------------------------------------------------------------
template <int SIGN, typename T> SSF_HD void dft2(cx<T> &a, cx<T> &b) {
    const cx<T> t = a - b;
    a = a + b;
    b = t;
}
template <int SIGN, typename T> SSF_HD void dft4(cx<T> &a0, cx<T> &a1, cx<T> &a2, cx<T> &a3) {
    const cx<T> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mulj<SIGN>(a1 - a3);
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = t1 + t3;
    a3 = t1 - t3;
}
// Butterfly constants as a (hi, lo) pair: in single precision the float nearest to sqrt(1/2) or
// cos/sin(pi/8) is off by up to 3e-8, always in the same direction, so every rotation by W8 / W16
// would shrink (or grow) the field a little -- a bias that adds up coherently over 10^4 steps x 35
// passes (measured: -0.26 % power after 2000 steps).  x*hi + x*lo restores the constant to ~1e-15.
template <typename T> struct KConst;
template <> struct KConst<double> {
    static constexpr double h_hi = 0.70710678118654752440, h_lo = 0.0;
    static constexpr double c_hi = 0.92387953251128675613, c_lo = 0.0;
    static constexpr double s_hi = 0.38268343236508977173, s_lo = 0.0;
};
template <> struct KConst<float> {
    static constexpr float h_hi = 0.70710678118654752440f, h_lo = (float)(0.70710678118654752440 - (double)0.70710678118654752440f);
    static constexpr float c_hi = 0.92387953251128675613f, c_lo = (float)(0.92387953251128675613 - (double)0.92387953251128675613f);
    static constexpr float s_hi = 0.38268343236508977173f, s_lo = (float)(0.38268343236508977173 - (double)0.38268343236508977173f);
};
// (a true fused multiply-add is essential: the correction x*lo is below half an ulp of x*hi)
template <typename T> SSF_HD T mul_h(T x) {
    using K = KConst<scalar_t<T>>;
    if constexpr (sizeof(scalar_t<T>) == 8) return x * K::h_hi;
    else return fma_s<T>(x, K::h_hi, x * K::h_lo);
}
template <typename T> SSF_HD T mul_c(T x) {
    using K = KConst<scalar_t<T>>;
    if constexpr (sizeof(scalar_t<T>) == 8) return x * K::c_hi;
    else return fma_s<T>(x, K::c_hi, x * K::c_lo);
}
template <typename T> SSF_HD T mul_s(T x) {
    using K = KConst<scalar_t<T>>;
    if constexpr (sizeof(scalar_t<T>) == 8) return x * K::s_hi;
    else return fma_s<T>(x, K::s_hi, x * K::s_lo);
}
// a * (cr + j*SIGN*ci) with cr, ci in {cos(pi/8), sin(pi/8)} up to sign: CR/CI select (+-)c or (+-)s
template <int SIGN, int CR, int CI, typename T> SSF_HD cx<T> rot16(cx<T> a) {
    // CR, CI: +1 = +cos(pi/8), -1 = -cos(pi/8), +2 = +sin(pi/8), -2 = -sin(pi/8)
    const T rr = (CR == 1 || CR == -1) ? mul_c(a.re) : mul_s(a.re);
    const T ir = (CR == 1 || CR == -1) ? mul_c(a.im) : mul_s(a.im);
    const T ri = (CI == 1 || CI == -1) ? mul_c(a.re) : mul_s(a.re);
    const T ii = (CI == 1 || CI == -1) ? mul_c(a.im) : mul_s(a.im);
    using S = scalar_t<T>;
    const S sr = CR > 0 ? (S)1 : (S)-1, si = (CI > 0 ? (S)1 : (S)-1) * (S)SIGN;
    return mk<T>(sr * rr - si * ii, sr * ir + si * ri);
}

template <int SIGN, typename T> SSF_HD void dft8(cx<T> *v) {   // v[0..7]
    constexpr scalar_t<T> kS = (scalar_t<T>)SIGN;
    cx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cx<T> o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4<SIGN>(e0, e1, e2, e3);
    dft4<SIGN>(o0, o1, o2, o3);
    // o_s *= W8^s
    o1 = mk<T>(mul_h(o1.re - kS * o1.im), mul_h(o1.im + kS * o1.re));           // (1 + SIGN j)/sqrt2
    o2 = mulj<SIGN>(o2);
    o3 = mk<T>(mul_h(-o3.re - kS * o3.im), mul_h(-o3.im + kS * o3.re));         // (-1 + SIGN j)/sqrt2
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
}
template <int SIGN, typename T> SSF_HD void dft16(cx<T> *v) {  // v[0..15]
    constexpr scalar_t<T> kS = (scalar_t<T>)SIGN;
    cx<T> e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        e[i] = v[2 * i];
        o[i] = v[2 * i + 1];
    }
    dft8<SIGN>(e);
    dft8<SIGN>(o);
    // o_s *= W16^s = cis(SIGN * pi s / 8)
    o[1] = rot16<SIGN, +1, +2>(o[1]);                                                  // ( c, SIGN s)
    o[2] = mk<T>(mul_h(o[2].re - kS * o[2].im), mul_h(o[2].im + kS * o[2].re));
    o[3] = rot16<SIGN, +2, +1>(o[3]);                                                  // ( s, SIGN c)
    o[4] = mulj<SIGN>(o[4]);
    o[5] = rot16<SIGN, -2, +1>(o[5]);                                                  // (-s, SIGN c)
    o[6] = mk<T>(mul_h(-o[6].re - kS * o[6].im), mul_h(-o[6].im + kS * o[6].re));
    o[7] = rot16<SIGN, -1, +2>(o[7]);                                                  // (-c, SIGN s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = e[i] + o[i];
        v[i + 8] = e[i] - o[i];
    }
}

// powers w^0..w^15 of a unit phasor, multiplication depth <= 4
template <typename T> SSF_HD void powers16(cx<T> w, cx<T> *p) {
    p[0] = mk<T>((T)1, (T)0);
    p[1] = w;
    p[2] = w * w;
    p[3] = p[2] * w;
    p[4] = p[2] * p[2];
    p[5] = p[4] * w;
    p[6] = p[3] * p[3];
    p[7] = p[4] * p[3];
    p[8] = p[4] * p[4];
    p[9] = p[8] * w;
    p[10] = p[8] * p[2];
    p[11] = p[8] * p[3];
    p[12] = p[8] * p[4];
    p[13] = p[8] * p[5];
    p[14] = p[8] * p[6];
    p[15] = p[8] * p[7];
}

// ---------------------------------------------------------------------------------------
// Single precision (one row per polarisation and packed pairs): apply unit factors (twiddles, operator) as hi + lo float pairs.  A factor rounded to
// one float is off by up to 3e-8 in magnitude and phase, and it is the SAME error at every step (the twiddle of a given
// butterfly never changes), so over 10^4 steps the errors add up coherently: -1.2e-7 of power per step and a spectral
// ripple of 1e-3 after BASELINE config 3's 10 010 steps (the reference's own complex64 path, pocketfft with float
// twiddles, drifts by 5e-4 there).  With the low part the effective factor is exact to 1e-15 and only the random
// rounding of the products is left, which grows with the square root of the step count.
#ifndef SSF_C64_HILO
#define SSF_C64_HILO 1
#endif
// v * h for a factor given in double precision
template <typename T> SSF_HD cx<T> mul_by_d(cx<T> v, cx<double> h) {
    using S = scalar_t<T>;
    if constexpr (sizeof(S) == 8) {
        return v * mk<T>((T)h.re, (T)h.im);
    } else if constexpr (SSF_C64_HILO) {                                    // float and packed float pairs alike
        const S hr = (S)h.re, hi = (S)h.im;
        const S lr = (S)(h.re - (double)hr), li = (S)(h.im - (double)hi);
        const T cr = fma_s<T>(v.re, lr, -(v.im * splat<T>(li)));            // v.re lr - v.im li
        const T ci = fma_s<T>(v.re, li, v.im * splat<T>(lr));               // v.re li + v.im lr
        return mk<T>(fma_s<T>(v.re, hr, fma_s<T>(-v.im, hi, cr)), fma_s<T>(v.re, hi, fma_s<T>(v.im, hr, ci)));
    } else {
        return tmul(v, mk<S>((S)h.re, (S)h.im));
    }
}
// ---------------------------------------------------------------------------------------
// Twiddle tables.  The twiddles of a pass never change: generating them in every launch -- sincospi, a double-precision power
// tree and, in single precision, the hi + lo split of every factor -- is half of the complex64 row kernel's vector instructions
// (1 905 double-precision-rate instructions next to 1 896 packed ones, tools/kernel_mix.py) and a fifth of the complex128
// one's.  An entry holds cis(-2 pi j s / L_i) in the form the kernels apply it: the factor itself in double precision, the
// (hi.re, hi.im, lo.re, lo.im) float quadruple of mul_by_d in single precision (16 bytes either way); the inverse transforms
// use the conjugate (sign flips, free).
struct TwEntryF {
    float hr, hi, lr, li;
};
template <typename T> using tw_entry_t = typename std::conditional<sizeof(scalar_t<T>) == 8, cx<double>, TwEntryF>::type;
template <typename T> SSF_HD tw_entry_t<T> tw_make(cx<double> h) {
    if constexpr (sizeof(scalar_t<T>) == 8) return h;
    else {
        TwEntryF e;
        e.hr = (float)h.re;
        e.hi = (float)h.im;
        e.lr = SSF_C64_HILO ? (float)(h.re - (double)e.hr) : 0.0f;
        e.li = SSF_C64_HILO ? (float)(h.im - (double)e.hi) : 0.0f;
        return e;
    }
}
// v * e (CONJ: v * conj(e))
template <bool CONJ, typename T> SSF_HD cx<T> tw_mul(cx<T> v, const tw_entry_t<T> &e) {
    using S = scalar_t<T>;
    if constexpr (sizeof(S) == 8) {
        return v * mk<T>((T)e.re, CONJ ? (T)-e.im : (T)e.im);
    } else {
        const S hr = e.hr, hi = CONJ ? -e.hi : e.hi, lr = e.lr, li = CONJ ? -e.li : e.li;
        if constexpr (SSF_C64_HILO) {
            const T cr = fma_s<T>(v.re, lr, -(v.im * splat<T>(li)));
            const T ci = fma_s<T>(v.re, li, v.im * splat<T>(lr));
            return mk<T>(fma_s<T>(v.re, hr, fma_s<T>(-v.im, hi, cr)), fma_s<T>(v.re, hi, fma_s<T>(v.im, hr, ci)));
        } else {
            return tmul(v, mk<S>(hr, hi));
        }
    }
}
// a real constant given in double precision (radix-3 / 5 butterfly constants of the mixed-radix rows): x * c
template <typename T> SSF_HD T mul_cd(T x, double c) {
    using S = scalar_t<T>;
    if constexpr (sizeof(S) == 8) return x * (T)c;
    else if constexpr (SSF_C64_HILO) {
        const S hi = (S)c, lo = (S)(c - (double)hi);
        return fma_s<T>(x, hi, x * splat<T>(lo));
    } else return x * splat<T>((S)c);
}


// ---------------------------------------------------------------------------------------
// Mixed-radix pass plan for one length-L transform (L = 2^m, 16 <= L <= 65536), V = 2^lgV values
// per thread (16, or 8 for the 128-register kernels that run four waves per SIMD), L/V threads per transform.
//   radices r_0..r_{p-1}:  [V, (2^rem), V, V, ...]   with rem = (m - lgV) mod lgV
//   L_0 = L, L_{i+1} = L_i / r_i   (L_p = 1)
// Pass i works in place on positions  block*L_i + j + L_{i+1}*q  (q = 0..r_i-1),
// butterfly id bb = block*L_{i+1} + j  in [0, L / r_i).
// DIF(natural in) leaves X[k] at position pos with k = rev(pos):
//   pos = sum_i s_i * L_{i+1},   k = sum_i s_i * (r_0 ... r_{i-1}).
// ---------------------------------------------------------------------------------------
struct PassPlan {
    int L, log2L, npass, tpf;      // tpf = threads per transform = L/V
    int lgV;                       // log2 of the values per thread
    unsigned lg_pk;                // 4 bits per pass: log2 r_i
    unsigned long long lgLn_pk;    // 8 bits per pass: log2 L_{i+1} (stride of pass i)
    // packed (not arrays) so that a runtime pass index never forces the plan into scratch memory
    SSF_HD int lg(int i) const { return (int)((lg_pk >> (4 * i)) & 15u); }
    SSF_HD int r(int i) const { return 1 << lg(i); }
    SSF_HD int lgLn(int i) const { return (int)((lgLn_pk >> (8 * i)) & 255ull); }
};

SSF_HD PassPlan make_plan(int log2L, int lgV = 4) {
    PassPlan p;
    p.L = 1 << log2L;
    p.log2L = log2L;
    p.lgV = lgV;
    p.tpf = p.L >> lgV;
    p.lg_pk = 0;
    p.lgLn_pk = 0;
    int n = 0, left = log2L;
    const int rem = (log2L - lgV) % lgV, n16 = (log2L - lgV) / lgV;
    for (int i = 0; i < 2 + n16; ++i) {
        int lg;
        if (i == 0) lg = lgV;
        else if (i == 1) lg = rem;
        else lg = lgV;
        if (lg == 0) continue;
        left -= lg;
        p.lg_pk |= (unsigned)lg << (4 * n);
        p.lgLn_pk |= (unsigned long long)left << (8 * n);
        ++n;
    }
    p.npass = n;
    return p;
}

// position (element index inside the transform) of value q of butterfly bb in pass i
SSF_HD int pass_pos(const PassPlan &p, int i, int bb, int q) {
    const int lgS = p.lgLn(i);                 // stride L_{i+1}
    const int j = bb & ((1 << lgS) - 1);
    const int block = bb >> lgS;
    return (block << (lgS + p.lg(i))) + j + (q << lgS);
}
// j (twiddle index) of butterfly bb in pass i, and log2 L_i
SSF_HD int pass_j(const PassPlan &p, int i, int bb) { return bb & ((1 << p.lgLn(i)) - 1); }
SSF_HD int pass_lgLi(const PassPlan &p, int i) { return p.lgLn(i) + p.lg(i); }

// digit reversal: position -> natural index
SSF_HD int rev_pos(const PassPlan &p, int pos) {
    int k = 0, shift = 0;
    for (int i = 0; i < p.npass; ++i) {
        const int s = (pos >> p.lgLn(i)) & (p.r(i) - 1);
        k += s << shift;
        shift += p.lg(i);
    }
    return k;
}

// LDS slot of transform-local position pos (pad one slot per 16 to spread banks)
SSF_HD int lds_slot(int pos) { return pos + (pos >> 4); }
SSF_HD int lds_slots_per_fft(int L) { return L + (L >> 4) + 1; }
// Column kernels: C transforms side by side in LDS, lane l of a wave works on column l % C, butterfly l / C.  A 16-lane
// group of a 16-byte LDS access then touches (16 / C) consecutive slots of each of C columns: they fall into 16 different
// 16-byte bank groups when the column stride is 16 / C slots modulo 16 (with the "+ 1" stride above, C = 8 columns leave
// 9 distinct groups for 16 lanes, C = 4 only 7: SQ_LDS_BANK_CONFLICT 2.3 x SQ_ACTIVE_INST_LDS in k_col_pk<10>).
// Measured (same box, A/B): C = 4 (packed pairs, columns of 1024) column launch 62.0 -> 61.0 us, +1 % steps/s at config 3;
// C = 8 (complex128 Manakov) no gain within the noise (5 768 vs 5 672 steps/s the other way): applied to C <= 4 only.
#ifndef SSF_COLPAD
#define SSF_COLPAD 1
#endif
SSF_HD int lds_col_stride(int L, int C, int elem_bytes) {
    if (!SSF_COLPAD || elem_bytes != 16 || C > 4 || C < 2) return lds_slots_per_fft(L);
    const int want = 16 / C, s = L + (L >> 4);
    return s + ((want - s) & 15);
}

}  // namespace fused
}  // namespace ssf
