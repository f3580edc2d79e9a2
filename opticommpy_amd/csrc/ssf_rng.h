// ssf_rng.h -- counter-based ASE noise for the EDFA span epilogue (devices.py:711-726):
// Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, counter = (sample index, row,
// span, 0), Box-Muller on the four 32-bit outputs -> one complex circular Gaussian sample
// per counter plus a spare.  Statistical parity only: the reference's
// own generators differ between its numpy, numba and cupy paths.
#pragma once
#include <cstdint>

#include "fused_core.h"

namespace ssf {

struct Philox4 {
    uint32_t v[4];
};

SSF_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

SSF_HD Philox4 philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// one complex sample CN(0, 2 sigma^2): real and imaginary parts ~ N(0, sigma^2)
SSF_HD void gauss_pair(uint64_t sample, uint32_t row, uint32_t span, uint64_t seed, double sigma, double &re, double &im) {
    const Philox4 r = philox4x32_10(sample, ((uint64_t)span << 32) | row, seed);
    const double u1 = ((double)r.v[0] + 0.5) * (1.0 / 4294967296.0);       // (0, 1)
    const double u2 = ((double)r.v[1] + 0.5) * (1.0 / 4294967296.0);
    const double rad = sigma * sqrt(-2.0 * log(u1));
    double c, s;
    fused::cis2pi_d(u2, c, s);
    re = rad * c;
    im = rad * s;
}

}  // namespace ssf
