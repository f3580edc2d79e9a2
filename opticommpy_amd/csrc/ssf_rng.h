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

// four unit normals from ONE Philox call, Box-Muller in single precision (the photodiodes' shot and thermal noise: eight normals
// per sample and polarisation -- in double precision, one pair per call, the generator was 85 % of the receiver's detection
// launch).  32-bit uniforms bound the tails at 6.7 sigma either way; the hardware's log / sin / cos (inputs in turns) on the
// device, libm in the CPU emulator: statistical parity.
SSF_HD void gauss_quad(uint64_t sample, uint32_t row, uint32_t span, uint64_t seed, float z[4]) {
    const Philox4 r = philox4x32_10(sample, ((uint64_t)span << 32) | row, seed);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float u1 = ((float)r.v[2 * j] + 0.5f) * (1.0f / 4294967296.0f);            // (0, 1]
        const float u2 = (float)(r.v[2 * j + 1] >> 8) * (1.0f / 16777216.0f);            // [0, 1): 24 bits, exact
#if defined(__HIP_DEVICE_COMPILE__)
        const float rad = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // -2 ln(u1) = -2 ln 2 * log2(u1)
        z[2 * j] = rad * __builtin_amdgcn_cosf(u2);
        z[2 * j + 1] = rad * __builtin_amdgcn_sinf(u2);
#else
        const float rad = sqrtf(-2.0f * logf(u1));
        z[2 * j] = rad * cosf(6.283185307179586f * u2);
        z[2 * j + 1] = rad * sinf(6.283185307179586f * u2);
#endif
    }
}

}  // namespace ssf
