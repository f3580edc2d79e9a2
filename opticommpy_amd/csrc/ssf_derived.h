// ssf_derived.h -- constants the reference derives at the top of each call; plain C++ (no HIP
// headers) so that the CPU kernel emulator in tests/emu can share it.
#pragma once
#include <cmath>
#include <cstdint>

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

}  // namespace ssf
