// engine_fused.hip -- entry points of the fused engine: dispatch on the field precision to the two translation units
// that hold the kernels (engine_fused_f64.hip, engine_fused_f32.hip: engine_fused_impl.h compiled once per precision, in
// parallel), and the length rule.
#include <vector>

#include "fused_engine.h"
#include "ssf_internal.h"

namespace ssf {

Engine *make_fused_engine_f64(ssf_plan *plan);
Engine *make_fused_engine_f32(ssf_plan *plan);
FusedConv *make_fused_conv_f64(ssf_plan *plan, int64_t M, int nrows);
FusedConv *make_fused_conv_f32(ssf_plan *plan, int64_t M, int nrows);
FusedRows *make_fused_rows_f64(ssf_plan *plan, int64_t N, int nrows);
FusedRows *make_fused_rows_f32(ssf_plan *plan, int64_t N, int nrows);
int fused_overlap_save_f64(int64_t sigLen, int nrows, int log2nfft, int K, const void *Hfft, const void *in, void *out, std::string *err);
int fused_overlap_save_f32(int64_t sigLen, int nrows, int log2nfft, int K, const void *Hfft, const void *in, void *out, std::string *err);

Engine *make_fused_engine(ssf_plan *plan) {
    return plan->precision == SSF_C128 ? make_fused_engine_f64(plan) : make_fused_engine_f32(plan);
}

FusedConv *make_fused_conv(ssf_plan *plan, int64_t M, int nrows) {
    return plan->precision == SSF_C128 ? make_fused_conv_f64(plan, M, nrows) : make_fused_conv_f32(plan, M, nrows);
}

bool fused_rows_supports(int64_t n) {
    if (n < 16 || n > fused::kMixMaxRow) return false;
    int64_t r = n;
    for (int q : {2, 3, 5})
        while (r % q == 0) r /= q;
    fused::MixPlan mp;
    return r == 1 && fused::mix_make_plan((int)n, &mp);
}

FusedRows *make_fused_rows(ssf_plan *plan, int64_t N, int nrows) {
    return plan->precision == SSF_C128 ? make_fused_rows_f64(plan, N, nrows) : make_fused_rows_f32(plan, N, nrows);
}

int fused_overlap_save(int device, int64_t sigLen, int nrows, int precision, int log2nfft, int K, const void *Hfft,
                       const void *in, void *out, std::string *err) {
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        *err = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return SSF_ERR_HIP;
    }
    return precision == SSF_C128 ? fused_overlap_save_f64(sigLen, nrows, log2nfft, K, Hfft, in, out, err)
                                 : fused_overlap_save_f32(sigLen, nrows, log2nfft, K, Hfft, in, out, err);
}

bool fused_supports(int64_t N, int nrows, int precision) {
    if (N >= 256 && (N & (N - 1)) && nrows >= 1) {          // 2^a 3^b 5^c: mixed-radix rows
        int l1, n2, n1, c;
        return fused::choose_mixed_split(N, precision, &l1, &n2) || fused::choose_mixed2_split(N, precision, &n1, &n2, &c);
    }
    if (N < 256 || (N & (N - 1)) || nrows < 1) return false;
    int l = 0;
    while ((1ll << l) < N) ++l;
    fused::Split s;
    return fused::choose_split(l, precision, &s);
}

}  // namespace ssf

#ifdef SSF_PHASE_TIMING
// phase stamps of the last launches (diagnostic build): the unit that ran has non-zero stamps
extern "C" int ssf_debug_marks_f64(unsigned long long *out);
extern "C" int ssf_debug_marks_f32(unsigned long long *out);
extern "C" int ssf_debug_marks(unsigned long long *out) {
    const size_t n = (size_t)4 * 4096 * 8;
    std::vector<unsigned long long> a(n), b(n);
    int rc = ssf_debug_marks_f64(a.data());
    if (!rc) rc = ssf_debug_marks_f32(b.data());
    for (size_t i = 0; i < n; ++i) out[i] = a[i] > b[i] ? a[i] : b[i];
    return rc;
}
#endif
