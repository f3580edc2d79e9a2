// engine_fused_f64.hip -- the fused engine's kernels and classes for double fields: one of the two
// translation units engine_fused_impl.h is compiled in (they build in parallel; engine_fused.hip dispatches on the precision).
// Twiddle bases of this translation unit (cis of an exact fraction of a turn, two to four per thread and launch): own quarter-turn
// reduction in front of the small-argument kernels instead of the library sincospi -- config 2 + 1.0 % in 4 of 4 interleaved
// repetitions; in the single-precision unit the same is 1.3 % slower (profiles/r4_ab_code_size.txt), so only here.
#define SSF_CIS2PI_OWN 1
#include "engine_fused_impl.h"

namespace ssf {
Engine *make_fused_engine_f64(ssf_plan *plan) { return make_fused_engine_t<double>(plan); }
FusedConv *make_fused_conv_f64(ssf_plan *plan, int64_t M, int nrows) { return make_fused_conv_t<double>(plan, M, nrows); }
FusedRows *make_fused_rows_f64(ssf_plan *plan, int64_t N, int nrows) { return make_fused_rows_t<double>(plan, N, nrows); }
int fused_couple_reduce_selftest(int nranks, int npart, const double *parts, double *out5, std::string *err) {
    return couple_reduce_selftest_impl(nranks, npart, parts, out5, err);
}
int fused_overlap_save_f64(int64_t sigLen, int nrows, int log2nfft, int K, const void *Hfft, const void *in, void *out, std::string *err) {
    return overlap_save_t<double>(sigLen, nrows, log2nfft, K, Hfft, in, out, err);
}
}  // namespace ssf

#ifdef SSF_PHASE_TIMING
extern "C" int ssf_debug_marks_f64(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssf::g_marks), sizeof(unsigned long long) * 4 * 4096 * 8);
}
#endif
