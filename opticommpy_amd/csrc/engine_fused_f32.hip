// engine_fused_f32.hip -- the fused engine's kernels and classes for float fields (and the packed polarisation pairs of the complex64 Manakov path): one of the two
// translation units engine_fused_impl.h is compiled in (they build in parallel; engine_fused.hip dispatches on the precision).
#include "engine_fused_impl.h"

namespace ssf {
Engine *make_fused_engine_f32(ssf_plan *plan) { return make_fused_engine_t<float>(plan); }
FusedConv *make_fused_conv_f32(ssf_plan *plan, int64_t M, int nrows) { return make_fused_conv_t<float>(plan, M, nrows); }
FusedRows *make_fused_rows_f32(ssf_plan *plan, int64_t N, int nrows) { return make_fused_rows_t<float>(plan, N, nrows); }
int fused_overlap_save_f32(int64_t sigLen, int nrows, int log2nfft, int K, const void *Hfft, const void *in, void *out, std::string *err) {
    return overlap_save_t<float>(sigLen, nrows, log2nfft, K, Hfft, in, out, err);
}
}  // namespace ssf

#ifdef SSF_PHASE_TIMING
extern "C" int ssf_debug_marks_f32(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssf::g_marks), sizeof(unsigned long long) * 4 * 4096 * 8);
}
#endif
