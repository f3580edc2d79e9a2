// engine_fused.hip -- placeholder until the fused radix-2^n pipeline lands.
#include "ssf_internal.h"
namespace ssf {
bool fused_supports(int64_t, int, int) { return false; }
Engine *make_fused_engine(ssf_plan *plan) {
    plan->err = "fused engine not built";
    return nullptr;
}
}  // namespace ssf
