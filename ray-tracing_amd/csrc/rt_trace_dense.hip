// rt_trace_dense.hip -- instantiations of the persistent kernel that carry the pruning, lane-by-lane exact pass for
// scenes whose rays cross many spheres (rt_hit.hpp resolve_dense_words), and the stamped diagnostic instantiations
// (which carry it too).  A translation unit of its own because these kernels want the compiler's default instruction
// scheduling, while the kernels every other scene runs (rt_trace.hip) are 2 to 5 % faster with the strategy that
// minimises registers (csrc/FLAGS): it leaves those without spilled vector registers and these with three times as many.
#include "rt_trace_kernel.hpp"

namespace rt {

template <int KIND>
static const void *pick(bool profile, bool single) {
    constexpr bool kHasDense = KIND != kFetchTiled;  // the tiled kind keeps its slots in spatial order: no pruning form
    if (profile) return (const void *)trace_kernel<KIND, true, false, kHasDense>;
    if constexpr (kHasDense) {
        return single ? (const void *)trace_kernel<KIND, false, true, true> : (const void *)trace_kernel<KIND, false, false, true>;
    }
    return nullptr;
}

const void *trace_kernel_dense(int fetch_kind, bool profile, bool single) {
    switch (fetch_kind) {
        case kFetchLds: return pick<kFetchLds>(profile, single);
        case kFetchScalar: return pick<kFetchScalar>(profile, single);
        case kFetchStream: return pick<kFetchStream>(profile, single);
        case kFetchMatrix: return pick<kFetchMatrix>(profile, single);
        case kFetchTiled: return pick<kFetchTiled>(profile, single);
    }
    return nullptr;
}

}  // namespace rt
