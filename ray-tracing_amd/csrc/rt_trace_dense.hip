// rt_trace_dense.hip -- instantiations of the persistent kernel that carry the pruning, lane-by-lane exact pass for
// scenes whose rays cross many spheres (rt_hit.hpp resolve_dense_words), the stamped diagnostic instantiations (which
// carry it too), and the plain kernels of the vector-filter kinds for scenes of many spheres.  A translation unit of its
// own because these kernels lose with the scheduling strategy the others are compiled with (csrc/FLAGS: the one that
// minimises registers leaves the staged matrix kind without spilled vector registers and 2 % faster, these with three
// times as many spills and 7 to 11 % slower).
#include "rt_trace_kernel.hpp"

namespace rt {

template <int KIND>
static const void *pick(bool profile, bool single) {
    constexpr bool kHasDense = !Fetch<KIND>::kTiled;  // the tiled kinds keep their slots in spatial order: no pruning form
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
        case kFetchTiledDirect: return pick<kFetchTiledDirect>(profile, single);
    }
    return nullptr;
}

template <int KIND>
static const void *pick_plain(bool single) {
    return single ? (const void *)trace_kernel<KIND, false, true, false, false, 1> : (const void *)trace_kernel<KIND, false, false, false, false, 1>;
}

const void *trace_kernel_many(int fetch_kind, bool single) {
    switch (fetch_kind) {
        case kFetchLds: return pick_plain<kFetchLds>(single);
        case kFetchScalar: return pick_plain<kFetchScalar>(single);
        case kFetchStream: return pick_plain<kFetchStream>(single);
    }
    return nullptr;
}

}  // namespace rt
