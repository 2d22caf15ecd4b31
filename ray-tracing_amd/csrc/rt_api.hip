// rt_api.hip -- the C-ABI of librt_amd.so (declared in include/rt_amd.h): context, render target, scene
// upload and compaction, and the calls that enqueue the kernels of rt_kernels.hip.
//
// Host-side mirror of what the reference's MetalRenderer does around its compute dispatch
// (reference: "Ray Tracer/ContentView.swift":328-378, 661-709).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_kernels.h"

using namespace rt;

namespace {
template <typename T>
struct DevBuf {
    T *ptr = nullptr;
    size_t cap = 0;  // elements
};
// What the staged matrix kind needs to run its filter at half width (TraceParams::narrow ..., narrow_setup).
struct NarrowSet {
    bool ok = false;
    float oo = 0.0f;
    V3 mid = {0.0f, 0.0f, 0.0f}, half = {0.0f, 0.0f, 0.0f};
    int words = 0;
    uint32_t word[kWideWords] = {}, test[kWideWords] = {}, always[kWideWords] = {};
};
}  // namespace

// Scratch of rt_render_noalloc: device memory that belongs to the library itself, so that entry point never has to
// allocate.  Layout (bytes): [0, 256) work and segment counters, [kScratchQueueOffset, + kScratchQueueBytes) camera-ray
// queues, [kScratchPlaneOffset, + kScratchPlaneBytes) partial-sum planes.  Every GPU the library runs on holds its own
// copy of the array, at its own address, until the process ends.  In host code the array's name is only a handle for
// hipGetSymbolAddress; the address the kernels are given is looked up once per device (scratch_for_device).
namespace {
constexpr size_t kScratchQueueOffset = 256;
constexpr size_t kScratchQueueBytes = (size_t)64 << 20;   // 32 768 lanes x 64 rays x 32 bytes
static_assert(kScratchQueueBytes % ((size_t)1024 * kQueueRays * 2 * sizeof(float4)) == 0, "whole workgroups of up to 1024 threads");
constexpr size_t kScratchPlaneOffset = kScratchQueueOffset + kScratchQueueBytes;
constexpr size_t kScratchPlaneBytes = RT_NOALLOC_PLANE_BYTES;   // chunk planes x width x height x 16 bytes
constexpr size_t kScratchBytes = kScratchPlaneOffset + kScratchPlaneBytes;
constexpr int kScratchMaxDevices = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
// rt_render_tile: largest set of per-sample planes (tile pixels x samples x 16 bytes) it will allocate
constexpr size_t kTilePlaneBytesMax = (size_t)512 << 20;

struct rt_ctx {
    int device = 0;
    int num_cus = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    int width = 0, height = 0;
    DevBuf<float4> target, accum, planes;
    DevBuf<char> wf_pool;
    DevBuf<float4> rayq;  // camera-ray queues of the trace kernel
    DevBuf<char> staging; // device side of rt_present_bgra8, rt_first_hit and the self-tests; grows, never shrinks
    uint32_t *wf_host = nullptr;  // pinned host words the wavefront loop reads its queue counters into

    // compact scene
    DevBuf<float4> sph, sphc;
    DevBuf<_Float16> sphm;  // matrix-core filter: lifted A operands (rt_lift.hpp)
    DevBuf<float4> tiles, tile_frames;  // tiled matrix-core kind (TraceParams::tiles)
    DevBuf<float4> round_runs;          // the run of shells of every resolve round (TraceParams::round_runs)
    DevBuf<float4> word_bounds;         // balls around and inside every 32 sphere slots (TraceParams::word_bounds)
    DevBuf<int> tile_obj;
    DevBuf<uint16_t> tile_rank;
    int ntiles = 0;
    bool scene_dense = false;  // lines through the scene cross many spheres (dense_scene): kernels with the lane-by-lane exact pass
    bool tiles_ok = false;     // all but a few spheres can be filtered in their tile (the others go to the exact test for every ray)
    LiftFrame lift = {0.0f, 0.0f, 0.0f, 1.0f};
    float oo_limit_m = -1.0f;  // origin limit of the matrix-core filter in the lifted frame; < 0: scene not eligible
    int matrix_outliers = 0;   // spheres the matrix-core filter always passes on to the exact test
    NarrowSet narrow;          // the staged matrix kind at half width (narrow_setup)
    DevBuf<int> sph_obj;
    DevBuf<QuadDev> quads;
    DevBuf<MatDev> mats;
    int nsph = 0, ngroups = 0, nquad = 0, nobj = 0;
    float oo_limit = -1.0f;
    bool scene_resident = false;
    std::vector<unsigned char> uploaded;  // bytes of the Object array the device arrays were built from
    bool uploaded_valid = false;

    uint32_t *counters = nullptr;  // 256 B: [0] work counter (u32), [2..3] segment counter (u64)
    unsigned char *scratch_dev = nullptr;  // this device's copy of the static scratch array (rt_render_noalloc)
    uint32_t *stats_src = nullptr;  // counter block the last trace launch wrote (counters, or the head of scratch_dev)
    static constexpr int kRing = 64;
    hipEvent_t ev0[kRing] = {}, ev1[kRing] = {};
    unsigned long long launches = 0;    // trace launches so far
    unsigned long long times_read = 0;  // launches already reported by rt_trace_times
    bool stats_pending = false;
    bool stats_timed = false;  // the pending launch was bracketed by timing events (not when it was captured into a graph)

    int opt_chunk = -1, opt_tail_chunk = -1, opt_tail_samples = -1;
    int opt_kernel = 0;
    int opt_fetch = 0;
    int opt_blocks_per_cu = 0;
    int opt_filter = 0;
    int opt_resolve = 0;
    int opt_regroup = 0;
    int opt_plane = -1;
    int opt_share = 1;   // RT_OPT_SHARE
    int opt_narrow = 1;  // RT_OPT_NARROW
    int opt_wf_slots = 1 << 20;

    rt_stats last = {};
    std::string err;
};

namespace {

// Largest sphere count the optional LDS staging accepts (two 16-byte records each): 1792 spheres = 56 KiB,
// which with the 16 KiB of candidate words leaves room for two workgroups per compute unit in the 160 KiB LDS.
constexpr int kMaxLdsSpheres = 1792;
// Matrix-core filter: 80 bytes per sphere (lifted operand + strict record) next to 4 KiB of candidate words per
// wave; 1024 spheres = 80 KiB leave room for a 1024-thread workgroup's 64 KiB of words in the 160 KiB LDS.
constexpr int kMaxMatrixSpheres = 1024;

int fail(rt_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
    if (c) {
        c->err = what;
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

// Device address of the static scratch array on `device`, which must be the current device.  Looked up once per
// device and kept: each GPU holds its own copy of the array at its own address.
hipError_t scratch_for_device(int device, unsigned char **out) {
    static std::mutex lock;
    static unsigned char *table[kScratchMaxDevices] = {};
    if (device < 0 || device >= kScratchMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> guard(lock);
    if (!table[device]) {
        unsigned char *p = nullptr;
        hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        table[device] = p;
    }
    *out = table[device];
    return hipSuccess;
}

#define RT_HIP(call)                                                    \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return fail(ctx, RT_ERR_HIP, #call, e_);  \
    } while (0)

#define RT_TRY(expr)                   \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ < 0) return rc_;       \
    } while (0)

struct Bind {  // make the context's device current for the duration of a call
    int prev = -1;
    explicit Bind(rt_ctx *c) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) (void)hipSetDevice(c->device);
    }
    ~Bind() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

V3 ld3(const rt_float3 &a) { V3 r; r.x = a.x; r.y = a.y; r.z = a.z; return r; }
V4 ld4(const rt_float4 &a) { V4 r; r.x = a.x; r.y = a.y; r.z = a.z; r.w = a.w; return r; }

template <typename T>
int grow(rt_ctx *ctx, DevBuf<T> &b, size_t need) {
    if (need <= b.cap) return RT_OK;
    if (b.ptr) RT_HIP(hipFree(b.ptr));
    b.ptr = nullptr;
    b.cap = 0;
    RT_HIP(hipMalloc((void **)&b.ptr, need * sizeof(T)));
    b.cap = need;
    return RT_OK;
}

template <typename T>
int upload(rt_ctx *ctx, DevBuf<T> &b, const std::vector<T> &h) {
    RT_TRY(grow(ctx, b, h.size() ? h.size() : 1));
    if (!h.empty()) RT_HIP(hipMemcpyAsync(b.ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return RT_OK;
}

// Tiles of the tiled matrix-core kind (rt_hit.hpp, closest_sphere_tiled).  Spheres of ordinary size are partitioned
// by recursive median splits along the longest axis of their centres' bounding box into tiles of at most 512 that
// are also small against their spheres -- half-diagonal <= 128 median radii, so that a sphere's own share of the
// lifted form's relative slack, 2^-16 |c'|^2, stays below a quarter of r^2 (down to 64 spheres per tile in sparse
// scenes).  Each tile is lifted (rt_lift.hpp) in a frame centred on the tile, with one power-of-two scale for the
// whole scene chosen so that any line passing within 1.5 scene diagonals of a tile's centre stays inside the FP16
// range.  Spheres that dwarf the rest (a ground sphere) ride in the last tile; no frame fits them, so their records
// make them candidates for every ray.
struct TileSet {
    std::vector<float4> tiles, frames;
    std::vector<int> obj;
    std::vector<uint16_t> rank;   // TraceParams::tile_rank
    int ntiles = 0;
    bool ok = false;
};
struct TileBuilder {
    const std::vector<float4> &sph;
    std::vector<int> ids;                       // regular spheres, permuted in place
    std::vector<std::pair<int, int>> leaves;    // [begin, end) into ids
    void split(int begin, int end) {
        const int n = end - begin;
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        std::vector<float> radii;
        for (int i = begin; i < end; ++i) {
            const float4 &q = sph[(size_t)ids[(size_t)i]];
            const double c[3] = {q.x, q.y, q.z};
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
            radii.push_back(q.w);
        }
        std::nth_element(radii.begin(), radii.begin() + n / 2, radii.end());
        const double rmed = radii[(size_t)(n / 2)];
        double diag2 = 0.0;
        int axis = 0;
        for (int a = 0; a < 3; ++a) {
            diag2 += (hi[a] - lo[a]) * (hi[a] - lo[a]);
            if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
        }
        const bool compact = 0.5 * std::sqrt(diag2) <= 128.0 * rmed;
        if (n <= 64 || (n <= kTileSpheres && compact)) {
            leaves.push_back({begin, end});
            return;
        }
        // full tiles on the left as long as the count decides; halves rounded to whole 64-sphere blocks below that
        int left;
        if (n > kTileSpheres) {
            const int nt = (n + kTileSpheres - 1) / kTileSpheres;
            left = (nt / 2) * kTileSpheres;
        } else {
            left = ((n / 2 + 63) / 64) * 64;
        }
        auto less = [&](int x, int y) {
            const float4 &p = sph[(size_t)x], &q = sph[(size_t)y];
            const float u = axis == 0 ? p.x : (axis == 1 ? p.y : p.z), v = axis == 0 ? q.x : (axis == 1 ? q.y : q.z);
            return u < v || (u == v && x < y);
        };
        std::nth_element(ids.begin() + begin, ids.begin() + begin + left, ids.begin() + end, less);
        split(begin, begin + left);
        split(begin + left, end);
    }
};
// The staged matrix kind at half width (rt_hit.hpp filter_round_narrow): a box around the scene's ordinary spheres, such
// that a ray that cannot touch it has no candidate among them, and the list of the others -- spheres much larger than
// the ordinary ones, which would blow the box up, and spheres the lifted form cannot hold (candidates for every ray).
//   ordinary: lifted, finite, radius <= 4 x the median radius.  The others may fill bits of at most kWideWords words
//     and number at most 8 (each costs every ray a turn of the vector filter), else the scene does without.
//   origins: the box test speaks for rays whose origin lies within O = 4 x the largest |c - T| + r of an ordinary sphere
//     of the frame's centre T (or the lifted form's own limit if that is smaller); rays from farther away count as
//     touching.  D = O + max |c - T| then bounds |o - c|.
//   padding: whenever the reference's discriminant is >= 0 the ray's line passes within sqrt(r^2 + 17u D^2) of the
//     centre, u = 2^-24, and an accepted root's point lies within sqrt(r^2 + 34u D^2); every radius is grown to
//     sqrt(r^2 + 64u D^2) (rt_selftest_box finds its first wrongly left-out rays at 1u D^2 and below).  The test's own roundings: its cross products err by <= 4u E (|d_a| + |d_b|) with
//     E >= |o - mid| per axis, which 2^-16 E added to every half extent covers sixteen times over; the products with
//     the half extents err by 4u relative, covered by the factor 1 + 2^-12.
NarrowSet narrow_setup(const std::vector<float4> &sph, int nsph, const std::vector<char> &lifted_ok, const LiftFrame &frame, float oo_limit_m,
                       double padding = 1.0) {  // padding: scales every allowance (the self-test probes the margin with less)
    NarrowSet N;
    if (nsph < 128 || oo_limit_m <= 0.0f || (int)lifted_ok.size() < nsph) return N;
    std::vector<float> radii;
    for (int i = 0; i < nsph; ++i)
        if (lifted_ok[(size_t)i]) radii.push_back(sph[(size_t)i].w);
    if (radii.size() < (size_t)nsph / 2) return N;
    std::nth_element(radii.begin(), radii.begin() + (long)(radii.size() / 2), radii.end());
    const double rmed = radii[radii.size() / 2];
    const double T[3] = {frame.tx, frame.ty, frame.tz};
    auto ordinary = [&](int i) {
        const float4 &q = sph[(size_t)i];
        return lifted_ok[(size_t)i] && std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z) && std::isfinite(q.w) &&
               (double)q.w <= 4.0 * rmed;
    };
    int others = 0;
    double cmax = 0.0, reach = 0.0;
    for (int i = 0; i < nsph; ++i) {
        const float4 &q = sph[(size_t)i];
        if (!ordinary(i)) {
            const uint32_t w = (uint32_t)i / 32u, bit = 0x80000000u >> ((uint32_t)i & 31u);
            int k = 0;
            while (k < N.words && N.word[k] != w) ++k;
            if (k == kWideWords || ++others > 8) return N;
            if (k == N.words) N.word[N.words++] = w;
            // a sphere that can never be hit (no finite record) needs no look at all
            const bool finite = std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z) && std::isfinite(q.w);
            if (finite) (lifted_ok[(size_t)i] ? N.test[k] : N.always[k]) |= bit;
            continue;
        }
        const double dx = (double)q.x - T[0], dy = (double)q.y - T[1], dz = (double)q.z - T[2];
        const double c = std::sqrt(dx * dx + dy * dy + dz * dz);
        cmax = std::max(cmax, c);
        reach = std::max(reach, c + std::fabs((double)q.w));
    }
    const double o_lifted = std::sqrt((double)oo_limit_m) / (double)frame.sigma;
    const double O = std::min(4.0 * reach, o_lifted);
    const double D = O + cmax;
    const double grow = padding * 64.0 * 0x1p-24 * D * D;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < nsph; ++i) {
        if (!ordinary(i)) continue;
        const float4 &q = sph[(size_t)i];
        const double r = std::sqrt((double)q.w * (double)q.w + grow);
        const double c[3] = {q.x, q.y, q.z};
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], c[a] - r);
            hi[a] = std::max(hi[a], c[a] + r);
        }
    }
    float mid[3], half[3];
    for (int a = 0; a < 3; ++a) {
        mid[a] = (float)(0.5 * (lo[a] + hi[a]));
        const double E = O + std::fabs((double)mid[a] - T[a]);
        const double h = (std::max(hi[a] - (double)mid[a], (double)mid[a] - lo[a])) * (1.0 + padding * 0x1p-12) + padding * 0x1p-16 * E;
        half[a] = std::nextafterf((float)h, std::numeric_limits<float>::infinity());
        if (!std::isfinite(mid[a]) || !std::isfinite(half[a])) return N;
    }
    N.mid.x = mid[0]; N.mid.y = mid[1]; N.mid.z = mid[2];
    N.half.x = half[0]; N.half.y = half[1]; N.half.z = half[2];
    const double oo = O * (double)frame.sigma * (O * (double)frame.sigma) * (1.0 - 0x1p-10);
    N.oo = (float)std::min(oo, (double)oo_limit_m);
    N.ok = true;
    return N;
}

// Matrix-core filter (rt_lift.hpp) of the staged kind: frame = median centre and a power-of-two scale that puts the 95th
// percentile of |c - T|_inf + r near 42 (of the 244 that the FP16 range allows for a ray origin), lifted
// records in A-operand order, and the origin limit: the hard range limit, or less when 10 % of the spheres
// would see their filter radius^2 doubled by the slack.
struct LiftSet {
    std::vector<_Float16> sphm;
    LiftFrame frame = {0.0f, 0.0f, 0.0f, 1.0f};
    float oo_limit_m = -1.0f;
    int outliers = 0;
    std::vector<char> lifted_ok;  // per sphere: the lifted form holds it (else it is a candidate for every ray)
};
LiftSet lift_setup(const std::vector<float4> &sph, int nsph) {
    LiftSet L;
    std::vector<_Float16> &sphm = L.sphm;
    LiftFrame &frame = L.frame;
    float &oo_limit_m = L.oo_limit_m;
    int &outliers = L.outliers;
    std::vector<char> &lifted_ok = L.lifted_ok;
    if (nsph > 0) {
        std::vector<float> med[3];
        for (int i = 0; i < nsph; ++i) {
            med[0].push_back(sph[(size_t)i].x); med[1].push_back(sph[(size_t)i].y); med[2].push_back(sph[(size_t)i].z);
        }
        float t[3];
        for (int a = 0; a < 3; ++a) {
            std::nth_element(med[a].begin(), med[a].begin() + nsph / 2, med[a].end());
            t[a] = med[a][(size_t)(nsph / 2)];
            if (!std::isfinite(t[a])) t[a] = 0.0f;
        }
        frame.tx = t[0]; frame.ty = t[1]; frame.tz = t[2];
        std::vector<double> ext;
        for (int i = 0; i < nsph; ++i) {
            const float4 &q = sph[(size_t)i];
            double e = std::max(std::fabs((double)q.x - t[0]), std::max(std::fabs((double)q.y - t[1]), std::fabs((double)q.z - t[2]))) +
                       std::fabs((double)q.w);
            ext.push_back(std::isfinite(e) ? e : 0.0);
        }
        size_t k95 = (size_t)((double)(nsph - 1) * 0.95);
        std::nth_element(ext.begin(), ext.begin() + (long)k95, ext.end());
        double e95 = ext[k95];
        int ex = (e95 > 0.0 && std::isfinite(e95)) ? (int)std::floor(std::log2(42.5 / e95)) : 0;
        ex = std::max(-60, std::min(60, ex));
        frame.sigma = std::ldexp(1.0f, ex);
        const int nblocks = (nsph + 63) / 64;
        sphm.assign((size_t)nblocks * 64 * kLiftSlots, (_Float16)0.0f);
        std::vector<double> room;
        lifted_ok.assign((size_t)nsph, 0);
        for (int i = 0; i < nblocks * 64; ++i) {
            _Float16 slots[kLiftSlots];
            if (i < nsph) {
                const float4 &q = sph[(size_t)i];
                if (lift_sphere(q.x, q.y, q.z, q.w, frame, (double)kMfmaSlack, slots)) {
                    lifted_ok[(size_t)i] = 1;
                    const double g = frame.sigma;
                    const double x = ((double)q.x - t[0]) * g, y = ((double)q.y - t[1]) * g, z = ((double)q.z - t[2]) * g, r2 = (double)q.w * g * ((double)q.w * g);
                    room.push_back(r2 / (double)kMfmaSlack - (x * x + y * y + z * z) - r2);
                } else {
                    ++outliers;
                }
            } else {
                lift_constant(false, slots);
            }
            for (int k = 0; k < kLiftSlots; ++k) sphm[lift_index(i / 64, i % 64, k)] = slots[k];
        }
        if (!room.empty() && outliers <= std::max(2, nsph / 16)) {
            size_t k = room.size() / 10;
            std::nth_element(room.begin(), room.begin() + (long)k, room.end());
            double lim = std::min(room[k], 57000.0);
            if (lim > 0.0) oo_limit_m = (float)lim;
        }
    }
    return L;
}

TileSet build_tiles(const std::vector<float4> &sph, const std::vector<int> &sph_obj, int nsph) {
    TileSet T;
    if (nsph <= 0) return T;
    std::vector<float> radii;
    for (int i = 0; i < nsph; ++i)
        if (std::isfinite(sph[(size_t)i].w) && sph[(size_t)i].w > 0.0f) radii.push_back(sph[(size_t)i].w);
    float rmed = 0.0f;
    if (!radii.empty()) {
        std::nth_element(radii.begin(), radii.begin() + (long)(radii.size() / 2), radii.end());
        rmed = radii[radii.size() / 2];
    }
    auto regular = [&](int i) {
        const float4 &q = sph[(size_t)i];
        return std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z) && std::isfinite(q.w) && q.w > 0.0f && q.w <= 64.0f * rmed;
    };
    TileBuilder B{sph, {}, {}};
    std::vector<int> irregular;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, rmax = 0.0;
    for (int i = 0; i < nsph; ++i) {
        if (!regular(i)) {
            irregular.push_back(i);
            continue;
        }
        B.ids.push_back(i);
        const double c[3] = {sph[(size_t)i].x, sph[(size_t)i].y, sph[(size_t)i].z};
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
        rmax = std::max(rmax, (double)sph[(size_t)i].w);
    }
    if (!B.ids.empty()) B.split(0, (int)B.ids.size());
    // tile contents in slot order.  Inside a block of 64 slots: descending object index -- the tie rule of the exact
    // pass, not the visiting order, is what gives an equal distance to the lower index, and this keeps that rule exercised ...
    std::vector<std::vector<int>> members;
    // ... and before that, blocks of 64 slots whose spheres are neighbours in space (each block gets a box of its own
    // below): median splits along the widest axis of the centres, whole blocks to the left.
    std::function<void(std::vector<int> &, int, int)> order_blocks = [&](std::vector<int> &m, int lo_i, int hi_i) {
        const int nb = (hi_i - lo_i + 63) / 64;
        if (nb <= 1) {
            std::sort(m.begin() + lo_i, m.begin() + hi_i, std::greater<int>());
            return;
        }
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        for (int k = lo_i; k < hi_i; ++k) {
            const float4 &q = sph[(size_t)m[(size_t)k]];
            const double c[3] = {q.x, q.y, q.z};
            for (int a = 0; a < 3; ++a) { blo[a] = std::min(blo[a], c[a]); bhi[a] = std::max(bhi[a], c[a]); }
        }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (bhi[a] - blo[a] > bhi[axis] - blo[axis]) axis = a;
        const int mid = lo_i + ((nb + 1) / 2) * 64;
        auto key = [&](int i) { const float4 &q = sph[(size_t)i]; return axis == 0 ? q.x : axis == 1 ? q.y : q.z; };
        std::nth_element(m.begin() + lo_i, m.begin() + mid, m.begin() + hi_i,
                         [&](int p, int q) { return key(p) < key(q) || (key(p) == key(q) && p < q); });
        order_blocks(m, lo_i, mid);
        order_blocks(m, mid, hi_i);
    };
    for (auto &lf : B.leaves) {
        std::vector<int> m(B.ids.begin() + lf.first, B.ids.begin() + lf.second);   // regular spheres only: finite centres
        order_blocks(m, 0, (int)m.size());
        members.push_back(m);
    }
    // spheres no tile's frame fits (a ground sphere; anything not finite) get tiles of their own: in a tile of ordinary
    // spheres one of them would make the tile's box infinite, and every ray would be put through that tile's filter
    bool own_tile = false;
    for (int i : irregular) {
        if (!own_tile || (int)members.back().size() >= kTileSpheres) members.push_back({});
        own_tile = true;
        members.back().push_back(i);
    }
    double diag = 0.0;
    if (!B.ids.empty()) {
        for (int a = 0; a < 3; ++a) diag += (hi[a] - lo[a]) * (hi[a] - lo[a]);
        diag = std::sqrt(diag) + 2.0 * rmax;
    }
    int ex = (diag > 0.0 && std::isfinite(diag)) ? (int)std::floor(std::log2(238.0 / (1.5 * diag))) : 0;
    ex = std::max(-60, std::min(60, ex));
    const float sigma = std::ldexp(1.0f, ex);
    T.ntiles = (int)members.size();
    T.tiles.assign((size_t)T.ntiles * kTileVec4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    T.frames.assign((size_t)T.ntiles * kTileFrameVec4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    T.obj.assign((size_t)T.ntiles * kTileSpheres, -1);
    T.rank.assign((size_t)T.ntiles * 512, 0);   // eight places per lane whatever the tile size
    int outliers = 0;
    std::vector<float4> block_boxes;   // appended to T.frames after the groups' boxes
    for (int t = 0; t < T.ntiles; ++t) {
        const std::vector<int> &m = members[(size_t)t];
        const int count = (int)m.size();
        double tlo[3] = {1e300, 1e300, 1e300}, thi[3] = {-1e300, -1e300, -1e300};
        bool any = false;
        for (int i : m) {
            if (!regular(i)) continue;
            const double c[3] = {sph[(size_t)i].x, sph[(size_t)i].y, sph[(size_t)i].z};
            for (int a = 0; a < 3; ++a) { tlo[a] = std::min(tlo[a], c[a]); thi[a] = std::max(thi[a], c[a]); }
            any = true;
        }
        LiftFrame F = {0.0f, 0.0f, 0.0f, sigma};
        if (any) { F.tx = (float)(0.5 * (tlo[0] + thi[0])); F.ty = (float)(0.5 * (tlo[1] + thi[1])); F.tz = (float)(0.5 * (tlo[2] + thi[2])); }
        _Float16 *lifted = (_Float16 *)&T.tiles[(size_t)t * kTileVec4];
        float4 *strict = &T.tiles[(size_t)t * kTileVec4 + kTileLifted];
        double cmax = 0.0;  // largest |c'| among the spheres of the tile that are filtered (lift_ray_near)
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        bool boxed = true;
        double klo[kTileBlocks][3], khi[kTileBlocks][3];   // the same per block of 64 slots
        bool kboxed[kTileBlocks];
        for (int k = 0; k < kTileBlocks; ++k) {
            kboxed[k] = true;
            for (int a = 0; a < 3; ++a) { klo[k][a] = 1e300; khi[k][a] = -1e300; }
        }
        {
            // the place of every slot's object index among the tile's, ascending
            std::vector<std::pair<int, int>> by_obj;
            for (int j = 0; j < count; ++j) by_obj.push_back({sph_obj[(size_t)m[(size_t)j]], j});
            std::sort(by_obj.begin(), by_obj.end());
            for (int k = 0; k < count; ++k) {
                const int j = by_obj[(size_t)k].second;
                T.rank[(size_t)t * 512 + (size_t)(j % 64) * 8 + (size_t)(j / 64)] = (uint16_t)k;
            }
        }
        for (int j = 0; j < kTileSpheres; ++j) {
            _Float16 slots[kLiftSlots];
            if (j < count) {
                const int i = m[(size_t)j];
                const float4 &q = sph[(size_t)i];
                strict[j] = q;
                T.obj[(size_t)t * kTileSpheres + j] = sph_obj[(size_t)i];
                if (lift_sphere(q.x, q.y, q.z, q.w, F, (double)kMfmaSlack, slots)) {
                    const double g = F.sigma;
                    const double x = ((double)q.x - F.tx) * g, y = ((double)q.y - F.ty) * g, z = ((double)q.z - F.tz) * g;
                    cmax = std::max(cmax, std::sqrt(x * x + y * y + z * z));
                    const double c[3] = {q.x, q.y, q.z};
                    for (int a = 0; a < 3; ++a) {
                        blo[a] = std::min(blo[a], c[a] - std::fabs((double)q.w));
                        bhi[a] = std::max(bhi[a], c[a] + std::fabs((double)q.w));
                        klo[j / 64][a] = std::min(klo[j / 64][a], c[a] - std::fabs((double)q.w));
                        khi[j / 64][a] = std::max(khi[j / 64][a], c[a] + std::fabs((double)q.w));
                    }
                } else {
                    ++outliers;
                    boxed = false;
                    kboxed[j / 64] = false;
                }
            } else {
                strict[j] = make_float4(0.0f, 0.0f, 0.0f, std::numeric_limits<float>::quiet_NaN());  // never a hit
                lift_constant(false, slots);
            }
            for (int k = 0; k < kLiftSlots; ++k) lifted[lift_index(j / 64, j % 64, k)] = slots[k];
        }
        // Range limit in the tile's frame: the hard limit of the lifted form.  The slack that grows with distance only
        // adds candidates; it is no reason to leave the matrix cores.
        T.frames[(size_t)kTileFrameVec4 * t] = make_float4(F.tx, F.ty, F.tz, F.sigma);
        T.frames[(size_t)kTileFrameVec4 * t + 1] = make_float4(57000.0f, (float)count, std::nextafterf((float)cmax, 1e30f), 0.0f);
        // The box around the tile's spheres (closest_sphere_tiled leaves out rays that cannot touch it; the allowances
        // that depend on the ray's distance are added there).  Only spheres the lifted form holds can be left to a box:
        // any other member (never hit, or a candidate for every ray) makes the box infinite.
        const float inf = std::numeric_limits<float>::infinity();
        auto box_of = [&](const double *lo3, const double *hi3, bool ok, float4 &bmid, float4 &bhalf) {
            bmid = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            bhalf = make_float4(inf, inf, inf, 0.0f);
            if (!ok) return;
            float mid[3], half[3];
            double diag2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                mid[a] = (float)(0.5 * (lo3[a] + hi3[a]));
                const double h = std::max(hi3[a] - (double)mid[a], (double)mid[a] - lo3[a]);
                half[a] = std::nextafterf((float)(h * (1.0 + 0x1p-12)), inf);
                diag2 += (double)half[a] * (double)half[a];
            }
            if (std::isfinite(mid[0] + mid[1] + mid[2]) && std::isfinite(half[0] + half[1] + half[2])) {
                bmid = make_float4(mid[0], mid[1], mid[2], std::nextafterf((float)std::sqrt(diag2), inf));
                bhalf = make_float4(half[0], half[1], half[2], 0.0f);
            }
        };
        box_of(blo, bhi, boxed && count > 0, T.frames[(size_t)kTileFrameVec4 * t + 2], T.frames[(size_t)kTileFrameVec4 * t + 3]);
        // ... and around the spheres of each of its blocks of 64 slots (a block without spheres is never looked at)
        for (int k = 0; k < kTileBlocks; ++k) {
            float4 bm, bh;
            box_of(klo[k], khi[k], kboxed[k] && 64 * k < count, bm, bh);
            block_boxes.push_back(bm);
            block_boxes.push_back(bh);
        }
    }
    // a box around every group of kTileGroup consecutive tiles (the leaves of the median splits in order: neighbours in space)
    for (int t0 = 0; t0 < T.ntiles; t0 += kTileGroup) {
        const float inf = std::numeric_limits<float>::infinity();
        double lo3[3] = {1e300, 1e300, 1e300}, hi3[3] = {-1e300, -1e300, -1e300};
        bool boxed = true;
        for (int t = t0; t < std::min(T.ntiles, t0 + kTileGroup); ++t) {
            const float4 m = T.frames[(size_t)kTileFrameVec4 * t + 2], h = T.frames[(size_t)kTileFrameVec4 * t + 3];
            const double mm[3] = {m.x, m.y, m.z}, hh[3] = {h.x, h.y, h.z};
            for (int a = 0; a < 3; ++a) {
                if (!std::isfinite(hh[a])) boxed = false;
                lo3[a] = std::min(lo3[a], mm[a] - hh[a]);
                hi3[a] = std::max(hi3[a], mm[a] + hh[a]);
            }
        }
        float4 bmid = make_float4(0.0f, 0.0f, 0.0f, 0.0f), bhalf = make_float4(inf, inf, inf, 0.0f);
        if (boxed) {
            float mid[3], half[3];
            double diag2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                mid[a] = (float)(0.5 * (lo3[a] + hi3[a]));
                half[a] = std::nextafterf((float)std::max(hi3[a] - (double)mid[a], (double)mid[a] - lo3[a]), inf);
                diag2 += (double)half[a] * (double)half[a];
            }
            if (std::isfinite(mid[0] + mid[1] + mid[2]) && std::isfinite(half[0] + half[1] + half[2])) {
                bmid = make_float4(mid[0], mid[1], mid[2], std::nextafterf((float)std::sqrt(diag2), inf));
                bhalf = make_float4(half[0], half[1], half[2], 0.0f);
            }
        }
        T.frames.push_back(bmid);
        T.frames.push_back(bhalf);
    }
    T.frames.insert(T.frames.end(), block_boxes.begin(), block_boxes.end());
    T.ok = outliers <= std::max(2, nsph / 16);  // else pass B would do most of the work
    return T;
}

// Object[] (160-byte AoS records) -> compact device arrays.  Sphere slots keep the objects' relative order;
// the slot count is padded to a multiple of 32 with records that can never be hit.
// Do lines through the scene cross many spheres?  128 lines through pseudo-random points of the box of the sphere
// centres, in pseudo-random directions, are tested against every sphere (the sphere filter tests lines, not rays); the
// scene counts as dense when they cross kDenseMeanSpheres or more on average: nested glass shells about 170, the cover
// scene 3.  Such scenes get the kernels that carry the pruning, lane-by-lane form of the exact pass (rt_hit.hpp) when
// their bounding balls can do something for it (balls_nested).
constexpr double kDenseMeanSpheres = 24.0;
bool dense_scene(const std::vector<float4> &sph, int nsph) {  // the first nsph records (the rest is padding)
    const size_t n = (size_t)nsph;
    if (n < 32) return false;
    // box of the centres without the outermost 2 % per axis (a ground sphere of radius 1000 must not set the scale)
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        std::vector<float> v;
        for (size_t i = 0; i < n; ++i) {
            const float c = a == 0 ? sph[i].x : a == 1 ? sph[i].y : sph[i].z;
            if (std::isfinite(c)) v.push_back(c);  // (a value that is not a number has no place in an order)
        }
        if (v.empty()) return false;
        std::sort(v.begin(), v.end());
        lo[a] = v[v.size() / 50];
        hi[a] = v[v.size() - 1 - v.size() / 50];
    }
    uint32_t st = 0x9e3779b9u;
    auto rnd = [&]() -> double {  // [0, 1)
        st = st * 747796405u + 2891336453u;
        uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
        return (double)((w >> 22u) ^ w) * (1.0 / 4294967296.0);
    };
    const int lines = 128;
    size_t crossed = 0;
    for (int l = 0; l < lines; ++l) {
        double o[3], d[3], dd;
        for (int a = 0; a < 3; ++a) o[a] = lo[a] + (hi[a] - lo[a]) * rnd();
        do {
            for (int a = 0; a < 3; ++a) d[a] = 2.0 * rnd() - 1.0;
            dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        } while (dd > 1.0 || dd < 1e-4);
        for (size_t i = 0; i < n; ++i) {
            const double cx = o[0] - sph[i].x, cy = o[1] - sph[i].y, cz = o[2] - sph[i].z, r = sph[i].w;
            const double b = cx * d[0] + cy * d[1] + cz * d[2];
            if (b * b - dd * (cx * cx + cy * cy + cz * cz - r * r) >= 0.0) ++crossed;  // NaN compares false
        }
    }
    return (double)crossed >= kDenseMeanSpheres * lines;
}

// Bounding balls of consecutive sphere slots, for the kernels that prune candidates a group at a time (rt_hit.hpp
// cull_words, cull_granules).  Per group -- a word of 32 slots, and each of its four granules of 8 -- two balls about
// one centre C (the component-wise median of the group's centres, rounded to float): an outer ball (C, Rout) that
// holds every sphere of the group and an inner ball (C, Rin) that every sphere of the group holds.  Spheres that
// would blow the outer ball up (|c - C| + r above four times the group's median of that number: a ground sphere
// among small ones), or that do not hold C where three quarters of the group do (a small sphere among shells: it
// would cost the group its inner ball), are left out of both balls and listed in a mask of exempt slots, which a
// prune of the group leaves standing.  Stored squared and with the margins the kernel's argument needs: R2out = Rout^2 (1 + 2^-6)
// rounded up, R2in = Rin^2 (1 - 2^-6) rounded down, and no inner ball (R2in = -1) unless Rin > 0 and the largest radius
// of the group is at most 64 Rin.  A group with a sphere that is not finite, or without spheres, gets R2out = +inf
// and R2in = -1: never pruned.  Layout: kBoundVec4 float4 per word: (C, R2out), (R2in, exempt mask bits, 1 if every
// sphere of the group but the exempt ones has exactly the centre C else 0, 0) of the
// word, then the same pair for each granule (mask bit of slot p of the word: 0x80000000 >> p).
void group_bounds(const std::vector<float4> &sph, int i0, int i1, int word0, float4 *out) {
    const float inf = std::numeric_limits<float>::infinity();
    out[0] = make_float4(0.0f, 0.0f, 0.0f, inf);
    out[1] = make_float4(-1.0f, 0.0f, 0.0f, 0.0f);
    if (i1 <= i0) return;
    const int n = i1 - i0;
    std::vector<float> v[3];
    for (int i = i0; i < i1; ++i) {
        const float4 &q = sph[(size_t)i];
        if (!(std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z) && std::isfinite(q.w))) return;
        v[0].push_back(q.x); v[1].push_back(q.y); v[2].push_back(q.z);
    }
    float c[3];
    for (int a = 0; a < 3; ++a) {
        std::nth_element(v[a].begin(), v[a].begin() + n / 2, v[a].end());
        c[a] = v[a][(size_t)(n / 2)];
    }
    std::vector<double> dist((size_t)n), reach((size_t)n);
    for (int i = i0; i < i1; ++i) {
        const float4 &q = sph[(size_t)i];
        const double dx = (double)q.x - c[0], dy = (double)q.y - c[1], dz = (double)q.z - c[2];
        dist[(size_t)(i - i0)] = std::sqrt(dx * dx + dy * dy + dz * dz) * (1.0 + 0x1p-40);
        reach[(size_t)(i - i0)] = dist[(size_t)(i - i0)] + std::fabs((double)q.w);
    }
    std::vector<double> sorted = reach;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const double limit = 4.0 * sorted[(size_t)(n / 2)];
    // when three quarters of the group hold the centre, the others (a small sphere among shells) stay out of the balls too
    int holding = 0;
    for (int i = i0; i < i1; ++i) holding += std::fabs((double)sph[(size_t)i].w) > dist[(size_t)(i - i0)] ? 1 : 0;
    const bool shells = 4 * holding >= 3 * n;
    double rout = 0.0, rin = std::numeric_limits<double>::infinity(), rmax = 0.0;
    uint32_t exempt = 0;
    for (int i = i0; i < i1; ++i) {
        const double r = std::fabs((double)sph[(size_t)i].w);
        if (reach[(size_t)(i - i0)] > limit || (shells && !(r > dist[(size_t)(i - i0)]))) {
            exempt |= 0x80000000u >> (i - word0);
            continue;
        }
        rout = std::max(rout, reach[(size_t)(i - i0)]);
        rin = std::min(rin, r - dist[(size_t)(i - i0)]);
        rmax = std::max(rmax, r);
    }
    const double r2o = rout * rout * (1.0 + 0x1p-6);
    float fo = (float)r2o;
    if ((double)fo < r2o) fo = std::nextafterf(fo, inf);
    out[0] = make_float4(c[0], c[1], c[2], fo);
    std::memcpy(&out[1].y, &exempt, sizeof exempt);
    // every sphere of the group that the balls hold has exactly this centre (concentric shells): the kernels then work
    // out the ray's numbers about it once for all of them (rt_hit.hpp prune_candidates)
    bool concentric = true;
    for (int i = i0; i < i1; ++i) concentric = concentric && ((exempt >> (31 - (i - word0))) & 1u || (sph[(size_t)i].x == c[0] && sph[(size_t)i].y == c[1] && sph[(size_t)i].z == c[2]));
    out[1].z = concentric ? 1.0f : 0.0f;
    if (rin > 0.0 && rmax <= 64.0 * rin) {
        const double r2i = rin * rin * (1.0 - 0x1p-6);
        float fi = (float)r2i;
        if ((double)fi > r2i) fi = std::nextafterf(fi, -inf);
        out[1].x = fi;
    }
}
std::vector<float4> word_bounds(const std::vector<float4> &sph, int nsph) {
    const size_t nwords = (sph.size() + 31) / 32;
    std::vector<float4> out((size_t)kBoundVec4 * nwords);
    for (size_t w = 0; w < nwords; ++w) {
        const int i0 = (int)w * 32;
        float4 *rec = out.data() + (size_t)kBoundVec4 * w;
        group_bounds(sph, i0, std::min(nsph, i0 + 32), i0, rec);
        for (int g = 0; g < 4; ++g) group_bounds(sph, std::min(nsph, i0 + 8 * g), std::min(nsph, i0 + 8 * g + 8), i0, rec + 2 + 2 * g);
    }
    // Runs: eight or more consecutive slots of finite spheres of positive radius that no word lists as exempt, with
    // exactly one centre, whose radii never rise (or never fall) from slot to slot -- shells in the order of their
    // size.  A word all of whose spheres but the exempt ones lie in one run, about the centre its balls are about, gets
    // the run in its record: third number of the second float4 = 2 (radii fall) or 3 (rise) instead of 1, fourth = the
    // bits of first slot | (last slot + 1) << 16.  The kernels then find the candidates of such a word that can win by
    // two binary searches over the run's radii instead of testing them one by one (rt_hit.hpp run_survivors).
    auto clean = [&](int k) -> bool {
        const float4 &q = sph[(size_t)k];
        const uint32_t exempt = [&] { uint32_t e; std::memcpy(&e, &out[(size_t)kBoundVec4 * (size_t)(k / 32) + 1].y, sizeof e); return e; }();
        return std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z) && std::isfinite(q.w) && q.w > 0.0f && !((exempt >> (31 - (k & 31))) & 1u);
    };
    std::vector<int> run_of((size_t)std::max(nsph, 0), -1);
    struct Run { int b, e, dir; };
    std::vector<Run> runs;
    for (int i = 0; i < nsph && i < 65535;) {
        if (!clean(i)) {
            ++i;
            continue;
        }
        int e = i + 1, dir = 0;
        while (e < nsph && e < 65535 && clean(e) && sph[(size_t)e].x == sph[(size_t)i].x && sph[(size_t)e].y == sph[(size_t)i].y &&
               sph[(size_t)e].z == sph[(size_t)i].z) {
            const float a = sph[(size_t)e - 1].w, b = sph[(size_t)e].w;
            const int step = b < a ? -1 : b > a ? 1 : 0;
            if (step != 0 && dir != 0 && step != dir) break;
            if (step != 0) dir = step;
            ++e;
        }
        if (e - i >= 8) {
            for (int k = i; k < e; ++k) run_of[(size_t)k] = (int)runs.size();
            runs.push_back({i, e, dir});
        }
        i = e;
    }
    for (size_t w = 0; w < nwords; ++w) {
        float4 *rec = out.data() + (size_t)kBoundVec4 * w;
        if (rec[1].z != 1.0f) continue;  // not about one centre
        uint32_t exempt;
        std::memcpy(&exempt, &rec[1].y, sizeof exempt);
        int run = -2;
        for (int k = (int)w * 32; k < std::min(nsph, (int)w * 32 + 32); ++k) {
            if ((exempt >> (31 - (k & 31))) & 1u) continue;
            if (run == -2) run = run_of[(size_t)k];
            else if (run != run_of[(size_t)k]) run = -1;
        }
        if (run < 0) continue;
        const Run &r = runs[(size_t)run];
        const float4 &first = sph[(size_t)r.b];
        if (!(first.x == rec[0].x && first.y == rec[0].y && first.z == rec[0].z)) continue;
        rec[1].z = r.dir <= 0 ? 2.0f : 3.0f;
        const uint32_t bits = (uint32_t)r.b | ((uint32_t)r.e << 16);
        std::memcpy(&rec[1].w, &bits, sizeof bits);
    }
    return out;
}

// Per resolve round of 512 slots (16 words): the run its words belong to, if all its words of runs belong to one
// (TraceParams::round_runs).
std::vector<float4> round_runs(const std::vector<float4> &balls) {
    const size_t nwords = balls.size() / (size_t)kBoundVec4, nrounds = (nwords + 15) / 16;
    std::vector<float4> out(2 * std::max<size_t>(nrounds, 1), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (size_t r = 0; r < nrounds; ++r) {
        uint32_t words = 0, with_exempt = 0, bits = 0;
        float kind = 0.0f;
        const float4 *first = nullptr;
        bool one = true;
        for (size_t w = 16 * r; w < std::min(nwords, 16 * r + 16); ++w) {
            const float4 *rec = balls.data() + (size_t)kBoundVec4 * w;
            uint32_t exempt, b;
            std::memcpy(&exempt, &rec[1].y, sizeof exempt);
            std::memcpy(&b, &rec[1].w, sizeof b);
            if (exempt) with_exempt |= 1u << (w - 16 * r);
            if (rec[1].z < 2.0f) continue;
            if (!first) {
                first = rec;
                bits = b;
                kind = rec[1].z;
            } else if (b != bits || rec[1].z != kind) {
                one = false;
            }
            words |= 1u << (w - 16 * r);
        }
        if (!first || !one) continue;
        const uint32_t masks = words | (with_exempt << 16);
        out[2 * r] = make_float4(first[0].x, first[0].y, first[0].z, 0.0f);
        std::memcpy(&out[2 * r].w, &bits, sizeof bits);
        out[2 * r + 1] = make_float4(0.0f, kind, 0.0f, 0.0f);
        std::memcpy(&out[2 * r + 1].x, &masks, sizeof masks);
    }
    return out;
}

// Do the bounding balls have something to strike with?  What the kernels for dense scenes gain, they gain from
// groups of spheres that share an interior (shells within shells: one test strikes every sphere around a ray's
// segment); among spheres that merely overlap without order (measured on 300 of them: 405 Msamples*px/s pooled, 364-388
// with the pruning kernels) the balls strike nothing.  True when at least half of the granules of 8 slots that hold
// finite spheres have an inner ball.
bool balls_nested(const std::vector<float4> &balls) {
    size_t groups = 0, nested = 0;
    for (size_t w = 0; w + (size_t)kBoundVec4 <= balls.size(); w += (size_t)kBoundVec4)
        for (int g = 0; g < 4; ++g) {
            const float4 &outer = balls[w + 2 + 2 * (size_t)g], &inner = balls[w + 3 + 2 * (size_t)g];
            if (!std::isfinite(outer.w)) continue;  // empty, or a sphere that is not finite
            ++groups;
            nested += inner.x > 0.0f ? 1 : 0;
        }
    return groups > 0 && 2 * nested >= groups;
}

// cache_hit (optional): set when the bytes equal the last upload and the device arrays were kept as they are.
int upload_scene(rt_ctx *ctx, const Object *objs, int n, bool *cache_hit = nullptr) {
    if (cache_hit) *cache_hit = false;
    if (n < 0 || (n > 0 && !objs)) return fail(ctx, RT_ERR_INVALID, "rt: bad object array");
    // The reference re-sends its objects with every dispatch (ContentView.swift:689-696), and so may callers of
    // rt_render_tile: 510 times per pass over a 1080p target.  The bytes are compared with those the device arrays
    // were built from, and the compaction, lifting, tiling and uploads below are skipped when nothing changed.
    const size_t nbytes = (size_t)n * sizeof(Object);
    if (ctx->uploaded_valid && ctx->uploaded.size() == nbytes && (nbytes == 0 || std::memcmp(ctx->uploaded.data(), objs, nbytes) == 0)) {
        if (cache_hit) *cache_hit = true;
        return RT_OK;
    }
    ctx->uploaded_valid = false;
    std::vector<float4> sph;
    std::vector<int> sph_obj;
    std::vector<QuadDev> quads;
    std::vector<MatDev> mats((size_t)n);
    for (int i = 0; i < n; ++i) {
        const Object &o = objs[i];
        MatDev m;
        m.albedo = ld4(o.mat.albedo);
        if (o.mat.emission > 0.0f) {  // Shaders.metal:51-54
            m.emis.x = o.mat.emission_color.x * o.mat.emission;
            m.emis.y = o.mat.emission_color.y * o.mat.emission;
            m.emis.z = o.mat.emission_color.z * o.mat.emission;
            m.emis.w = o.mat.emission_color.w * o.mat.emission;
        } else {
            m.emis.x = 0.0f; m.emis.y = 0.0f; m.emis.z = 0.0f; m.emis.w = 1.0f;
        }
        m.type = o.mat.type;
        m.fuzz = o.mat.reflection_fuzz;
        m.ior = o.mat.refraction_index;
        m.inv_ior = 1.0f / o.mat.refraction_index;  // Lib.metal:185
        if (m.type == DIELECTRIC) {
            // Lib.metal:178-179 for the two values refraction_index takes there (Lib.metal:185); a dielectric's albedo is
            // never read (its attenuation is 1, Lib.metal:184), so its slots carry them
            volatile float rf = (1.0f - m.inv_ior) / (1.0f + m.inv_ior), rb = (1.0f - m.ior) / (1.0f + m.ior);
            m.albedo.x = rf * rf;
            m.albedo.y = rb * rb;
        }
        mats[(size_t)i] = m;
        if (o.type == TYPE_SPHERE) {
            sph.push_back(make_float4(o.s.center.x, o.s.center.y, o.s.center.z, o.s.radius));  // strict record
            sph_obj.push_back(i);
        } else if (o.type == TYPE_QUAD) {
            QuadDev q;
            q.origin = ld3(o.q.origin); q.obj = i;
            q.ex = ld3(o.q.extentX); q.pad0 = 0.0f;
            q.ey = ld3(o.q.extentY); q.pad1 = 0.0f;
            quads.push_back(q);
        }  // any other type never hits (Lib.metal:283-291)
    }
    const int nsph = (int)sph.size();
    const int ngroups = (nsph + 3) / 4;
    // read-ahead room: the staged/scalar fetchers read one group past the end, the streaming fetcher two tiles
    const int nslots = (4 * ngroups + 127) / 128 * 128 + 256;
    // Compact filter records and the origin limit below which a ray may use them (rt_trace.hip, closest_sphere).
    std::vector<float4> sphc;
    std::vector<double> budget;
    const double slack = 0x1p-16;
    for (int i = 0; i < nsph; ++i) {
        const Object &o = objs[sph_obj[(size_t)i]];
        const double cx = o.s.center.x, cy = o.s.center.y, cz = o.s.center.z;
        const double r2 = (double)(o.s.radius * o.s.radius);
        const double cc = cx * cx + cy * cy + cz * cz;
        const double kq = cc * (1.0 - slack) - r2 * (1.0 + slack);
        float f = (float)kq;
        if ((double)f > kq) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());  // round toward -inf
        sphc.push_back(make_float4(o.s.center.x, o.s.center.y, o.s.center.z, f));
        budget.push_back(0.25 * r2 / slack - cc);  // |o|^2 at which this sphere's filter radius^2 has grown by 25 %
    }
    float oo_limit = -1.0f;  // no sphere: the general filter (nothing to test either way)
    if (nsph > 0) {
        size_t k = (size_t)(nsph / 10);
        std::nth_element(budget.begin(), budget.begin() + (long)k, budget.end());
        double lim = budget[k];  // 10th percentile: 90 % of the spheres keep their inflation below 25 %
        oo_limit = lim > 0.0 ? (float)std::min(lim, 1e30) : -1.0f;
    }
    // matrix-core filter of the staged kind: frame, lifted records, origin limit
    LiftSet lifted_set = lift_setup(sph, nsph);
    std::vector<_Float16> &sphm = lifted_set.sphm;
    const LiftFrame frame = lifted_set.frame;
    const float oo_limit_m = lifted_set.oo_limit_m;
    const int outliers = lifted_set.outliers;
    const std::vector<char> &lifted_ok = lifted_set.lifted_ok;
    const NarrowSet narrow = narrow_setup(sph, nsph, lifted_ok, frame, oo_limit_m);
    TileSet tileset = build_tiles(sph, sph_obj, nsph);
    while ((int)sph.size() < nslots) {
        sph.push_back(make_float4(0.0f, 0.0f, 0.0f, std::numeric_limits<float>::quiet_NaN()));    // never a hit
        sphc.push_back(make_float4(0.0f, 0.0f, 0.0f, std::numeric_limits<float>::infinity()));    // q = -inf
        sph_obj.push_back(-1);
    }
    RT_TRY(upload(ctx, ctx->sph, sph));
    RT_TRY(upload(ctx, ctx->sphc, sphc));
    RT_TRY(upload(ctx, ctx->sphm, sphm));
    RT_TRY(upload(ctx, ctx->tiles, tileset.tiles));
    RT_TRY(upload(ctx, ctx->tile_frames, tileset.frames));
    const std::vector<float4> balls = word_bounds(sph, nsph);
    RT_TRY(upload(ctx, ctx->word_bounds, balls));
    RT_TRY(upload(ctx, ctx->round_runs, round_runs(balls)));
    RT_TRY(upload(ctx, ctx->tile_obj, tileset.obj));
    RT_TRY(upload(ctx, ctx->tile_rank, tileset.rank));
    RT_TRY(upload(ctx, ctx->sph_obj, sph_obj));
    RT_TRY(upload(ctx, ctx->quads, quads));
    RT_TRY(upload(ctx, ctx->mats, mats));
    // the host vectors die at return: wait for the copies
    RT_HIP(hipStreamSynchronize(ctx->stream));
    ctx->nsph = nsph;
    ctx->ngroups = ngroups;
    ctx->oo_limit = oo_limit;
    ctx->lift = frame;
    ctx->oo_limit_m = oo_limit_m;
    ctx->matrix_outliers = outliers;
    ctx->narrow = narrow;
    ctx->ntiles = tileset.ntiles;
    ctx->tiles_ok = tileset.ok;
    ctx->scene_dense = dense_scene(sph, nsph) && balls_nested(balls);
    ctx->nquad = (int)quads.size();
    ctx->nobj = n;
    ctx->uploaded.assign((const unsigned char *)objs, (const unsigned char *)objs + nbytes);
    ctx->uploaded_valid = true;
    return RT_OK;
}

// Scene of one call.  objs != NULL: copy the first u->objCount objects (the reference's kernel loops over
// uniforms.objCount, Shaders.metal:135, and the host sends that many, ContentView.swift:682-696); this
// replaces any resident scene.  objs == NULL: use the scene kept by rt_set_scene.
int scene_for_call(rt_ctx *ctx, const Uniforms *u, const Object *objs, int objCount) {
    if (objs) {
        if (u->objCount < 0 || u->objCount > objCount)
            return fail(ctx, RT_ERR_INVALID, "rt: uniforms.objCount must be within [0, objCount]");
        const bool was_resident = ctx->scene_resident;
        ctx->scene_resident = false;
        bool same = false;
        RT_TRY(upload_scene(ctx, objs, u->objCount, &same));
        // a call that re-sends exactly the resident scene (same bytes: the device arrays were kept) leaves it resident;
        // any other scene replaces it, and a later objs == NULL call fails until rt_set_scene is called again
        if (same && was_resident) ctx->scene_resident = true;
        return RT_OK;
    }
    if (!ctx->scene_resident) return fail(ctx, RT_ERR_NO_SCENE, "rt: objs is NULL and no scene was set with rt_set_scene");
    return RT_OK;
}

void fill_params(const rt_ctx *ctx, const Uniforms *u, TraceParams &P) {
    std::memset(&P, 0, sizeof P);
    P.cam.pixelOrigin = ld3(u->pixelOrigin);
    P.cam.pixelDeltaX = ld3(u->pixelDeltaX);
    P.cam.pixelDeltaY = ld3(u->pixelDeltaY);
    P.cam.cameraCenter = ld3(u->cameraCenter);
    P.cam.defocusDiskU = ld3(u->defocusDiskU);
    P.cam.defocusDiskV = ld3(u->defocusDiskV);
    P.cam.defocusAngle = u->defocusAngle;
    P.sky = ld4(u->globalIllumation);
    P.sphg = ctx->sph.ptr;
    P.sphc = ctx->sphc.ptr;
    P.sph_obj = ctx->sph_obj.ptr;
    P.quads = ctx->quads.ptr;
    P.mats = ctx->mats.ptr;
    P.ngroups = ctx->ngroups;
    P.nsph = ctx->nsph;
    P.nquad = ctx->nquad;
    P.oo_limit = ctx->opt_filter == 1 ? -1.0f : ctx->oo_limit;
    P.sphm = ctx->sphm.ptr;
    P.tiles = ctx->tiles.ptr;
    P.tile_frames = ctx->tile_frames.ptr;
    P.word_bounds = ctx->word_bounds.ptr;
    P.round_runs = ctx->round_runs.ptr;
    P.tile_obj = ctx->tile_obj.ptr;
    P.tile_rank = ctx->tile_rank.ptr;
    P.ntiles = ctx->ntiles;
    P.lift = ctx->lift;
    P.nobj = ctx->nobj;
    P.resolve_mode = ctx->opt_resolve;
    P.sub_shift = 0;
    P.fine_begin = 0x7FFFFFFF;
    P.fine_plane0 = 0;
    P.share_begin = 0xFFFFFFFFu;
    P.regroup = ctx->opt_regroup == 2 && ctx->opt_kernel != 1 ? 1 : 0;
    P.narrow = ctx->narrow.ok && ctx->opt_narrow != 0 ? 1 : 0;  // (cleared below for the kernels that do without: P.dense_kernel)
    P.narrow_oo = ctx->narrow.oo;
    P.cull_mid = ctx->narrow.mid;
    P.cull_half = ctx->narrow.half;
    P.wide_words = ctx->narrow.words;
    for (int k = 0; k < kWideWords; ++k) {
        P.wide_word[k] = ctx->narrow.word[k];
        P.wide_test[k] = ctx->narrow.test[k];
        P.wide_always[k] = ctx->narrow.always[k];
    }
    P.dense_kernel = ctx->opt_resolve == 2 || (ctx->opt_resolve == 0 && ctx->scene_dense) ? 1 : 0;
    if (P.dense_kernel) P.narrow = 0;
    P.width = ctx->width;
    P.height = ctx->height;
    P.pass = u->currentSample;
    P.maxDepth = u->maxRayDepth;
    P.tileX = 0; P.tileY = 0; P.tileW = ctx->width; P.tileH = ctx->height;
    P.counter = ctx->counters;
    P.seg_counter = (unsigned long long *)(ctx->counters + 2);
    P.target = ctx->target.ptr;
}

// Enqueue one trace launch (counter reset, events, kernel).  noalloc: counters and camera-ray queues come from the
// device's static scratch array, the grid is cut to what its queues hold, and nothing is allocated.
int enqueue_trace(rt_ctx *ctx, TraceParams &P, bool noalloc = false) {
    // 0: staged in LDS when the scene fits, else scalar cache (measured faster than streaming on a 100 000-sphere
    // scene: 84 vs 78 TFLOP/s algorithmic); 1: scalar cache; 2: streamed through per-wave LDS tiles
    int kind = ctx->opt_fetch == 2 ? 2 : ((ctx->opt_fetch == 0 || ctx->opt_fetch == 3) && 4 * P.ngroups <= kMaxLdsSpheres) ? 0 : 1;
    // Matrix-core filter: staged scenes the lifted form covers (rt_lift.hpp), from 48 spheres up -- below that the
    // fixed cost of lifting the ray outweighs the vector filter -- or any size when asked for with RT_OPT_FILTER 3.
    if (kind == 0 && ctx->oo_limit_m > 0.0f && 4 * P.ngroups <= kMaxMatrixSpheres &&
        ((ctx->opt_filter == 0 && ctx->nsph >= 48) || ctx->opt_filter == 3)) {
        kind = 3;
        P.oo_limit = ctx->oo_limit_m;
    }
    // Tiled matrix-core filter: scenes beyond the staged matrix kind whose tiles the lifted form covers (measured
    // about twice as fast as either vector kind from 1100 spheres up), or when asked for.
    // ... or staged scenes from 768 spheres up that the single frame of the staged matrix kind does not cover while
    // the tiles' frames do (measured at 1000 spheres: 71 against 52 TFLOP/s; no gain below that).
    // ... unless rays cross many of the scene's spheres and the spheres nest (upload_scene): the kinds that keep the
    // spheres in object order prune the candidates before the exact pass (rt_hit.hpp resolve_dense_words), the tiled
    // kind does not (measured on 1500 nested shells: 38 Msamples*px/s through the scalar cache, 28 staged in LDS
    // with the vector filter, 5.6 tiled).
    const bool beyond_staged = kind == 1 || (kind == 0 && (4 * P.ngroups > kMaxMatrixSpheres || ctx->nsph >= 768));
    const bool pruned = P.dense_kernel && ctx->opt_fetch == 0 && ctx->opt_filter == 0 && ctx->opt_resolve == 0;
    if (pruned && kind == 0) kind = 1;
    if (ctx->ntiles > 0 && (ctx->opt_fetch == 3 || (ctx->opt_fetch == 0 && beyond_staged && ctx->tiles_ok && !pruned &&
                                                    (ctx->opt_filter == 0 || ctx->opt_filter == 3))))
        kind = 4;
    // ... read in place, a wave looking only at the tiles its rays can touch (RT_OPT_NARROW 1, the default: 4.4 times as
    // fast on 100 000 spheres with the boxes alone, the tiles still staged; ... without staging)
    if (kind == 4 && ctx->opt_narrow != 0) kind = 5;
    int threads = 256, per_cu = 1;
    trace_launch_shape(P, kind, &threads, &per_cu);
    if (kind == 3) {
        // the slot -> object table and the materials join the staged scene when that costs no resident waves
        P.stage_mats = 1;
        int t2 = 256, n2 = 1;
        trace_launch_shape(P, kind, &t2, &n2);
        if (ctx->nobj <= 65536 && t2 * n2 >= threads * per_cu && trace_lds_bytes(P, kind, t2) <= 160 * 1024) {
            threads = t2;
            per_cu = n2;
        } else {
            P.stage_mats = 0;
        }
    }
    if (ctx->opt_blocks_per_cu > 0) per_cu = ctx->opt_blocks_per_cu;
    long long blocks = (long long)per_cu * ctx->num_cus;
    // no more workgroups than there are waves' worth of items (a wave draws 64 at a time); small launches -- one 64 x 64
    // tile of the reference's progressive flow -- are spread over the compute units rather than packed into few
    long long need = ((long long)P.nitems + 63) / 64;
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    // ... and with workgroups no larger than the items can fill: a wave without an item exits at once, and which
    // waves those are is left to chance, so big workgroups would leave some SIMDs with four busy waves and others
    // with none (the staged matrix kind runs with 256, 512 or 1024 threads)
    if (kind == 3 && ctx->opt_kernel != 1)
        while (threads > 256 && blocks * (threads / 64) >= 2 * need) threads /= 2;
    uint32_t *counters = ctx->counters;
    if (noalloc) {
        if (ctx->opt_kernel == 1)
            return fail(ctx, RT_ERR_INVALID, "rt_render_noalloc: the wavefront formulation allocates its path pool; use RT_OPT_KERNEL 0");
        const long long fit = (long long)(kScratchQueueBytes / ((size_t)threads * kQueueRays * 2 * sizeof(float4)));
        if (blocks > fit) blocks = fit;  // which lane traces which item does not change any sum
        counters = (uint32_t *)ctx->scratch_dev;
        P.counter = counters;
        P.seg_counter = (unsigned long long *)(counters + 2);
    }
    ctx->stats_src = counters;
    // A stream that is being captured into a HIP graph records no timing events: an event recorded under capture has no
    // time of its own, and hipEventElapsedTime on it fails.  The launch is then not counted by rt_trace_times, and
    // rt_last_stats (after the graph has run) reports its counters with kernel_ms = 0.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &capture) != hipSuccess) capture = hipStreamCaptureStatusNone;
    const bool timed = capture == hipStreamCaptureStatusNone;
    RT_HIP(hipMemsetAsync(counters, 0, 256, ctx->stream));
    const int slot = (int)(ctx->launches % rt_ctx::kRing);
    int path = kind;  // what rt_stats reports
    if (ctx->opt_kernel == 1) {
        // wavefront formulation: a pool of path slots, generate/extend/shade kernels per bounce step
        if (ctx->width > 65535 || ctx->height > 65535)
            return fail(ctx, RT_ERR_TOO_LARGE, "rt: the wavefront kernels pack pixel coordinates in 16 bits");
        long long slots = ((long long)P.nitems + 255) / 256 * 256;
        if (slots > ctx->opt_wf_slots) slots = ctx->opt_wf_slots;
        if (!timed) return fail(ctx, RT_ERR_INVALID, "rt: the wavefront formulation reads its queue counters on the host between steps and cannot be captured into a graph");
        RT_TRY(grow(ctx, ctx->wf_pool, wavefront_pool_alloc_bytes((int)slots)));
        RT_HIP(hipEventRecord(ctx->ev0[slot], ctx->stream));
        int steps = 0;
        const int wf_kind = kind == 3 ? 3 : (kind == 0 ? 0 : 1);  // the tiled kind has no wavefront form: scalar cache
        path = wf_kind;
        RT_HIP(run_wavefront(P, wf_kind, wf_kind == 3 ? threads : 256, ctx->wf_pool.ptr, (int)slots, per_cu * ctx->num_cus,
                             ctx->stream, ctx->wf_host, &steps));
        RT_HIP(hipEventRecord(ctx->ev1[slot], ctx->stream));
        blocks = slots / 256;
    } else {
        if (noalloc) {
            P.rayq = (float4 *)(ctx->scratch_dev + kScratchQueueOffset);
        } else {
            RT_TRY(grow(ctx, ctx->rayq, (size_t)blocks * threads * kQueueRays * 2));
            P.rayq = ctx->rayq.ptr;
        }
        if (timed) RT_HIP(hipEventRecord(ctx->ev0[slot], ctx->stream));
        RT_HIP(launch_trace(P, kind, ctx->opt_kernel == 2, (int)blocks, threads, ctx->stream));
        if (timed) RT_HIP(hipEventRecord(ctx->ev1[slot], ctx->stream));
    }
    if (timed) ctx->launches++;
    ctx->stats_pending = true;
    ctx->stats_timed = timed;
    ctx->last = {};
    ctx->last.grid_blocks = (int)blocks;
    ctx->last.block_threads = ctx->opt_kernel == 1 ? 256 : threads;
    ctx->last.scene_path = path;
    return RT_OK;
}

}  // namespace

// ---- exported C-ABI -------------------------------------------------------------------------------------

extern "C" {

RT_API int rt_create(rt_ctx **out, int device, void *stream) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RT_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return RT_ERR_NO_DEVICE;
    rt_ctx *ctx = new rt_ctx();
    ctx->device = device;
    Bind bind(ctx);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        delete ctx;
        return RT_ERR_NO_DEVICE;
    }
    ctx->num_cus = prop.multiProcessorCount;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return RT_ERR_NO_DEVICE;
        }
        ctx->own_stream = true;
    }
    bool ok = scratch_for_device(device, &ctx->scratch_dev) == hipSuccess &&
              hipMalloc((void **)&ctx->counters, 256) == hipSuccess &&
              hipHostMalloc((void **)&ctx->wf_host, 64, hipHostMallocDefault) == hipSuccess;
    for (int i = 0; ok && i < rt_ctx::kRing; ++i)
        ok = hipEventCreate(&ctx->ev0[i]) == hipSuccess && hipEventCreate(&ctx->ev1[i]) == hipSuccess;
    if (!ok) {
        rt_destroy(ctx);
        return RT_ERR_HIP;
    }
    *out = ctx;
    return RT_OK;
}

RT_API void rt_destroy(rt_ctx *ctx) {
    if (!ctx) return;
    Bind bind(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->target.ptr);
    (void)hipFree(ctx->accum.ptr);
    (void)hipFree(ctx->planes.ptr);
    (void)hipFree(ctx->wf_pool.ptr);
    (void)hipFree(ctx->rayq.ptr);
    (void)hipFree(ctx->sph.ptr);
    (void)hipFree(ctx->sphc.ptr);
    (void)hipFree(ctx->sphm.ptr);
    (void)hipFree(ctx->tiles.ptr);
    (void)hipFree(ctx->tile_frames.ptr);
    (void)hipFree(ctx->word_bounds.ptr);
    (void)hipFree(ctx->round_runs.ptr);
    (void)hipFree(ctx->tile_obj.ptr);
    (void)hipFree(ctx->tile_rank.ptr);
    (void)hipFree(ctx->sph_obj.ptr);
    (void)hipFree(ctx->quads.ptr);
    (void)hipFree(ctx->mats.ptr);
    (void)hipFree(ctx->staging.ptr);
    (void)hipFree(ctx->counters);
    if (ctx->wf_host) (void)hipHostFree(ctx->wf_host);
    for (int i = 0; i < rt_ctx::kRing; ++i) {
        if (ctx->ev0[i]) (void)hipEventDestroy(ctx->ev0[i]);
        if (ctx->ev1[i]) (void)hipEventDestroy(ctx->ev1[i]);
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

RT_API const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : "rt: null context"; }

RT_API int rt_set_option(rt_ctx *ctx, int key, int value) {
    if (!ctx) return RT_ERR_INVALID;
    switch (key) {
        case RT_OPT_CHUNK:
            if (value < -1) return fail(ctx, RT_ERR_INVALID, "rt: chunk must be >= -1");
            ctx->opt_chunk = value;
            return RT_OK;
        case RT_OPT_TAIL_CHUNK:
            if (value < -1) return fail(ctx, RT_ERR_INVALID, "rt: tail chunk must be >= -1");
            ctx->opt_tail_chunk = value;
            return RT_OK;
        case RT_OPT_TAIL_SAMPLES:
            if (value < -1) return fail(ctx, RT_ERR_INVALID, "rt: tail samples must be >= -1");
            ctx->opt_tail_samples = value;
            return RT_OK;
        case RT_OPT_KERNEL:
            if (value < 0 || value > 2) return fail(ctx, RT_ERR_INVALID, "rt: kernel variant must be 0, 1 or 2");
            ctx->opt_kernel = value;
            return RT_OK;
        case RT_OPT_SCENE_FETCH:
            if (value < 0 || value > 3) return fail(ctx, RT_ERR_INVALID, "rt: scene fetch must be 0, 1, 2 or 3");
            ctx->opt_fetch = value;
            return RT_OK;
        case RT_OPT_BLOCKS_PER_CU:
            if (value < 0 || value > 64) return fail(ctx, RT_ERR_INVALID, "rt: blocks per CU out of range");
            ctx->opt_blocks_per_cu = value;
            return RT_OK;
        case RT_OPT_FILTER:
            if (value < 0 || value > 3) return fail(ctx, RT_ERR_INVALID, "rt: filter must be 0, 1, 2 or 3");
            ctx->opt_filter = value;
            return RT_OK;
        case RT_OPT_RESOLVE:
            if (value < 0 || value > 2) return fail(ctx, RT_ERR_INVALID, "rt: resolve form must be 0, 1 or 2");
            ctx->opt_resolve = value;
            return RT_OK;
        case RT_OPT_PLANE_SAMPLES:
            if (value < -1 || value > 64 || (value > 0 && (value & (value - 1)) != 0))
                return fail(ctx, RT_ERR_INVALID, "rt: samples per plane must be -1, 0 or a power of two up to 64");
            ctx->opt_plane = value;
            return RT_OK;
        case RT_OPT_SHARE:
            if (value < 0 || value > 1) return fail(ctx, RT_ERR_INVALID, "rt: share must be 0 or 1");
            ctx->opt_share = value;
            return RT_OK;
        case RT_OPT_NARROW:
            if (value < 0 || value > 1) return fail(ctx, RT_ERR_INVALID, "rt: narrow must be 0 or 1");
            ctx->opt_narrow = value;
            return RT_OK;
        case RT_OPT_REGROUP:
            if (value < 0 || value > 2) return fail(ctx, RT_ERR_INVALID, "rt: regroup must be 0, 1 or 2");
            ctx->opt_regroup = value;
            return RT_OK;
        case RT_OPT_WAVEFRONT_SLOTS:
            if (value < 256 || value > (1 << 24)) return fail(ctx, RT_ERR_INVALID, "rt: wavefront slots out of range");
            ctx->opt_wf_slots = value / 256 * 256;
            return RT_OK;
    }
    return fail(ctx, RT_ERR_INVALID, "rt: unknown option");
}

RT_API int rt_get_option(const rt_ctx *ctx, int key, int *value) {
    if (!ctx || !value) return RT_ERR_INVALID;
    switch (key) {
        case RT_OPT_CHUNK: *value = ctx->opt_chunk; return RT_OK;
        case RT_OPT_TAIL_CHUNK: *value = ctx->opt_tail_chunk; return RT_OK;
        case RT_OPT_TAIL_SAMPLES: *value = ctx->opt_tail_samples; return RT_OK;
        case RT_OPT_KERNEL: *value = ctx->opt_kernel; return RT_OK;
        case RT_OPT_SCENE_FETCH: *value = ctx->opt_fetch; return RT_OK;
        case RT_OPT_BLOCKS_PER_CU: *value = ctx->opt_blocks_per_cu; return RT_OK;
        case RT_OPT_FILTER: *value = ctx->opt_filter; return RT_OK;
        case RT_OPT_RESOLVE: *value = ctx->opt_resolve; return RT_OK;
        case RT_OPT_REGROUP: *value = ctx->opt_regroup; return RT_OK;
        case RT_OPT_PLANE_SAMPLES: *value = ctx->opt_plane; return RT_OK;
        case RT_OPT_SHARE: *value = ctx->opt_share; return RT_OK;
        case RT_OPT_NARROW: *value = ctx->opt_narrow; return RT_OK;
        case RT_OPT_WAVEFRONT_SLOTS: *value = ctx->opt_wf_slots; return RT_OK;
    }
    return RT_ERR_INVALID;
}

RT_API int rt_set_target(rt_ctx *ctx, int width, int height) {
    if (!ctx) return RT_ERR_INVALID;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 28))
        return fail(ctx, RT_ERR_INVALID, "rt: bad target size");
    Bind bind(ctx);
    RT_HIP(hipStreamSynchronize(ctx->stream));
    size_t n = (size_t)width * height;
    RT_TRY(grow(ctx, ctx->target, n));
    RT_TRY(grow(ctx, ctx->accum, n));
    ctx->width = width;
    ctx->height = height;
    RT_TRY(rt_clear(ctx));
    RT_TRY(rt_accum_clear(ctx, nullptr));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

RT_API int rt_clear(rt_ctx *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->target.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    Bind bind(ctx);
    RT_HIP(launch_fill_rect(ctx->target.ptr, ctx->width, ctx->height, 0, 0, ctx->width, ctx->height,
                            make_float4(0.0f, 0.0f, 0.0f, 1.0f), ctx->stream));
    return RT_OK;
}

RT_API int rt_write_target(rt_ctx *ctx, const float *rgba) {
    if (!ctx || !rgba) return RT_ERR_INVALID;
    if (!ctx->target.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    Bind bind(ctx);
    RT_HIP(hipMemcpyAsync(ctx->target.ptr, rgba, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyHostToDevice,
                          ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

RT_API int rt_read_target(rt_ctx *ctx, float *rgba) {
    if (!ctx || !rgba) return RT_ERR_INVALID;
    if (!ctx->target.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    Bind bind(ctx);
    RT_HIP(hipMemcpyAsync(rgba, ctx->target.ptr, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyDeviceToHost,
                          ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

RT_API int rt_set_scene(rt_ctx *ctx, const Object *objs, int objCount) {
    if (!ctx) return RT_ERR_INVALID;
    Bind bind(ctx);
    ctx->scene_resident = false;
    RT_TRY(upload_scene(ctx, objs, objCount));
    ctx->scene_resident = true;
    return RT_OK;
}

RT_API int rt_render_tile(rt_ctx *ctx, const Uniforms *u, const Object *objs, int objCount) {
    if (!ctx || !u) return RT_ERR_INVALID;
    if (!ctx->target.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    if (u->tileWidth <= 0 || u->tileHeight <= 0 || u->tileX < 0 || u->tileY < 0)
        return fail(ctx, RT_ERR_INVALID, "rt: tile rectangle must have positive size and non-negative origin");
    Bind bind(ctx);
    RT_TRY(scene_for_call(ctx, u, objs, objCount));

    // The reference dispatches ceil(tile/8) groups of 8x8 threads (ContentView.swift:698-705).
    const int gw = (u->tileWidth + 7) / 8 * 8, gh = (u->tileHeight + 7) / 8 * 8;
    const bool all_fail = u->currentSample > 10000;  // Shaders.metal:111-114
    const bool clipped = u->tileX + u->tileWidth > ctx->width || u->tileY + u->tileHeight > ctx->height;
    int status = RT_OK;
    if (all_fail || clipped || gw != u->tileWidth || gh != u->tileHeight) {
        RT_HIP(launch_error_pixels(ctx->target.ptr, ctx->width, ctx->height, gw, gh, u->tileX, u->tileY, u->tileWidth,
                                   u->tileHeight, all_fail ? 1 : 0, ctx->stream));
        status = 1;
    }
    int tw = u->tileWidth, th = u->tileHeight;
    if (u->tileX + tw > ctx->width) tw = ctx->width - u->tileX;
    if (u->tileY + th > ctx->height) th = ctx->height - u->tileY;
    if (!all_fail && tw > 0 && th > 0) {
        TraceParams P;
        fill_params(ctx, u, P);
        P.tileX = u->tileX; P.tileY = u->tileY; P.tileW = tw; P.tileH = th;
        P.ray_begin = 0;
        P.ray_end = u->sampleCount > 0 ? u->sampleCount : 0;
        P.spp_pass = u->sampleCount;
        // The reference sums a pixel's samples one after the other (Shaders.metal:120-140), and a 64 x 64 tile has 4096
        // pixels for a GPU of 262 144 lanes.  So every sample is a work item of its own, its colour goes to a plane of
        // the tile (plane k = sample k), and a second kernel adds the planes in sample order -- the same additions in the
        // same order -- and applies the pass's blend.  When the planes would be too large (a "tile" that is a whole
        // frame with thousands of samples) a pixel's samples stay in one item and the kernel blends by itself.
        const size_t plane_bytes = (size_t)P.ray_end * (size_t)tw * (size_t)th * sizeof(float4);
        if (P.ray_end > 1 && plane_bytes <= kTilePlaneBytesMax && (unsigned long long)P.ray_end * tw * th <= 0xF0000000ull) {
            RT_TRY(grow(ctx, ctx->planes, plane_bytes / sizeof(float4)));
            P.chunk = P.chunk_small = 1;
            P.nbig = P.ray_end;
            P.nitems = (uint32_t)((unsigned long long)P.ray_end * tw * th);
            P.mode = kModeAccum;
            P.planes = ctx->planes.ptr;
            P.plane_x0 = P.tileX; P.plane_y0 = P.tileY; P.plane_w = tw;
            P.plane_stride = (unsigned long long)tw * th;
            RT_TRY(enqueue_trace(ctx, P));
            RT_HIP(launch_blend_tile(ctx->target.ptr, P.planes, P.ray_end, ctx->width, P.tileX, P.tileY, tw, th, P.pass, P.spp_pass,
                                     ctx->stream));
            ctx->last.chunk = 1;
            ctx->last.chunk_count = P.ray_end;
            ctx->last.tail_chunk = 1;
        } else {
            P.chunk = P.chunk_small = P.ray_end > 0 ? P.ray_end : 1;
            P.nbig = 1;
            P.nitems = (uint32_t)tw * (uint32_t)th;
            P.mode = kModeProgressive;
            RT_TRY(enqueue_trace(ctx, P));
        }
        ctx->last.samples = (uint64_t)tw * th * (uint64_t)P.ray_end;
    }
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return status;
}

RT_API int rt_present_bgra8(rt_ctx *ctx, uint8_t *out, int out_w, int out_h) {
    if (!ctx || !out || out_w <= 0 || out_h <= 0) return RT_ERR_INVALID;
    if (!ctx->target.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    Bind bind(ctx);
    size_t bytes = (size_t)out_w * out_h * 4;
    RT_TRY(grow(ctx, ctx->staging, bytes));
    uint32_t *dev = (uint32_t *)ctx->staging.ptr;
    hipError_t e = launch_present_bgra8(ctx->target.ptr, ctx->width, ctx->height, dev, out_w, out_h, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_present_bgra8", e);
    return RT_OK;
}

static int render_batch(rt_ctx *ctx, const Uniforms *u, const Object *objs, int objCount, int ray_begin, int ray_end,
                        float *accum_dev, bool noalloc) {
    if (!ctx || !u) return RT_ERR_INVALID;
    if (!ctx->target.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    if (ray_begin < 0 || ray_end < ray_begin) return fail(ctx, RT_ERR_INVALID, "rt: bad ray index range");
    if (u->tileWidth <= 0 || u->tileHeight <= 0 || u->tileX < 0 || u->tileY < 0 || u->tileX + u->tileWidth > ctx->width ||
        u->tileY + u->tileHeight > ctx->height)
        return fail(ctx, RT_ERR_INVALID, "rt: tile rectangle must lie inside the render target");
    if (u->currentSample > 10000) return fail(ctx, RT_ERR_INVALID, "rt: currentSample > 10000 (the reference paints yellow)");
    Bind bind(ctx);
    if (noalloc) {
        if (objs) return fail(ctx, RT_ERR_INVALID, "rt_render_noalloc: objs must be NULL (uploading a scene allocates); call rt_set_scene first");
        if (!accum_dev && !ctx->accum.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    }
    RT_TRY(scene_for_call(ctx, u, objs, objCount));
    if (ray_end == ray_begin) return RT_OK;

    const int nrays = ray_end - ray_begin;
    // Work items: a pixel's samples are cut into `nbig` chunks of `chunk` samples followed by chunks of `small`
    // samples, and items are handed out chunk by chunk.  opt_chunk > 0: chunk as given; 0: one item per pixel (the
    // reference's sequential sum); < 0: automatic.
    int chunk = ctx->opt_chunk, small = ctx->opt_tail_chunk, tail = ctx->opt_tail_samples;
    const long long lanes = (long long)ctx->num_cus * 2048;
    const long long pixels = (long long)u->tileWidth * u->tileHeight;
    const bool automatic = chunk < 0;
    if (chunk < 0) {
        // Long items cost little at the end of a launch -- lanes without work take whole planes over from other lanes of
        // their wave (below) -- but a wave draws 64 items at a time and cannot take anything over from another wave,
        // so there must be several items per resident lane; short items pay for drawing the item and for generating
        // its camera rays (one wave-wide batch, however few rays).  Measured on the cover scene (tools/exp_chunk.py):
        // 0.4 of the samples a lane of a twice-resident grid would get, at most 64 (one batch), is within 4 % of the best
        // fixed size from 320x180 x 64 to 1920x1080 x 1000.
        // Where tracing a sample is cheap (few objects) the batch dominates and longer items win even when that
        // leaves fewer items than lanes, down to one item per resident lane: the second term (three spheres 400x225 x
        // 100: 1730 Msamples*px/s with items of 8, 3840 with 32, 2570 with 64).
        const double per_lane = (double)nrays * (double)pixels / (double)lanes;
        const double objects = (double)std::max(1, ctx->nsph + ctx->nquad);
        chunk = (int)std::lround(std::max(0.4 * per_lane, std::min(2.0 * per_lane, std::sqrt(4000.0 * per_lane / objects))));
        chunk = std::max(1, std::min(64, chunk));
        if (chunk >= 6) chunk = std::min(64, (chunk + 2) & ~3);  // whole planes of 4 samples (below)
        // The wavefront kernels run as many bounce steps as the slot with the most segments needs, and a slot works its
        // item's samples off one after the other: short items (a sample each up to 64 samples per pixel, so that there
        // are at most 64 planes) make the launch 2.2 times shorter on the nested shells and 1.2 on the cover scene.
        if (ctx->opt_kernel == 1) chunk = std::max(1, (nrays + 63) / 64);
    }
    // A tail of short items after the long ones lets the grid drain more evenly, but on the cover scene it costs
    // more in ray generation than it gains (1 to 2 %): off unless asked for.
    if (small < 0) small = 0;
    if (chunk == 0 || chunk > nrays) chunk = nrays;
    int nbig, nchunks;
    if (small <= 0 || small >= chunk) {
        small = chunk;
        nbig = nchunks = (nrays + chunk - 1) / chunk;
    } else {
        // the short items must outlast the longest long item drawn just before them: a long item whose paths all
        // run to the depth limit takes about 2.5 mean items, while the rest of the grid works through the tail
        if (tail < 0) tail = (int)std::min<long long>(nrays / 2, (5 * (long long)chunk * lanes + 2 * pixels - 1) / (2 * pixels));
        tail = std::min(tail, nrays);
        nbig = (nrays - tail) / chunk;
        nchunks = nbig + (nrays - nbig * chunk + small - 1) / small;
    }
    const unsigned long long nitems = (unsigned long long)pixels * nchunks;
    if (nitems > 0xF0000000ull) return fail(ctx, RT_ERR_TOO_LARGE, "rt: too many work items; raise RT_OPT_CHUNK");
    // The items drawn last -- the last chunks of every pixel: items are handed out chunk by chunk, and when the counter
    // runs out the grid holds about lanes / pixels chunks' worth of them -- are cut into finer planes, so that lanes
    // without work can take whole planes over from others (trace_kernel): automatic for items of 16 samples or more,
    // or as asked.  Every item must be a whole number of planes.
    int plane = ctx->opt_plane;
    if (plane < 0) plane = automatic && small == chunk && chunk >= 8 && chunk % 4 == 0 ? 4 : 0;  // (a given item size is summed as given)
    if (plane > 0 && (small % plane != 0 || chunk % plane != 0 || plane >= chunk)) plane = 0;
    if (ctx->opt_kernel == 1) plane = 0;  // the wavefront kernels keep one sum per work item
    int coarse = nchunks, nplanes = nchunks, sub_shift = 0;
    if (plane > 0) {
        const int fine_layers = (int)std::min<long long>(nchunks, (lanes + pixels - 1) / pixels + 1);
        // (with shorter items at the end: those and the last layer of long ones)
        coarse = small == chunk ? nchunks - fine_layers : std::max(0, nbig - 1);
        nplanes = coarse + (nrays - coarse * chunk + plane - 1) / plane;
        if (noalloc && (size_t)nplanes * ctx->width * ctx->height * sizeof(float4) > kScratchPlaneBytes) {
            plane = 0;   // the static scratch holds fewer planes: one per item
            coarse = nplanes = nchunks;
        }
    }
    if (plane > 0)
        for (sub_shift = 0; (1 << sub_shift) < plane; ++sub_shift) {}
    float4 *planes;
    if (noalloc) {
        const size_t need = (size_t)nplanes * ctx->width * ctx->height * sizeof(float4);
        if (need > kScratchPlaneBytes) {
            char msg[256];
            std::snprintf(msg, sizeof msg,
                          "rt_render_noalloc: %d partial-sum planes of %dx%d need %zu bytes, the static scratch holds %zu; "
                          "raise RT_OPT_CHUNK, trace fewer samples per call, or use rt_render",
                          nplanes, ctx->width, ctx->height, need, kScratchPlaneBytes);
            return fail(ctx, RT_ERR_TOO_LARGE, msg);
        }
        planes = (float4 *)(ctx->scratch_dev + kScratchPlaneOffset);
    } else {
        RT_TRY(grow(ctx, ctx->planes, (size_t)nplanes * ctx->width * ctx->height));
        planes = ctx->planes.ptr;
    }

    TraceParams P;
    fill_params(ctx, u, P);
    P.tileX = u->tileX; P.tileY = u->tileY; P.tileW = u->tileWidth; P.tileH = u->tileHeight;
    P.ray_begin = ray_begin;
    P.ray_end = ray_end;
    P.chunk = chunk;
    P.nbig = nbig;
    P.chunk_small = small;
    if (plane > 0) {
        P.sub_shift = sub_shift;
        P.fine_begin = ray_begin + coarse * chunk;
        P.fine_plane0 = coarse;
        // the lanes of a wave work the last item or so per resident lane off together (never before the fine planes begin)
        if (ctx->opt_share == 0) {
            const unsigned long long first_fine = (unsigned long long)coarse * (unsigned long long)pixels;
            const unsigned long long last = nitems > (unsigned long long)lanes ? nitems - (unsigned long long)lanes : 0;
            P.share_begin = (uint32_t)std::max(first_fine, last);
        }
    }
    P.nitems = (uint32_t)nitems;
    P.mode = kModeAccum;
    P.planes = planes;
    P.plane_x0 = 0; P.plane_y0 = 0; P.plane_w = ctx->width;
    P.plane_stride = (unsigned long long)ctx->width * ctx->height;
    RT_TRY(enqueue_trace(ctx, P, noalloc));
    ctx->last.samples = (uint64_t)u->tileWidth * u->tileHeight * (uint64_t)nrays;
    // the schedule of the sums, as rt_stats reports it: runs of `chunk` samples (planes), added in order
    ctx->last.chunk = chunk;
    ctx->last.chunk_count = plane > 0 ? coarse : nbig;
    ctx->last.tail_chunk = plane > 0 ? plane : small;
    float4 *acc = accum_dev ? (float4 *)accum_dev : ctx->accum.ptr;
    RT_HIP(launch_fold(acc, planes, nplanes, ctx->width, ctx->height, P.tileX, P.tileY, P.tileW, P.tileH,
                       ctx->stream));
    return RT_OK;
}

RT_API int rt_render(rt_ctx *ctx, const Uniforms *u, const Object *objs, int objCount, int ray_begin, int ray_end,
                     float *accum_dev) {
    return render_batch(ctx, u, objs, objCount, ray_begin, ray_end, accum_dev, false);
}

RT_API int rt_render_noalloc(rt_ctx *ctx, const Uniforms *u, int ray_begin, int ray_end, float *accum_dev) {
    return render_batch(ctx, u, nullptr, 0, ray_begin, ray_end, accum_dev, true);
}

RT_API int rt_accum_clear(rt_ctx *ctx, float *accum_dev) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->target.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    Bind bind(ctx);
    float4 *acc = accum_dev ? (float4 *)accum_dev : ctx->accum.ptr;
    RT_HIP(hipMemsetAsync(acc, 0, (size_t)ctx->width * ctx->height * sizeof(float4), ctx->stream));
    return RT_OK;
}

RT_API int rt_accum_ptr(rt_ctx *ctx, float **accum_dev) {
    if (!ctx || !accum_dev) return RT_ERR_INVALID;
    if (!ctx->accum.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    *accum_dev = (float *)ctx->accum.ptr;
    return RT_OK;
}

RT_API int rt_resolve(rt_ctx *ctx, const float *accum_dev, int spp) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->target.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    Bind bind(ctx);
    const float4 *acc = accum_dev ? (const float4 *)accum_dev : ctx->accum.ptr;
    RT_HIP(launch_resolve(ctx->target.ptr, acc, ctx->width * ctx->height, spp, ctx->stream));
    return RT_OK;
}

RT_API int rt_synchronize(rt_ctx *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    Bind bind(ctx);
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

RT_API int rt_first_hit(rt_ctx *ctx, const Uniforms *u, const Object *objs, int objCount, float *out4, int32_t *ids) {
    if (!ctx || !u || !out4 || !ids) return RT_ERR_INVALID;
    if (!ctx->target.ptr) return fail(ctx, RT_ERR_NO_TARGET, "rt: no render target");
    Bind bind(ctx);
    RT_TRY(scene_for_call(ctx, u, objs, objCount));
    TraceParams P;
    fill_params(ctx, u, P);
    size_t n = (size_t)ctx->width * ctx->height;
    RT_TRY(grow(ctx, ctx->staging, n * (sizeof(float4) + sizeof(int))));
    float4 *d4 = (float4 *)ctx->staging.ptr;
    int *dids = (int *)(d4 + n);
    hipError_t e = launch_first_hit(P, d4, dids, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out4, d4, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ids, dids, n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_first_hit", e);
    return RT_OK;
}

RT_API int rt_last_stats(rt_ctx *ctx, rt_stats *out) {
    if (!ctx || !out) return RT_ERR_INVALID;
    Bind bind(ctx);
    if (ctx->stats_pending) {
        RT_HIP(hipStreamSynchronize(ctx->stream));
        float ms = 0.0f;
        if (ctx->stats_timed) {
            const int slot = (int)((ctx->launches - 1) % rt_ctx::kRing);
            RT_HIP(hipEventElapsedTime(&ms, ctx->ev0[slot], ctx->ev1[slot]));
        }
        uint32_t host[24] = {0};
        RT_HIP(hipMemcpyAsync(host, ctx->stats_src ? ctx->stats_src : ctx->counters, sizeof host, hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(hipStreamSynchronize(ctx->stream));
        uint64_t seg = (uint64_t)host[2] | ((uint64_t)host[3] << 32);
        ctx->last.kernel_ms = ms;
        ctx->last.segments = seg;
        ctx->last.tests = seg * (uint64_t)ctx->nobj;
        for (int i = 0; i < 8; ++i) ctx->last.phase_cycles[i] = (uint64_t)host[4 + 2 * i] | ((uint64_t)host[5 + 2 * i] << 32);
        ctx->last.wave_trips = (uint64_t)host[20] | ((uint64_t)host[21] << 32);
        ctx->last.matrix_blocks = ((uint64_t)host[22] | ((uint64_t)host[23] << 32)) / 2u;  // counted in halves (rt_hit.hpp Work)
        ctx->stats_pending = false;
    }
    *out = ctx->last;
    return RT_OK;
}

RT_API int rt_selftest_filter(rt_ctx *ctx, uint32_t seed, int blocks, int pairs_per_thread, uint64_t out[6]) {
    if (!ctx || !out || blocks <= 0 || pairs_per_thread <= 0) return RT_ERR_INVALID;
    Bind bind(ctx);
    RT_TRY(grow(ctx, ctx->staging, 256));
    unsigned long long *dev = (unsigned long long *)ctx->staging.ptr;
    hipError_t e = hipMemsetAsync(dev, 0, 256, ctx->stream);
    if (e == hipSuccess) e = launch_filter_selftest(seed, blocks, pairs_per_thread, dev, ctx->stream);
    unsigned long long host[6] = {0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof host, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_selftest_filter", e);
    for (int i = 0; i < 6; ++i) out[i] = host[i];
    return RT_OK;
}

RT_API int rt_selftest_prune(rt_ctx *ctx, uint32_t seed, int blocks, int rays_per_thread, uint64_t out[6]) {
    if (!ctx || !out || blocks <= 0 || rays_per_thread <= 0) return RT_ERR_INVALID;
    if (!ctx->scene_resident || ctx->nsph <= 0) return fail(ctx, RT_ERR_NO_SCENE, "rt_selftest_prune: set a scene with spheres first (rt_set_scene)");
    Bind bind(ctx);
    RT_TRY(grow(ctx, ctx->staging, 256));
    unsigned long long *dev = (unsigned long long *)ctx->staging.ptr;
    hipError_t e = hipMemsetAsync(dev, 0, 256, ctx->stream);
    if (e == hipSuccess) e = launch_prune_selftest(seed, blocks, rays_per_thread, ctx->sph.ptr, ctx->word_bounds.ptr, ctx->nsph, dev, ctx->stream);
    unsigned long long host[6] = {0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof host, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_selftest_prune", e);
    for (int i = 0; i < 6; ++i) out[i] = host[i];
    return RT_OK;
}

RT_API int rt_selftest_box(rt_ctx *ctx, uint32_t seed, int blocks, int rays_per_thread, int padding_log2, uint64_t out[6]) {
    if (!ctx || !out || blocks <= 0 || rays_per_thread <= 0 || padding_log2 > 0 || padding_log2 < -40) return RT_ERR_INVALID;
    if (!ctx->scene_resident || ctx->nsph <= 0) return fail(ctx, RT_ERR_NO_SCENE, "rt_selftest_box: set a scene with spheres first (rt_set_scene)");
    if (!ctx->narrow.ok) return fail(ctx, RT_ERR_INVALID, "rt_selftest_box: the scene has no box (fewer than 128 spheres, more than 8 outside it, or not covered by the lifted form)");
    Bind bind(ctx);
    TraceParams P;
    Uniforms u;
    std::memset(&u, 0, sizeof u);
    fill_params(ctx, &u, P);
    if (padding_log2 < 0) {  // the same box with its allowances scaled down
        std::vector<float4> sph((size_t)ctx->nsph);
        RT_HIP(hipMemcpyAsync(sph.data(), ctx->sph.ptr, sph.size() * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<char> ok((size_t)ctx->nsph, 0);
        for (int i = 0; i < ctx->nsph; ++i) {
            _Float16 slots[kLiftSlots];
            ok[(size_t)i] = lift_sphere(sph[(size_t)i].x, sph[(size_t)i].y, sph[(size_t)i].z, sph[(size_t)i].w, ctx->lift, (double)kMfmaSlack, slots) ? 1 : 0;
        }
        const NarrowSet N = narrow_setup(sph, ctx->nsph, ok, ctx->lift, ctx->oo_limit_m, padding_log2 <= -40 ? 0.0 : std::ldexp(1.0, padding_log2));
        if (!N.ok) return fail(ctx, RT_ERR_INVALID, "rt_selftest_box: no box at this padding");
        P.cull_mid = N.mid;
        P.cull_half = N.half;
    }
    RT_TRY(grow(ctx, ctx->staging, 256));
    unsigned long long *dev = (unsigned long long *)ctx->staging.ptr;
    hipError_t e = hipMemsetAsync(dev, 0, 256, ctx->stream);
    if (e == hipSuccess) e = launch_box_selftest(seed, blocks, rays_per_thread, ctx->sph.ptr, ctx->nsph, P, dev, ctx->stream);
    unsigned long long host[6] = {0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof host, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_selftest_box", e);
    for (int i = 0; i < 6; ++i) out[i] = host[i];
    return RT_OK;
}

RT_API int rt_selftest_tile_boxes(rt_ctx *ctx, uint32_t seed, int blocks, int rays_per_thread, int padding_log2, uint64_t out[6]) {
    if (!ctx || !out || blocks <= 0 || rays_per_thread <= 0 || padding_log2 > 0 || padding_log2 < -40) return RT_ERR_INVALID;
    if (!ctx->scene_resident || ctx->nsph <= 0 || ctx->ntiles <= 0)
        return fail(ctx, RT_ERR_NO_SCENE, "rt_selftest_tile_boxes: set a scene with spheres first (rt_set_scene)");
    Bind bind(ctx);
    TraceParams P;
    Uniforms u;
    std::memset(&u, 0, sizeof u);
    fill_params(ctx, &u, P);
    RT_TRY(grow(ctx, ctx->staging, 256));
    unsigned long long *dev = (unsigned long long *)ctx->staging.ptr;
    hipError_t e = hipMemsetAsync(dev, 0, 256, ctx->stream);
    if (e == hipSuccess) e = launch_tile_box_selftest(seed, blocks, rays_per_thread, P, padding_log2 <= -40 ? 0.0f : std::ldexp(1.0f, padding_log2), dev, ctx->stream);
    unsigned long long host[6] = {0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof host, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_selftest_tile_boxes", e);
    for (int i = 0; i < 6; ++i) out[i] = host[i];
    return RT_OK;
}

static int selftest_matrix(rt_ctx *ctx, uint32_t seed, int blocks, int rounds, int slack_log2, int far, uint64_t out[6]) {
    if (!ctx || !out || blocks <= 0 || rounds <= 0 || slack_log2 > -8 || slack_log2 < -30) return RT_ERR_INVALID;
    Bind bind(ctx);
    RT_TRY(grow(ctx, ctx->staging, 256));
    unsigned long long *dev = (unsigned long long *)ctx->staging.ptr;
    hipError_t e = hipMemsetAsync(dev, 0, 256, ctx->stream);
    if (e == hipSuccess) e = launch_matrix_selftest(seed, blocks, rounds, std::ldexp(1.0f, slack_log2), far, dev, ctx->stream);
    unsigned long long host[6] = {0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof host, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_selftest_matrix_filter", e);
    for (int i = 0; i < 6; ++i) out[i] = host[i];
    return RT_OK;
}

RT_API int rt_selftest_matrix_filter(rt_ctx *ctx, uint32_t seed, int blocks, int rounds, int slack_log2, uint64_t out[6]) {
    return selftest_matrix(ctx, seed, blocks, rounds, slack_log2, 0, out);
}

RT_API int rt_selftest_tiled_filter(rt_ctx *ctx, uint32_t seed, int blocks, int rounds, int slack_log2, uint64_t out[6]) {
    return selftest_matrix(ctx, seed, blocks, rounds, slack_log2, 1, out);
}

RT_API int rt_group_bounds(const float *spheres_xyzr, int count, float *out) {
    static_assert(RT_BOUND_FLOATS == 4 * kBoundVec4, "rt_amd.h and rt_kernels.h disagree about the bounds record");
    if (count < 0 || (count > 0 && !spheres_xyzr) || !out) return RT_ERR_INVALID;
    std::vector<float4> sph((size_t)count);
    for (int i = 0; i < count; ++i)
        sph[(size_t)i] = make_float4(spheres_xyzr[4 * i], spheres_xyzr[4 * i + 1], spheres_xyzr[4 * i + 2], spheres_xyzr[4 * i + 3]);
    const std::vector<float4> b = word_bounds(sph, count);
    if (!b.empty()) std::memcpy(out, b.data(), b.size() * sizeof(float4));
    return (int)(b.size() / (size_t)kBoundVec4);
}

RT_API int rt_filter_box(const float *spheres_xyzr, int count, float out[RT_FILTER_BOX_FLOATS]) {
    static_assert(RT_FILTER_BOX_FLOATS == 8 + 3 * kWideWords, "rt_amd.h and rt_kernels.h disagree about the record");
    if (count < 0 || (count > 0 && !spheres_xyzr) || !out) return RT_ERR_INVALID;
    std::vector<float4> sph((size_t)count);
    for (int i = 0; i < count; ++i)
        sph[(size_t)i] = make_float4(spheres_xyzr[4 * i], spheres_xyzr[4 * i + 1], spheres_xyzr[4 * i + 2], spheres_xyzr[4 * i + 3]);
    const LiftSet L = lift_setup(sph, count);
    const NarrowSet N = narrow_setup(sph, count, L.lifted_ok, L.frame, L.oo_limit_m);
    for (int i = 0; i < RT_FILTER_BOX_FLOATS; ++i) out[i] = 0.0f;
    if (!N.ok) return 0;
    out[0] = N.mid.x; out[1] = N.mid.y; out[2] = N.mid.z;
    out[3] = N.half.x; out[4] = N.half.y; out[5] = N.half.z;
    out[6] = std::sqrt(N.oo) / L.frame.sigma;  // in scene units
    out[7] = (float)N.words;
    for (int k = 0; k < kWideWords; ++k) {
        uint32_t w = N.word[k], t = N.test[k], a = N.always[k];
        std::memcpy(&out[8 + 3 * k], &w, 4);
        std::memcpy(&out[9 + 3 * k], &t, 4);
        std::memcpy(&out[10 + 3 * k], &a, 4);
    }
    return 1;
}

RT_API int rt_trace_times(rt_ctx *ctx, float *ms, int cap, int *count) {
    if (!ctx || !ms || !count || cap < 0) return RT_ERR_INVALID;
    Bind bind(ctx);
    *count = 0;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long first = ctx->times_read;
    if (ctx->launches - first > (unsigned long long)rt_ctx::kRing) first = ctx->launches - rt_ctx::kRing;
    for (unsigned long long i = first; i < ctx->launches && *count < cap; ++i) {
        const int slot = (int)(i % rt_ctx::kRing);
        RT_HIP(hipEventElapsedTime(&ms[*count], ctx->ev0[slot], ctx->ev1[slot]));
        ++*count;
    }
    ctx->times_read = ctx->launches;
    return RT_OK;
}

}  // extern "C"
