// rt_hit.hpp -- the closest-hit search shared by the persistent kernel (rt_trace.hip) and the wavefront kernels
// (rt_wavefront.hip): record fetchers, the conservative sphere filter (pass A), the exact resolve (pass B), quads,
// and the hit record.  Restates world_hit / hitObject / hitSphere / hitPlane of the reference
// (reference: "Ray Tracer/Shaders.metal":13-32, "Ray Tracer/Lib.metal":215-291).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "rt_device.hpp"
#include "rt_kernels.h"
#include "rt_lift.hpp"

namespace rt {

constexpr int kWordGroups = 8;    // groups of 4 spheres per 32-bit candidate word
constexpr int kRoundWords = 16;   // candidate words per resolve round (512 spheres)
constexpr int kListOffset = kRoundWords * 64;        // after a wave's candidate words: 64 words of candidate list ...
constexpr int kBestOffset = kListOffset + 64;        // ... and 64 x 64 bits of (distance, slot) minima (resolve_balanced)
constexpr int kMaskWordsPerWave = kBestOffset + 128; // LDS words per wave
#ifndef RT_FEW_RAYS
#define RT_FEW_RAYS 12
#endif
#ifndef RT_REST_POOLED
#define RT_REST_POOLED 64
#endif
constexpr int kRestPooled = RT_REST_POOLED;  // dense rounds: candidates in the whole wave after the prune up to which the pooled form resolves them
constexpr int kRestPerLane = 12;   // dense rounds: candidates per lane after the prune up to which every lane resolves its own
constexpr int kFewRays = RT_FEW_RAYS;  // tiled kind read in place: rays of a wave that touch a tile up to which every lane tests its share of the tile's
                                       // slots exactly for one ray after the other (closest_sphere_direct); more go through the matrix-core filter
constexpr int kFewSpheres = 8;      // tiled kind read in place: spheres in a tile up to which no filter runs and all are candidates
constexpr int kDenseWordBits = 8;  // candidates in a word of 32 spheres from which the word counts as dense (closest_sphere)

// Filter margins (see closest_sphere).  kDirScale inflates the b side of both filters; the c side is inflated
// on the host when the records are built: r*r * (1 + 2^-17) in the general records, and a slack of
// kCompactSlack * (|o|^2 + |c|^2 + r^2) in the compact form, whose per-ray share is applied here.
constexpr float kDirScale = 1.0f + 0x1p-18f;
constexpr float kRadiusScale = 1.0f + 0x1p-17f;
constexpr float kCompactSlack = 0x1p-16f;

typedef const f4v __attribute__((address_space(4))) *const_f4_ptr;

struct Rec4 { f4v r0, r1, r2, r3; };

// Where pass A reads its records from.  All three kinds expose: start(general) -- choose the record array for
// this closest-hit search; begin_word(group) -- called before each 32-sphere word; load4(group) / settle();
// load1(slot) -- the strict record (center, radius) of one sphere for pass B and the hit record.
// kFetchMatrix is the staged kind with the matrix-core filter in place of the compact vector filter: strict
// records and the lifted A operands (rt_lift.hpp) are staged in LDS.
// kFetchTiled is the matrix-core filter for scenes beyond that: the spheres are partitioned in space into tiles of
// at most 512, each lifted in a frame of its own; a workgroup streams the tiles through two LDS buffers
// (asynchronous global -> LDS copies), all its waves working on the same tile between two barriers.
// kFetchTiledDirect works on the same tiles where they lie in global memory: every wave first asks which tiles the rays
// it holds can touch at all (a box per tile and per group of 64 tiles) and runs the filter for those tiles alone -- a
// few of hundreds -- with its operands read straight from the L2 cache; no staging, no barriers, waves independent.
enum { kFetchLds = 0, kFetchScalar = 1, kFetchStream = 2, kFetchMatrix = 3, kFetchTiled = 4, kFetchTiledDirect = 5 };
constexpr int kTileRecords = 128;  // kFetchStream: records per LDS tile (two tiles per wave)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
struct Fetch;

template <>
struct Fetch<kFetchLds> {  // both arrays staged once per workgroup in LDS, read with a wave-uniform address (broadcast)
    static constexpr bool kMatrix = false;
    static constexpr bool kTiled = false;
    static constexpr bool kScalarRecords = false;
    const f4v *pc, *pg, *p;
    RT_DEV void start(bool general) { p = general ? pg : pc; }
    RT_DEV void begin_word(int) {}
    RT_DEV Rec4 load4(int group) const {
        const f4v *q = p + 4 * group;
        Rec4 r = {q[0], q[1], q[2], q[3]};
        return r;
    }
    RT_DEV void settle(Rec4 &) const {}
    RT_DEV f4v load1(int k) const { return pg[k]; }
    RT_DEV float radius(int k) const { return ((const float *)(pg + k))[3]; }
    RT_DEV int object_of(const TraceParams &P, int slot) const { return P.sph_obj[slot]; }
    RT_DEV MatDev material(const TraceParams &P, int obj) const { return P.mats[obj]; }
};

template <>
struct Fetch<kFetchScalar> {  // scalar cache: uniform address -> s_load_dwordx16 into scalar registers
    static constexpr bool kMatrix = false;
    static constexpr bool kTiled = false;
    static constexpr bool kScalarRecords = true;
    const_f4_ptr pc, pg, p;
    const f4v *strict;
    RT_DEV void start(bool general) { p = general ? pg : pc; }
    RT_DEV void begin_word(int) {}
    RT_DEV Rec4 load4(int group) const {
        const_f4_ptr q = p + 4 * group;
        Rec4 r = {q[0], q[1], q[2], q[3]};
        return r;
    }
    // Scalar loads return out of order, so the only wait is "all of them".  Forcing that wait here, before the
    // next group's load is issued, keeps exactly one group in flight behind the one being computed on.
    // The records pass through the statement as read-write operands so that the arithmetic on them cannot be
    // scheduled above it.
    RT_DEV void settle(Rec4 &r) const {
        asm volatile("" : "+s"(r.r0), "+s"(r.r1), "+s"(r.r2), "+s"(r.r3)::"memory");
    }
    RT_DEV f4v load1(int k) const { return strict[k]; }
    RT_DEV float radius(int k) const { return ((const float *)(strict + k))[3]; }
    RT_DEV int object_of(const TraceParams &P, int slot) const { return P.sph_obj[slot]; }
    RT_DEV MatDev material(const TraceParams &P, int obj) const { return P.mats[obj]; }
};

// Scenes too large for LDS: every wave streams the record array through its own pair of 128-record LDS tiles.
// Each lane loads two records per tile from global memory (coalesced 16-byte loads, served by L2: all waves read
// the same array) two tiles ahead of use, parks them in registers, and writes them to LDS one tile ahead; pass A
// then broadcast-reads them like the staged kind.  No workgroup barrier is involved: a wave only reads what it
// wrote itself, and LDS operations of one wave execute in order.
template <>
struct Fetch<kFetchStream> {
    static constexpr bool kMatrix = false;
    static constexpr bool kTiled = false;
    static constexpr bool kScalarRecords = false;
    const f4v *gc, *gg, *src;
    f4v *tile;  // this wave's 2 x kTileRecords records
    f4v pre0, pre1;
    int lane;
    RT_DEV void start(bool general) {
        src = general ? gg : gc;
        tile[lane] = src[lane];
        tile[64 + lane] = src[64 + lane];
        pre0 = src[kTileRecords + lane];
        pre1 = src[kTileRecords + 64 + lane];
        __builtin_amdgcn_wave_barrier();
    }
    RT_DEV void begin_word(int group) {
        if ((group & (kTileRecords / 4 - 1)) == 0) {
            const int t = group / (kTileRecords / 4);
            f4v *next = tile + ((t + 1) & 1) * kTileRecords;
            next[lane] = pre0;
            next[64 + lane] = pre1;
            const f4v *ahead = src + (t + 2) * kTileRecords;  // the arrays carry two spare tiles at the end
            pre0 = ahead[lane];
            pre1 = ahead[64 + lane];
            __builtin_amdgcn_wave_barrier();
        }
    }
    RT_DEV Rec4 load4(int group) const {
        const f4v *q = tile + ((group / (kTileRecords / 4)) & 1) * kTileRecords + (group & (kTileRecords / 4 - 1)) * 4;
        Rec4 r = {q[0], q[1], q[2], q[3]};
        return r;
    }
    RT_DEV void settle(Rec4 &) const {}
    RT_DEV f4v load1(int k) const { return gg[k]; }
    RT_DEV float radius(int k) const { return ((const float *)(gg + k))[3]; }
    RT_DEV int object_of(const TraceParams &P, int slot) const { return P.sph_obj[slot]; }
    RT_DEV MatDev material(const TraceParams &P, int obj) const { return P.mats[obj]; }
};

template <>
struct Fetch<kFetchMatrix> {  // strict records and lifted A operands staged in LDS; the vector path is the general filter
    static constexpr bool kMatrix = true;
    static constexpr bool kTiled = false;
    static constexpr bool kScalarRecords = false;
    const f4v *pg;
    const h8 *pm;        // [32-sphere group][K half][lane]
    const uint16_t *pobj;  // staged copies of sph_obj and mats, or null when they do not fit beside the rest
    const MatDev *pmat;
    // The staged copies are read through LDS-typed pointers: a load through a pointer that may be LDS or global is
    // a flat load, which has to wait for every outstanding global memory operation of the wave as well as for LDS.
    RT_DEV int object_of(const TraceParams &P, int slot) const {
        if (pobj) return ((const __attribute__((address_space(3))) uint16_t *)pobj)[slot];
        return P.sph_obj[slot];
    }
    RT_DEV MatDev material(const TraceParams &P, int obj) const {
        typedef uint32_t __attribute__((may_alias)) word;
        static_assert(sizeof(MatDev) % sizeof(word) == 0, "MatDev is copied word by word");
        if (pmat) {
            MatDev m;
            const __attribute__((address_space(3))) word *src = (const __attribute__((address_space(3))) word *)(pmat + obj);
            word *dst = (word *)&m;
            for (unsigned i = 0; i < sizeof(MatDev) / sizeof(word); ++i) dst[i] = src[i];
            return m;
        }
        return P.mats[obj];
    }
    RT_DEV void start(bool) {}
    RT_DEV void begin_word(int) {}
    RT_DEV Rec4 load4(int group) const {
        const f4v *q = pg + 4 * group;
        Rec4 r = {q[0], q[1], q[2], q[3]};
        return r;
    }
    RT_DEV void settle(Rec4 &) const {}
    RT_DEV f4v load1(int k) const { return pg[k]; }
    RT_DEV float radius(int k) const { return ((const float *)(pg + k))[3]; }
};

template <>
struct Fetch<kFetchTiled> {  // lifted tiles streamed through two LDS buffers per workgroup (closest_sphere_tiled)
    static constexpr bool kMatrix = true;
    static constexpr bool kTiled = true;
    static constexpr bool kScalarRecords = false;
    f4v *buf;            // LDS: two buffers of kTileVec4
    const f4v *tiles;    // global
    RT_DEV f4v load1(int k) const { return tiles[(size_t)(k / kTileSpheres) * kTileVec4 + kTileLifted + (k % kTileSpheres)]; }
    RT_DEV int object_of(const TraceParams &P, int slot) const { return P.tile_obj[slot]; }
    RT_DEV MatDev material(const TraceParams &P, int obj) const { return P.mats[obj]; }
};

template <>
struct Fetch<kFetchTiledDirect> {  // the lifted tiles read in place (closest_sphere_direct)
    static constexpr bool kMatrix = true;
    static constexpr bool kTiled = true;
    static constexpr bool kScalarRecords = false;
    const f4v *tiles;    // global
    f4v *rays;           // LDS, this wave's: [2][64] (origin, d.d) and (direction, refined reciprocal of d.d) of every lane's ray
    RT_DEV f4v load1(int k) const { return tiles[(size_t)(k / kTileSpheres) * kTileVec4 + kTileLifted + (k % kTileSpheres)]; }
    RT_DEV int object_of(const TraceParams &P, int slot) const { return P.tile_obj[slot]; }
    RT_DEV MatDev material(const TraceParams &P, int obj) const { return P.mats[obj]; }
};
constexpr int kDirectRayVec4 = 128;   // float4 of LDS per wave behind the candidate words of all waves (Fetch<kFetchTiledDirect>::rays)

// The strict records of one tile in LDS, with the interface filter_round expects.
struct TileStrict {
    const f4v *pg;
    RT_DEV void begin_word(int) {}
    RT_DEV Rec4 load4(int group) const {
        const f4v *q = pg + 4 * group;
        Rec4 r = {q[0], q[1], q[2], q[3]};
        return r;
    }
    RT_DEV void settle(Rec4 &) const {}
};

// Optional in-kernel cycle stamps (diagnostic instantiation only): wave-level s_memtime differences summed
// per phase.  With PROF = false every member function is empty and nothing is emitted.
template <bool PROF>
struct Prof {
    // 0 regenerate, 1 filter pass, 2 resolve pass, 3 quads+hit record, 4 shade (cycles); 5 trips, 6 trips of the
    // resolve loop, 7 candidates resolved summed over lanes (counts)
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t last = 0;
    RT_DEV void start() { if constexpr (PROF) last = __builtin_amdgcn_s_memtime(); }
    RT_DEV void lap(int i) {
        if constexpr (PROF) {
            uint64_t now = __builtin_amdgcn_s_memtime();
            t[i] += now - last;
            last = now;
        }
    }
    RT_DEV void count(int i, uint64_t n = 1) { if constexpr (PROF) t[i] += n; }
};

// Executed-work counters of one wave, always on (wave-uniform values: scalar registers): closest-hit searches the wave
// took part in, and 64-sphere blocks it put through the matrix-core filter (8 matrix instructions each).
struct Work {
    uint32_t trips = 0, blocks = 0;  // blocks: in halves (four matrix instructions: one ray tile against 64 spheres)
    uint32_t wide_waves = 0;  // (matrix kind) searches in a row in which more than 32 rays touched the box of the half-width filter
    uint32_t unpaid = 0; // (DENSE kernels) rounds in a row in which resolve_dense_words did not pay, up to 4 (closest_sphere)
    int last_slot = 0;   // per lane (the only member that is): the sphere slot this lane's path hit last (dense rounds
                         // resolve the candidates next to it first)
};

// Per-ray constants of the two filters.
struct RayGeneral { V3 o, dn; };                                          // dn = d * kDirScale / |d|
struct RayCompact { float ndnx, ndny, ndnz, odn, m2ox, m2oy, m2oz, oo1; };  // -dn, o.dn, -2*o, |o|^2 * (1 - kCompactSlack)

// General filter on the strict record (c, r): q = (oc.dn)^2 - (|oc|^2 - r2q), oc = o - c, r2q = r*r * (1 + 2^-17).
// 12 instructions per sphere with the mask update.
RT_DEV float sphere_q(f4v r, const RayGeneral &y) {
    float ocx = y.o.x - r.x, ocy = y.o.y - r.y, ocz = y.o.z - r.z;
    float bn = __builtin_fmaf(ocz, y.dn.z, __builtin_fmaf(ocy, y.dn.y, ocx * y.dn.x));
    float nr2q = (r.w * kRadiusScale) * -r.w;  // -(r*r)(1 + 2^-17) up to 2 ulp, covered by the margin
    float cq = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, nr2q)));
    return __builtin_fmaf(bn, bn, -cq);
}
// Compact filter, record (c, kq): the same quantity expanded so that no per-sphere subtraction is needed,
// bn = o.dn - c.dn and |oc|^2 - r2q - slack = |o|^2(1-k) - 2 o.c + kq.  9 instructions with the mask update.
RT_DEV float sphere_q(f4v r, const RayCompact &y) {
    float cq = __builtin_fmaf(r.z, y.m2oz, __builtin_fmaf(r.y, y.m2oy, __builtin_fmaf(r.x, y.m2ox, r.w + y.oo1)));
    float bn = __builtin_fmaf(r.z, y.ndnz, __builtin_fmaf(r.y, y.ndny, __builtin_fmaf(r.x, y.ndnx, y.odn)));
    return __builtin_fmaf(bn, bn, -cq);
}
// The compact filter with its two accumulation chains (cq and bn) carried as the halves of packed FP32 registers:
// four v_pk_fma_f32 instead of one add and six fused multiply-adds, the same operations in the same order, so the
// same q bit for bit.  It pays where the record sits in scalar registers: a vector instruction with a scalar
// source issues at half rate on gfx950 (4.2 clocks), and so does a packed one (4.3) -- which then does two.
typedef float f2v __attribute__((ext_vector_type(2)));
struct RayCompactPk { f2v one_zero, acc0, cx, cy, cz; };  // (1, 0), (oo1, o.dn), (-2ox, -dnx), (-2oy, -dny), (-2oz, -dnz)
template <bool SCALAR>
RT_DEV float sphere_q_packed(f4v r, const RayCompactPk &y) {
    const f2v lo = {r.x, r.y}, hi = {r.z, r.w};
    f2v t;
    if constexpr (SCALAR) {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(t) : "s"(hi), "v"(y.one_zero), "v"(y.acc0));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(t) : "s"(lo), "v"(y.cx));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(t) : "s"(lo), "v"(y.cy));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(t) : "s"(hi), "v"(y.cz));
    } else {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(t) : "v"(hi), "v"(y.one_zero), "v"(y.acc0));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(t) : "v"(lo), "v"(y.cx));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(t) : "v"(lo), "v"(y.cy));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(t) : "v"(hi), "v"(y.cz));
    }
    return __builtin_fmaf(t.y, t.y, -t.x);
}
RT_DEV float sphere_q(f4v r, const RayCompactPk &y) { return sphere_q_packed<true>(r, y); }
RT_DEV RayCompactPk pack_compact(const RayCompact &c) {
    RayCompactPk p;
    p.one_zero = f2v{1.0f, 0.0f};
    p.acc0 = f2v{c.oo1, c.odn};
    p.cx = f2v{c.m2ox, c.ndnx};
    p.cy = f2v{c.m2oy, c.ndny};
    p.cz = f2v{c.m2oz, c.ndnz};
    return p;
}
// (m << 1) | sign(q): sphere results enter the candidate word from the right.
RT_DEV uint32_t push_sign(uint32_t m, float q) { return __builtin_amdgcn_alignbit(m, __float_as_uint(q), 31); }
template <typename RayT>
RT_DEV uint32_t sphere_bit(uint32_t m, f4v r, const RayT &y) { return push_sign(m, sphere_q(r, y)); }
// Four spheres: the four q chains are independent and are written side by side so that the scheduler can
// interleave them; only the four mask updates are serial.
template <typename RayT>
RT_DEV uint32_t group_bits(uint32_t m, const Rec4 &g, const RayT &y) {
    float q0 = sphere_q(g.r0, y), q1 = sphere_q(g.r1, y), q2 = sphere_q(g.r2, y), q3 = sphere_q(g.r3, y);
    return push_sign(push_sign(push_sign(push_sign(m, q0), q1), q2), q3);
}

// Pass A over sphere groups [g0, g1): writes one candidate word per 32 spheres to masks[word * 64 + lane]
// (sphere j of the word at bit 31 - j) and returns the bitmap of non-empty words.
template <typename F, typename RayT>
RT_DEV uint32_t filter_round(F &f, int g0, int g1, int nsph, uint32_t *masks, int lane, const RayT &y) {
    uint32_t nz = 0;
    int g = g0;
    f.begin_word(g);
    Rec4 A = f.load4(g);
    for (int w = 0; g < g1; ++w) {
        if (w) f.begin_word(g);
        const int ng = min(kWordGroups, g1 - g);
        uint32_t m = 0;
        if (ng == kWordGroups) {
#pragma unroll
            for (int i = 0; i < kWordGroups; i += 2) {
                f.settle(A);
                Rec4 B = f.load4(g + i + 1);
                m = group_bits(m, A, y);
                f.settle(B);
                A = f.load4(g + i + 2);  // the record arrays carry one spare group at the end
                m = group_bits(m, B, y);
            }
        } else {
            for (int i = 0; i < ng; ++i) {
                f.settle(A);
                Rec4 B = f.load4(g + i + 1);
                m = group_bits(m, A, y);
                A = B;
            }
        }
        // keep only real spheres: the last group may hold up to three padding slots
        const int nvalid = nsph - 4 * g;
        const uint32_t valid = nvalid >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> nvalid);
        g += ng;
        const uint32_t cand = ((~m) << (32 - 4 * ng)) & valid;
        masks[w * 64 + lane] = cand;
        nz |= (cand != 0u ? 1u : 0u) << w;
    }
    return nz;
}

// ---- matrix-core filter (rt_lift.hpp) -----------------------------------------------------------------------
// B operands of the wave's 64 rays: [ray half][K half], lane l supplying column l % 32, K slots 8 (l / 32) .. + 7.
struct RayLifted { h8 b00, b01, b10, b11; };

// v0 keeps its lanes 0-31 and receives v1's lanes 0-31 in its lanes 32-63; v1 the other way round.
RT_DEV void swap_halves(uint32_t &v0, uint32_t &v1) {
    auto r = __builtin_amdgcn_permlane32_swap(v0, v1, false, false);
    v0 = r[0];
    v1 = r[1];
}
RT_DEV void swap_halves(h8 &x0, h8 &x1) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 a = __builtin_bit_cast(u4, x0), b = __builtin_bit_cast(u4, x1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t p = a[i], q = b[i];
        swap_halves(p, q);
        a[i] = p;
        b[i] = q;
    }
    x0 = __builtin_bit_cast(h8, a);
    x1 = __builtin_bit_cast(h8, b);
}

// The ten ray-side values -> FP16 pairs in slot order: every lane the 32 slots of its own ray, x[i] = slots 8i .. 8i + 7.
RT_DEV void pack_lifted_slots(float (&v)[10], h8 (&x)[4]) {
    // The FP16 pairs must satisfy hi + lo = v to 2^-22: pin each v as a binary32 value so that both uses of its
    // conversion round the same number (otherwise the compiler converts one of them straight from the unrounded
    // fused product, and the two halves of the pair can then disagree by a whole FP16 ulp).
#pragma unroll
    for (int i = 0; i < 10; ++i) asm volatile("" : "+v"(v[i]));
    _Float16 r[kLiftSlots];
    _Float16 hi = (_Float16)v[0];
    r[0] = hi;
    r[1] = (_Float16)(v[0] - (float)hi);
    r[2] = r[3] = r[4] = (_Float16)1.0f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        hi = (_Float16)v[1 + t];
        r[5 + 3 * t] = hi;
        r[6 + 3 * t] = hi;
        r[7 + 3 * t] = (_Float16)(v[1 + t] - (float)hi);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[0][i] = r[i]; x[1][i] = r[8 + i]; x[2][i] = r[16 + i]; x[3][i] = r[24 + i]; }
}
// Per-lane slot vectors -> the four B operands of the wave's 64 rays (the two halves of the wave exchange them).
RT_DEV void exchange_lifted(h8 (&x)[4], RayLifted &L) {
    swap_halves(x[0], x[1]);  // -> K half 0 of ray halves 0 and 1
    swap_halves(x[2], x[3]);  // -> K half 1
    L.b00 = x[0]; L.b10 = x[1];
    L.b01 = x[2]; L.b11 = x[3];
}
RT_DEV void pack_lifted(float (&v)[10], RayLifted &L) {
    h8 x[4];
    pack_lifted_slots(v, x);
    exchange_lifted(x, L);
}

// Ray side of the lifted form, every lane for its own ray: the slot vectors (lift_ray_slots), or those already
// exchanged into the four B operands (lift_ray).  oo is |o'|^2 in the scaled frame (the caller compares it with the
// scene's limit).
RT_DEV void lift_ray_slots(const LiftFrame &F, V3 o, V3 dn, float slack, float &oo, h8 (&x)[4]) {
    const float ox = (o.x - F.tx) * F.sigma, oy = (o.y - F.ty) * F.sigma, oz = (o.z - F.tz) * F.sigma;
    oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float odn = __builtin_fmaf(oz, dn.z, __builtin_fmaf(oy, dn.y, ox * dn.x));
    float v[10];
    v[0] = __builtin_fmaf(odn, odn, -(oo * (1.0f - slack)));
    v[1] = 2.0f * __builtin_fmaf(-odn, dn.x, ox);
    v[2] = 2.0f * __builtin_fmaf(-odn, dn.y, oy);
    v[3] = 2.0f * __builtin_fmaf(-odn, dn.z, oz);
    v[4] = dn.x * dn.x; v[5] = dn.y * dn.y; v[6] = dn.z * dn.z;
    v[7] = dn.x * dn.y; v[8] = dn.x * dn.z; v[9] = dn.y * dn.z;
    pack_lifted_slots(v, x);
}
RT_DEV void lift_ray(const LiftFrame &F, V3 o, V3 dn, float slack, float &oo, RayLifted &L) {
    h8 x[4];
    lift_ray_slots(F, o, dn, slack, oo, x);
    exchange_lifted(x, L);
}

// The same for a frame that is local to a tile of spheres while the ray may start far away (tiled kind).  The
// filter tests the ray's LINE, so any point of it can stand in for the origin: p = o' - (o'.dn) dn, the point next
// to the frame's centre, is used, and the relative slack k (|p|^2 + |c'|^2 + r^2) then grows with the distance
// between line and tile centre instead of with the distance of the ray's origin.  Two things still grow with that
// distance, and `extra`, added to the ray's constant term, covers them (A = |o'| + cmax >= |o - c| for every sphere
// of the tile, cmax >= |c'|, u = 2^-24):
//   the reference's own rounding: its discriminant can be >= 0 while the line misses by 17u |o - c|^2 (see
//     closest_sphere) -- no conservative filter can do with less; covered by kNearSlack A^2 = 64u A^2;
//   the displacement of the computed line near the tile: <= 1.75u (|o'| + |p|) from the roundings of o - T and of
//     p's fused update (an error in the dot product only moves p along the line), <= 1.75u |o'| from the rounding
//     of the direction, together <= 5.25u |o'|, which moves q by at most 2 |p - c'| times that = 10.5u |o'|
//     (|p| + cmax); covered by 64u A (|p| + cmax).
// At 140 units from a tile of r = 0.1 spheres that is r_eff = 0.30 where the origin-based slack gave 0.55 and no
// tile-local relative slack at all.
// oo returns |p|^2 for the range check; a ray whose constant term leaves the FP16 range reports oo = inf.
constexpr float kNearSlack = 0x1p-18f;  // 64u
RT_DEV void lift_ray_near_slots(const LiftFrame &F, float cmax, V3 o, V3 dn, float slack, float near_slack, float &oo, h8 (&x)[4]) {
    const float fx = (o.x - F.tx) * F.sigma, fy = (o.y - F.ty) * F.sigma, fz = (o.z - F.tz) * F.sigma;
    const float far2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
    const float t0 = __builtin_fmaf(fz, dn.z, __builtin_fmaf(fy, dn.y, fx * dn.x));
    const float ox = __builtin_fmaf(-t0, dn.x, fx), oy = __builtin_fmaf(-t0, dn.y, fy), oz = __builtin_fmaf(-t0, dn.z, fz);
    const float pp = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float odn = __builtin_fmaf(oz, dn.z, __builtin_fmaf(oy, dn.y, ox * dn.x));
    const float A = __builtin_sqrtf(far2) * (1.0f + 0x1p-20f) + cmax;
    const float extra = near_slack * A * (A + __builtin_sqrtf(pp) * (1.0f + 0x1p-20f) + cmax);
    float v[10];
    v[0] = __builtin_fmaf(odn, odn, -(pp * (1.0f - slack))) + extra;
    v[1] = 2.0f * __builtin_fmaf(-odn, dn.x, ox);
    v[2] = 2.0f * __builtin_fmaf(-odn, dn.y, oy);
    v[3] = 2.0f * __builtin_fmaf(-odn, dn.z, oz);
    v[4] = dn.x * dn.x; v[5] = dn.y * dn.y; v[6] = dn.z * dn.z;
    v[7] = dn.x * dn.y; v[8] = dn.x * dn.z; v[9] = dn.y * dn.z;
    oo = __builtin_fabsf(v[0]) <= 57000.0f ? pp : __builtin_inff();   // NaN compares false: inf
    pack_lifted_slots(v, x);
}
RT_DEV void lift_ray_near(const LiftFrame &F, float cmax, V3 o, V3 dn, float slack, float near_slack, float &oo, RayLifted &L) {
    h8 x[4];
    lift_ray_near_slots(F, cmax, o, dn, slack, near_slack, oo, x);
    exchange_lifted(x, L);
}

// Pass A on the matrix cores over 64-sphere blocks [blk0, blk1): same candidate words as filter_round.
//
// The four matrix instructions of a 32-sphere group are issued back to back and fenced off from the vector code
// on both sides with scheduling barriers, and the results are read only after the whole group has been issued, in
// ascending register order.  The reason: MI355X does not interlock a vector read of a matrix result.  The 16 result
// registers of v_mfma_f32_32x32x16_f16 are written four at a time, register r being safe to read 5 + ceil(r / 2)
// wait states after issue (5 for r = 0, 12 for r = 13..15); an earlier read silently returns the old contents
// (tools/mfma_hazard_probe.hip).  The ROCm 7.2 compiler spaces such reads by 11 wait states, one short for the last
// three registers unless another wave stretches the distance -- which showed as counts of rt_selftest_matrix_filter
// that drifted from run to run when the compiler was left to weave the sign extraction of one group between the
// matrix instructions of the next.  With the group fenced and read in order every read is at least 6 wait states
// beyond the requirement; tools/check_mfma_distance.py proves that on the compiled ISA of every kernel (it runs in
// the CPU tests).  -DRT_NO_MFMA_FENCE=0 builds without the barriers.
RT_DEV void matrix_group(const h8 a_lo, const h8 a_hi, const RayLifted &L, f32x16 &q0, f32x16 &q1) {
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#if !defined(RT_NO_MFMA_FENCE) || (RT_NO_MFMA_FENCE & 1)
    __builtin_amdgcn_sched_barrier(0);
#endif
    q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, L.b00, zero, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, L.b10, zero, 0, 0, 0);
    q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, L.b01, q0, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, L.b11, q1, 0, 0, 0);
#if !defined(RT_NO_MFMA_FENCE) || (RT_NO_MFMA_FENCE & 2)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// SPARSE: for scenes of thousands of spheres, where a ray has a candidate in few of the blocks.  Writing the sign bits
// of a 32-sphere group into the words costs 32 half-rate instructions; finding out whether any of the wave's 64 rays
// has a candidate in the group at all costs 31 full-rate ones and a vote, after which the 32 are skipped when nobody
// has.  That pays once more than about two thirds of the groups are skipped.
RT_DEV bool group_has_candidate(const f32x16 &q0, const f32x16 &q1) {
    // the AND of the 32 bit patterns keeps the sign bit only if every result has it; four independent chains of
    // v_and_b32 (a full-rate instruction; v_max3_f32 was measured at about 7 clocks, twice v_max_f32)
    uint32_t m[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x16 &q = c < 2 ? q0 : q1;
        const int o = (c & 1) * 8;
        m[c] = __float_as_uint(q[o]);
#pragma unroll
        for (int k = 1; k < 8; ++k) m[c] &= __float_as_uint(q[o + k]);
    }
    const uint32_t all = (m[0] & m[1]) & (m[2] & m[3]);
    return __builtin_amdgcn_ballot_w64((int)all >= 0) != 0;  // some lane holds a result without the sign bit
}

template <bool SPARSE = false>
RT_DEV uint32_t filter_round_matrix(const h8 *pm, int blk0, int blk1, uint32_t *masks, int lane, const RayLifted &L) {
    uint32_t nz = 0;
    const h8 *A = pm + (size_t)blk0 * 256 + lane;
    for (int blk = blk0, w = 0; blk < blk1; ++blk, w += 2, A += 256) {
        f32x16 q0, q1;
        uint32_t w0 = 0, w1 = 0;
        h8 a_lo = A[0], a_hi = A[64];
        matrix_group(a_lo, a_hi, L, q0, q1);
        a_lo = A[128];  // issued behind the group's barrier: the operands of one group at a time hold registers
        a_hi = A[192];
        if (!SPARSE || group_has_candidate(q0, q1)) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { w0 = push_sign(w0, q0[k]); w1 = push_sign(w1, q1[k]); }
        } else {
            w0 = 0xFFFFu;
            w1 = 0xFFFFu;
        }
        matrix_group(a_lo, a_hi, L, q0, q1);
        if (!SPARSE || group_has_candidate(q0, q1)) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { w0 = push_sign(w0, q0[k]); w1 = push_sign(w1, q1[k]); }
        } else {
            w0 = (w0 << 16) | 0xFFFFu;
            w1 = (w1 << 16) | 0xFFFFu;
        }
        swap_halves(w0, w1);
        const uint32_t c0 = ~w0, c1 = ~w1;
        masks[w * 64 + lane] = c0;
        masks[(w + 1) * 64 + lane] = c1;
        // (unsigned minimum and shift-or: four instructions where the compiler's comparisons and selects take six, at
        // half rate; it turns a plain min(c, 1) back into those, hence the statement)
        uint32_t b0, b1;
        asm("v_min_u32 %0, 1, %2\n\tv_min_u32 %1, 1, %3" : "=&v"(b0), "=v"(b1) : "v"(c0), "v"(c1));
        nz |= ((b1 << 1) | b0) << w;
    }
    return nz;
}

// ---- the matrix-core filter at half width ---------------------------------------------------------------------
// Most rays of a scene whose small spheres lie in a thin layer (the cover scene: a field of spheres of radius 0.2 on
// the ground) pass that layer by: their line misses the bounding box of those spheres, so no sphere inside it can be
// a candidate, and the filter has nothing to find out about them.  When at most 32 lanes of a wave hold a ray that
// does touch the box, those rays are moved into the 32 columns of ONE ray tile -- lane l and lane l + 32 supply K
// slots 0-7 and 8-15 of the ray in column l % 32 -- and a 64-sphere block costs four matrix instructions and 32
// sign extractions instead of eight and 64.  The spheres the box leaves out (few and large, or not representable in
// the lifted form; the host lists them) are tested for every ray by the general vector filter afterwards.
//
// Conservativeness.  Whenever the reference's discriminant is >= 0 the ray's line passes within
// sqrt(r^2 + 17u |o - c|^2) of the centre (closest_sphere), and the point of an accepted root lies within
// sqrt(r^2 + 34u |o - c|^2) of it; the host grows every sphere's radius to
// sqrt(r^2 + 64u D^2) with D >= |o - c| for every origin this path admits (TraceParams::narrow_oo) before it takes the
// box, and pads the box for the roundings of the test itself (rt_api.hip narrow_setup).  The test declares a miss only
// on a separating axis of line and box (the three cross products of the direction with the box axes) or when the
// origin lies outside a slab and the ray does not move towards it; every comparison is false on a NaN operand, so
// such a ray counts as touching.

// May the ray (o, d), t >= 0, touch the box with centre m and half extents h?
RT_DEV bool ray_may_touch_box(V3 o, V3 d, V3 m, V3 h) {
    const float ex = o.x - m.x, ey = o.y - m.y, ez = o.z - m.z;
    const float adx = __builtin_fabsf(d.x), ady = __builtin_fabsf(d.y), adz = __builtin_fabsf(d.z);
    // separating axes d x e_i
    bool miss = __builtin_fabsf(__builtin_fmaf(ey, d.z, -(ez * d.y))) > __builtin_fmaf(h.y, adz, h.z * ady);
    miss = miss || __builtin_fabsf(__builtin_fmaf(ez, d.x, -(ex * d.z))) > __builtin_fmaf(h.z, adx, h.x * adz);
    miss = miss || __builtin_fabsf(__builtin_fmaf(ex, d.y, -(ey * d.x))) > __builtin_fmaf(h.x, ady, h.y * adx);
    // outside a slab and not moving towards it
    miss = miss || (__builtin_fabsf(ex) > h.x && ex * d.x >= 0.0f);
    miss = miss || (__builtin_fabsf(ey) > h.y && ey * d.y >= 0.0f);
    miss = miss || (__builtin_fabsf(ez) > h.z && ez * d.z >= 0.0f);
    return !miss;
}

// The same for a box the ray may start any distance from (the tiled kinds): bm = (centre, length of the half diagonal),
// bh = half extents, already widened by 1 + 2^-12 on the host.  The allowances are taken per ray from
// D >= |o - centre| + half diagonal, which bounds |o - c| for every sphere centre c in the box: a radius grows to
// sqrt(r^2 + 64u D^2) <= r + 2^-9 D, and the test's own roundings are within 2^-16 D.  (The distance is taken as the sum
// of the three absolute differences, which is never below the Euclidean one and costs two additions.)  scale (self-test
// only) scales both allowances.
RT_DEV bool ray_may_touch_far_box(V3 o, V3 d, f4v bm, f4v bh, float scale = 1.0f) {
    const float ex = o.x - bm.x, ey = o.y - bm.y, ez = o.z - bm.z;
    const float dist = __builtin_fmaf(__builtin_fabsf(ex) + __builtin_fabsf(ey) + __builtin_fabsf(ez), 1.0f + 0x1p-20f, bm.w);
    const float pad = dist * ((0x1p-9f * (1.0f + 0x1p-6f) + 0x1p-16f) * scale);
    return ray_may_touch_box(o, d, v3(bm.x, bm.y, bm.z), v3(bh.x + pad, bh.y + pad, bh.z + pad));
}

// Two 32-sphere groups against one ray tile: q0 = group 0, q1 = group 1 (fenced and read like matrix_group).
RT_DEV void matrix_pair(const h8 a0_lo, const h8 a0_hi, const h8 a1_lo, const h8 a1_hi, const h8 b_lo, const h8 b_hi, f32x16 &q0,
                        f32x16 &q1) {
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#if !defined(RT_NO_MFMA_FENCE) || (RT_NO_MFMA_FENCE & 1)
    __builtin_amdgcn_sched_barrier(0);
#endif
    q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0_lo, b_lo, zero, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1_lo, b_lo, zero, 0, 0, 0);
    q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0_hi, b_hi, q0, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1_hi, b_hi, q1, 0, 0, 0);
#if !defined(RT_NO_MFMA_FENCE) || (RT_NO_MFMA_FENCE & 2)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// Pass A over 64-sphere blocks [blk0, blk1) for the rays of one tile: b_lo / b_hi are this lane's B operands (column
// lane % 32 of the tile, K slots 8 (lane / 32) .. + 7 of each K half), `owner` the lane whose ray the column holds.
// Lane half h ends up with word 2 blk + h of its column's ray -- the rows a lane half receives are the spheres of one
// word (lift_index) -- and writes it straight into the owner's place.  Returns the bitmap of the non-empty words this
// lane wrote (the owner needs the OR over both halves of its column).
// AHEAD: the sphere operands come from global memory; those of the next block are asked for before this block's matrix
// instructions are issued.
template <bool SPARSE = false, bool AHEAD = false>
RT_DEV uint32_t filter_round_narrow(const h8 *pm, int blk0, int blk1, uint32_t *masks, int lane, const h8 b_lo, const h8 b_hi, int owner,
                                    bool column_used) {
    uint32_t nz = 0;
    const h8 *A = pm + (size_t)blk0 * 256 + lane;
    const int h = lane >> 5;
    [[maybe_unused]] h8 n0, n1, n2, n3;
    if constexpr (AHEAD) { n0 = A[0]; n1 = A[64]; n2 = A[128]; n3 = A[192]; }
    for (int blk = blk0, w = 0; blk < blk1; ++blk, w += 2, A += 256) {
        f32x16 q0, q1;
        uint32_t word = 0xFFFFFFFFu;
        if constexpr (AHEAD) {
            const h8 a0 = n0, a1 = n1, a2 = n2, a3 = n3;
            if (blk + 1 < blk1) { n0 = A[256]; n1 = A[320]; n2 = A[384]; n3 = A[448]; }
            matrix_pair(a0, a1, a2, a3, b_lo, b_hi, q0, q1);
        } else {
            matrix_pair(A[0], A[64], A[128], A[192], b_lo, b_hi, q0, q1);
        }
        if (!SPARSE || group_has_candidate(q0, q1)) {   // (see filter_round_matrix)
#pragma unroll
            for (int k = 0; k < 16; ++k) word = push_sign(word, q0[k]);
#pragma unroll
            for (int k = 0; k < 16; ++k) word = push_sign(word, q1[k]);
        }
        const uint32_t c = ~word;
        if (column_used) masks[(w + h) * 64 + owner] = c;
        uint32_t b;
        asm("v_min_u32 %0, 1, %1" : "=v"(b) : "v"(c));
        nz |= b << (w + h);
    }
    return column_used ? nz : 0u;
}

// The rays of the lanes in `touching` (at most 32) -> the columns of one ray tile, through the wave's candidate-word area,
// which nothing uses at that time: a lane with such a ray writes its four slot octets to row `rank` (its number among those
// lanes) of four arrays and its lane number to the list; lanes l and l + 32 then read the K slots they supply for the ray of row l % 32
// and that ray's lane.  never_unused: columns without a ray get operands that make every real sphere's q negative (for
// the form of the filter that first asks whether a group has any candidate at all).
RT_DEV void gather_columns(uint32_t *wave_masks, int lane, uint64_t touching, const h8 (&slots)[4], bool never_unused, h8 &b_lo, h8 &b_hi,
                           int &owner) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 *rows = (u4 *)wave_masks;
    uint32_t *list = wave_masks + kListOffset;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(touching >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)touching, 0u));
    if ((touching >> lane) & 1ull) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rows[(uint32_t)i * 32u + rank] = __builtin_bit_cast(u4, slots[i]);   // [slot octet][row]: no bank conflicts
        list[rank] = (uint32_t)lane;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int col = lane & 31, half = lane >> 5;
    b_lo = __builtin_bit_cast(h8, rows[half * 32 + col]);
    b_hi = __builtin_bit_cast(h8, rows[(2 + half) * 32 + col]);
    owner = (int)list[col];
    if (never_unused && col >= __builtin_popcountll(touching)) {
        // slots 0 and 1 (x 1 on the sphere side) = -60000 each, slot 2 (x the leading part of -kq, at most 60000) = 1
        const h8 never = {(_Float16)-60000.0f, (_Float16)-60000.0f, (_Float16)1.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
        const h8 zero = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
        b_lo = half == 0 ? never : zero;
        b_hi = zero;
    }
    // (the words are written only after these reads have returned: LDS operations of a wave execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Non-empty-word bitmaps of filter_round_narrow -> the lanes the rays belong to.
RT_DEV uint32_t scatter_bitmaps(uint32_t *wave_masks, int lane, uint64_t touching, uint32_t written) {
    uint32_t *list = wave_masks + kListOffset;
    list[lane] = written;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(touching >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)touching, 0u));
    return (touching >> lane) & 1ull ? list[rank] | list[rank + 32u] : 0u;
}

// Pass B for slots that are in object order, balanced over the wave.  Lanes differ in how many candidates pass A left
// them (the cover scene: 2.2 on average, a few lanes of a wave four or five), and a loop in which every lane resolves
// its own runs as long as the busiest lane.  Here the wave's candidates are pooled instead: layer by layer (every
// lane's first candidate, then every lane's second, ...) they are appended to a list of 64 in LDS; whenever the next
// layer would not fit, lane j resolves entry j -- it fetches the owner's ray with cross-lane reads, evaluates the
// reference's arithmetic exactly as the owner would, and offers (distance, slot) to the owner's 64-bit minimum in
// LDS, distance in the upper half.  Distances are positive floats, whose bit patterns order like unsigned integers,
// so that minimum is the reference's answer: the smallest accepted root and, among equal roots, the lowest index.
// (The reference tests a sphere against the best distance so far, Shaders.metal:17-22; a sphere's root that would
// pass that test is the same root it offers against an unbounded interval, because its second root is never
// smaller than its first.)
template <typename F, bool PROF>
RT_DEV void resolve_balanced(const F &f, uint32_t *wave_masks, uint32_t nz, int slot0, int lane, V3 o, V3 d, const RootDiv &dv,
                             Prof<PROF> &prof) {
    uint32_t *list = wave_masks + kListOffset;
    unsigned long long *best = (unsigned long long *)(wave_masks + kBestOffset);
    auto flush = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t e = list[lane];
        const int owner = lane < n ? (int)(e >> 16) : lane, k = (int)(e & 0xFFFFu);
        // the owner's ray (every lane takes part in the cross-lane reads)
        const float ox = __shfl(o.x, owner, 64), oy = __shfl(o.y, owner, 64), oz = __shfl(o.z, owner, 64);
        const float dx = __shfl(d.x, owner, 64), dy = __shfl(d.y, owner, 64), dz = __shfl(d.z, owner, 64);
        RootDiv q;
        q.a = __shfl(dv.a, owner, 64);
        q.r = __shfl(dv.r, owner, 64);
        q.nmin = q.a * 0.0009f;
        q.slow = dv.slow;
        if (lane < n) {
            if constexpr (PROF) prof.count(7, (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)));
            // Lib.metal:216-238 as written
            f4v rec = f.load1(slot0 + k);  // (center, radius)
            float ocx = ox - rec.x, ocy = oy - rec.y, ocz = oz - rec.z;
            float b = ocx * dx + ocy * dy + ocz * dz;
            float c = (ocx * ocx + ocy * ocy + ocz * ocz) - rec.w * rec.w;
            float disc = b * b - q.a * c;
            float root, nbest = __builtin_inff();
            if (sphere_root(disc, b, q, kMaxFloat, nbest, root))
                atomicMin(&best[owner], ((unsigned long long)__float_as_uint(root) << 32) | (uint32_t)(slot0 + k));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    uint32_t m = 0;
    int w = 0, filled = 0;
    for (;;) {
        const bool has = m != 0 || nz != 0;
        const uint64_t layer = __builtin_amdgcn_ballot_w64(has);
        const int n = __builtin_popcountll(layer);
        if (filled != 0 && (n == 0 || filled + n > 64)) {
            prof.count(6);
            flush(filled);
            filled = 0;
        }
        if (n == 0) break;
        if (has) {
            if (m == 0) {
                w = __builtin_ctz(nz);
                nz &= nz - 1;
                m = wave_masks[w * 64 + lane];
            }
            const int j = __builtin_clz(m);
            m &= ~(0x80000000u >> j);
            const uint32_t pos = (uint32_t)filled + __builtin_amdgcn_mbcnt_hi((uint32_t)(layer >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)layer, 0u));
            list[pos] = ((uint32_t)lane << 16) | (uint32_t)(w * 32 + j);
        }
        filled += n;
    }
}

// Pass B for slots that are in object order, every lane resolving its own candidates.  It is the form for waves whose
// lanes all hold long lists (rays that cross many nested shells: 170 candidates per ray), where pooling has nothing
// to balance and pays for its list, its cross-lane reads and its LDS minima.  What limits a per-lane loop is the
// length of one candidate's dependent chain (record read, discriminant, square root, division), so four candidates
// are in flight at a time -- the two highest and the two lowest set bits of the current word -- and their code has
// no branches: every lane runs the same instructions anyway.  Four at once is possible because the root a sphere
// offers does not depend on the best distance so far (see resolve_balanced), so the lane's answer is the minimum
// over (distance, slot) in any visiting order; the running best only serves to discard numerators clearly above its
// own (LaneBest::prune_above: an equal distance at a lower slot must still be seen, and it can come from a numerator
// a few ulps larger).  When
// a word has fewer than four bits left some chains resolve the same candidate twice, which changes nothing.
struct LaneBest {
    float t;        // smallest accepted root so far (tbest on entry)
    uint32_t slot;  // its slot, 0xFFFFFFFF while none
    float n;        // its numerator (root * a up to rounding); +inf while none was found in this call
    // Numerator up to which a candidate is still looked at.  Two numerators a few ulps apart can give the same
    // correctly rounded quotient by a, and an equal root at a lower slot must win (Shaders.metal:21 keeps the first of
    // equal distances), so the cut is not best.n itself but best.n (1 + 2^-22): four ulps, from where the quotients
    // by the same a can no longer coincide (the same allowance as sphere_root_any_order's nskip).
    RT_DEV float prune_above() const { return n * (1.0f + 0x1p-22f); }
};
struct LaneRoot {
    bool ok, retry;  // retry: the first root exists but is not above kTMin, so the second has to be looked at
    float root, n, n2;
};
// Lib.metal:216-238 for one candidate without branches.  The root the reference would take is the first one if it is
// above kTMin, else the second if that is; numerators at or below dv.nmin give quotients below kTMin without
// dividing (RootDiv).  A negative discriminant makes every numerator NaN, and a NaN passes no comparison.
RT_DEV LaneRoot lane_root(f4v rec, V3 o, V3 d, const RootDiv &dv, float nprune) {  // rec = (center, radius)
    float ocx = o.x - rec.x, ocy = o.y - rec.y, ocz = o.z - rec.z;
    float b = ocx * d.x + ocy * d.y + ocz * d.z;
    float c = (ocx * ocx + ocy * ocy + ocz * ocz) - rec.w * rec.w;
    float disc = b * b - dv.a * c;
    float sqrtd = __builtin_sqrtf(disc);
    LaneRoot r;
    const float n1 = -b - sqrtd;
    r.n2 = -b + sqrtd;
    const bool first = n1 > dv.nmin;
    r.n = first ? n1 : r.n2;
    r.root = dv.quotient_fast(r.n);
    const bool above = kTMin < r.root;
    // (bitwise operators: no short-circuit branches, the four candidates of a turn stay in one basic block)
    r.ok = (r.n > dv.nmin) & (r.n <= nprune) & above & !(disc < 0.0f);
    r.retry = first & !above & !(disc < 0.0f);
    return r;
}
RT_DEV void lane_offer(LaneBest &best, const LaneRoot &r, uint32_t slot) {
    const bool better = r.ok & ((r.root < best.t) | ((r.root == best.t) & (slot < best.slot)));
    best.t = better ? r.root : best.t;
    best.slot = better ? slot : best.slot;
    best.n = better ? r.n : best.n;
}
// The rare second look: the first root was a hair below kTMin although its numerator was above dv.nmin.
RT_DEV void lane_retry(LaneBest &best, const LaneRoot &r, uint32_t slot, const RootDiv &dv, float nprune) {
    if (r.retry && r.n2 > dv.nmin && r.n2 <= nprune) {
        const float root = dv.quotient_fast(r.n2);
        if (kTMin < root && (root < best.t || (root == best.t && slot < best.slot))) {
            best.t = root;
            best.slot = slot;
            best.n = r.n2;
        }
    }
}
// One candidate word (32 spheres from slot `base` on) of every lane, four candidates per turn.
template <typename F, bool PROF>
RT_DEV void resolve_word_per_lane(const F &f, uint32_t m, int base, V3 o, V3 d, const RootDiv &dv, LaneBest &best, Prof<PROF> &prof) {
    while (m) {
        // positions in the word (0 = first sphere): the two highest and the two lowest set bits
        const int p0 = __builtin_clz(m), p3 = 31 - __builtin_ctz(m);
        const uint32_t m1 = m & ~((0x80000000u >> p0) | (0x80000000u >> p3));
        const int p1 = m1 ? __builtin_clz(m1) : p0, p2 = m1 ? 31 - __builtin_ctz(m1) : p3;
        m = m1 & ~((0x80000000u >> p1) | (0x80000000u >> p2));
        if constexpr (PROF) {
            prof.count(6);
            prof.count(7, (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) * 4);
        }
        const float nprune = best.prune_above();
        // all four records are asked for before the first is used (the statement below keeps the reads together)
        f4v c0 = f.load1(base + p0), c1 = f.load1(base + p1), c2 = f.load1(base + p2), c3 = f.load1(base + p3);
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
        const LaneRoot r0 = lane_root(c0, o, d, dv, nprune);
        const LaneRoot r1 = lane_root(c1, o, d, dv, nprune);
        const LaneRoot r2 = lane_root(c2, o, d, dv, nprune);
        const LaneRoot r3 = lane_root(c3, o, d, dv, nprune);
        lane_offer(best, r0, (uint32_t)(base + p0));
        lane_offer(best, r1, (uint32_t)(base + p1));
        lane_offer(best, r2, (uint32_t)(base + p2));
        lane_offer(best, r3, (uint32_t)(base + p3));
        if (__builtin_amdgcn_ballot_w64(r0.retry || r1.retry || r2.retry || r3.retry)) {
            lane_retry(best, r0, (uint32_t)(base + p0), dv, nprune);
            lane_retry(best, r1, (uint32_t)(base + p1), dv, nprune);
            lane_retry(best, r2, (uint32_t)(base + p2), dv, nprune);
            lane_retry(best, r3, (uint32_t)(base + p3), dv, nprune);
        }
    }
}

// Prune of candidates that cannot win against the lane's best accepted root t so far.  With u = o - c, the function
// g(s) = |u + s d|^2 - r^2 = a s^2 + 2 (u.d) s + (|u|^2 - r^2) is convex and its zeros are the sphere's roots, so the
// sphere has no root in [0, t] -- and therefore no root the reference would accept that is not above t -- when
//   g(0) < 0 and g(t) < 0 (o and P = o + t d inside: the only positive root is the exit, beyond P), or
//   g(0) > 0 and u.d > 0 (o outside, centre behind o: both roots negative), or
//   g(0) > 0, g(t) > 0 and u.d + t a < 0 (o and P outside, closest approach beyond P: the ray enters beyond P).
// The same three tests prune a whole word of 32 spheres when "inside" is taken against a ball that every sphere of
// the word contains and "outside" against a ball that contains them all (TraceParams::word_bounds).
// Every comparison is made with a margin M = m X + 2^-21 |o|^2, X = |u|^2 + 2 t |u.d| + t^2 a >= max(|u|^2,
// |P - c|^2), that is far above the rounding of g (2^-21 X) and keeps the true roots away from [0, t] by more than
// the reference's computed root can be off.  One sphere: m = 2^-15 and the radius^2 is widened by 2^-10 either way;
// then |g(t)| > M puts the exit 2^-11 r / |d| beyond t (the reference's exit root is within 2^-21 r / |d|) and a
// grazing entry 2^-8.8 sqrt(X) / |d| beyond t (the reference's first root of a near-zero discriminant is within
// 2^-10.2 sqrt(X) / |d|; away from grazing both bounds grow apart); in the second case the reference's numerators
// are not positive whatever the rounding, given the signs of its b and c.  A word: the balls are widened by 2^-6 and
// m = 2^-8 in the comparisons against the outer ball, which keeps every root of every sphere of the word
// 2^-9 sqrt(X) / |d| outside [0, t]; against the inner ball m = 2^-15 is enough, because a ray that starts and ends
// 2^-7 of the inner radius inside every sphere grazes none of them (every exit is 2^-13 r / |d| beyond t).  The signs of
// u.d and u.d + t a are taken only when clear of their rounding (2^-19 and 2^-18 of |u||d|, sqrt(X)|d|).  NaN or
// infinite operands pass no comparison.  has_t = false (no accepted root yet): only the second case applies.
struct PruneRay {
    V3 o, d;
    float k1, a, a38, a36, t, t2, t2a;  // 2^-21 |o|^2, d.d, 2^-38 a, 2^-36 a, best root t, 2 t, t^2 a
    bool has_t;
};
RT_DEV PruneRay make_prune_ray(V3 o, V3 d, float a, bool has_t, float t) {
    PruneRay y;
    y.o = o;
    y.d = d;
    y.k1 = 0x1p-21f * __builtin_fmaf(o.z, o.z, __builtin_fmaf(o.y, o.y, o.x * o.x));
    y.a = a;
    y.a38 = 0x1p-38f * a;
    y.a36 = 0x1p-36f * a;
    y.has_t = has_t;
    y.t = has_t ? t : 0.0f;
    y.t2 = 2.0f * y.t;
    y.t2a = y.t * y.t * a;
    return y;
}
// The ray against one centre: everything in the three tests but the radii (groups of one scene often share a centre).
struct PruneAbout {
    float so, sp, X;      // |o - c|^2, |P - c|^2, the margins' scale
    bool behind, ahead;   // u.d > 0, u.d + t a < 0, both clear of their rounding
};
RT_DEV PruneAbout prune_about(float cx, float cy, float cz, const PruneRay &y) {
    const float ux = y.o.x - cx, uy = y.o.y - cy, uz = y.o.z - cz;
    PruneAbout q;
    q.so = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
    const float bf = __builtin_fmaf(uz, y.d.z, __builtin_fmaf(uy, y.d.y, ux * y.d.x));
    const float pd = __builtin_fmaf(y.t, y.a, bf);  // (P - c).d
    q.sp = __builtin_fmaf(y.t, pd + bf, q.so);
    q.X = __builtin_fmaf(y.t2, __builtin_fabsf(bf), q.so + y.t2a);
    q.behind = (bf > 0.0f) & (bf * bf > q.so * y.a38);
    q.ahead = (pd < 0.0f) & (pd * pd > q.X * y.a36);
    return q;
}
// m_in, m_out: the margins' m of the inside and of the outside comparisons.
RT_DEV bool no_root_up_to_t(const PruneAbout &q, float r2in, float r2out, float m_in, float m_out, const PruneRay &y) {
    const float Mi = __builtin_fmaf(m_in, q.X, y.k1), Mo = __builtin_fmaf(m_out, q.X, y.k1);
    const bool neg = (q.so + Mi < r2in) & (q.sp + Mi < r2in);
    const bool pos0 = q.so - Mo > r2out, post = q.sp - Mo > r2out;
    return (pos0 & q.behind) | (y.has_t & (neg | (pos0 & post & q.ahead)));
}
RT_DEV bool cannot_win(f4v rec, const PruneRay &y) {
    const float r2 = rec.w * rec.w;
    return no_root_up_to_t(prune_about(rec.x, rec.y, rec.z, y), __builtin_fmaf(-0x1p-10f, r2, r2), __builtin_fmaf(0x1p-10f, r2, r2), 0x1p-15f,
                           0x1p-15f, y);
}
// The same test for spheres about one centre (a word of concentric shells): with the ray's numbers q about that centre
// the comparisons of no_root_up_to_t depend on the sphere through its radius alone, so they are folded into two
// thresholds per word -- cannot_win(sphere) = r2in > above_in || r2out < below_out, the very comparisons of cannot_win
// on the very same values, regrouped (max / min of the left-hand sides; a comparison that a flag switches off gets an
// infinite threshold).
struct PruneRadii { float above_in, below_out; };
// no_root_up_to_t(q, r2in, r2out, m_in, m_out, y) == (r2in > above_in) | (r2out < below_out).  A NaN on the left of one of
// its comparisons makes that comparison fail there; here it makes the threshold infinite, which does the same.
RT_DEV PruneRadii prune_radii(const PruneAbout &q, const PruneRay &y, float m_in, float m_out) {
    const float Mi = __builtin_fmaf(m_in, q.X, y.k1), Mo = __builtin_fmaf(m_out, q.X, y.k1);
    const float inf = __builtin_inff();
    PruneRadii t;
    // neg: so + Mi < r2in and sp + Mi < r2in, counted only with a bound
    const float i0 = q.so + Mi, i1 = q.sp + Mi;
    t.above_in = (y.has_t & (i0 == i0) & (i1 == i1)) ? __builtin_fmaxf(i0, i1) : inf;
    // pos0 and behind: so - Mo > r2out;  pos0 and post and ahead (with a bound): so - Mo > r2out and sp - Mo > r2out
    const float t0 = q.so - Mo, t1 = q.sp - Mo;
    const float a = (q.behind & (t0 == t0)) ? t0 : -inf;
    const float b = (y.has_t & q.ahead & (t0 == t0) & (t1 == t1)) ? __builtin_fminf(t0, t1) : -inf;
    t.below_out = __builtin_fmaxf(a, b);
    return t;
}
RT_DEV bool struck_by(const PruneRadii &t, float r2in, float r2out) { return (r2in > t.above_in) | (r2out < t.below_out); }
RT_DEV bool cannot_win_radius(const PruneRadii &t, float radius) {  // t = prune_radii(..., 2^-15, 2^-15): cannot_win's margins
    const float r2 = radius * radius;
    return struck_by(t, __builtin_fmaf(-0x1p-10f, r2, r2), __builtin_fmaf(0x1p-10f, r2, r2));
}
constexpr float kBallMarginIn = 0x1p-15f, kBallMarginOut = 0x1p-8f;
// Shells about one centre whose radii fall (or rise) along their slots -- a run, found at scene upload
// (TraceParams::word_bounds) -- need no walk over candidates at all: cannot_win_radius is monotone in the radius on
// either side (r2in > above_in holds for the large shells of the run, r2out < below_out for the small ones), so the
// shells it leaves standing are one range of slots, and two binary searches over the radii find it: [sb, se).
// Every probe evaluates the very comparison of cannot_win_radius on that slot's radius, so the range is exactly the
// set of slots of the run that the single-sphere test would leave.
template <typename F>
RT_DEV void run_survivors(const F &f, const PruneRadii &t, int rb, int re, bool falling, int &sb, int &se) {
    const int len = re - rb;
    // positions j = 0 .. len - 1 in the order of falling radii; first position that is not too large, first that is too small
    int lo0 = 0, hi0 = len, lo1 = 0, hi1 = len;
    for (int span = len; span > 0; span >>= 1) {  // enough turns for a range of len + 1 answers (wave-uniform)
        const int m0 = (lo0 + hi0) >> 1, m1 = (lo1 + hi1) >> 1;
        const int j0 = min(m0, len - 1), j1 = min(m1, len - 1);
        float ra = f.radius(falling ? rb + j0 : re - 1 - j0), rc = f.radius(falling ? rb + j1 : re - 1 - j1);
        asm volatile("" : "+v"(ra), "+v"(rc));
        const float r2a = ra * ra, r2c = rc * rc;
        const bool large = __builtin_fmaf(-0x1p-10f, r2a, r2a) > t.above_in;
        const bool small = __builtin_fmaf(0x1p-10f, r2c, r2c) < t.below_out;
        if (lo0 < hi0) {
            if (large) lo0 = m0 + 1; else hi0 = m0;
        }
        if (lo1 < hi1) {
            if (small) hi1 = m1; else lo1 = m1 + 1;
        }
    }
    sb = falling ? rb + lo0 : re - lo1;
    se = falling ? rb + lo1 : re - lo0;
}
// Strikes the words whose 32 spheres cannot win as a whole (wave-uniform loop over the words that are non-empty in
// some lane; the balls come through scalar loads, and a word about the same centre as the one before reuses the
// ray's numbers about it).  Slots the word's balls leave out (exempt) get the single-sphere test, whether the word
// goes or not.  Words of a run of shells (above) are finished here: every lane keeps the candidates inside its range of
// the run; runmask returns those words (wave-uniform), run_before / run_after the lane's candidates in them before and
// after (run_before's upper half: how many of them the lane lost altogether).  Returns the lane's remaining non-empty words.
template <typename F>
RT_DEV uint32_t cull_words(const F &f, const float4 *word_bounds, const float4 *round_runs, uint32_t *wave_masks, uint32_t nz, int slot0, int lane,
                           const PruneRay &y, uint32_t &runmask, int &run_before, int &run_after) {
    const const_f4_ptr balls = (const_f4_ptr)(uintptr_t)word_bounds + kBoundVec4 * (slot0 >> 5);
    uint32_t any = nz;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) any |= (uint32_t)__shfl_xor((int)any, off, 64);
    any = (uint32_t)__builtin_amdgcn_readfirstlane((int)any);
    float pcx = __builtin_nanf(""), pcy = pcx, pcz = pcx;  // the centre q is about (none yet)
    PruneAbout about = {};
    PruneRadii q = {}, q1 = {};   // thresholds with the balls' margins, and with the single-sphere test's (runs)
    bool q1_valid = false;
    uint32_t run = 0xFFFFFFFFu;   // the run [sb, se) is the lane's range of
    int sb = 0, se = 0;
    runmask = 0;
    run_before = run_after = 0;
    // The round's run, when its words of runs all belong to one (TraceParams::round_runs: the usual case, and the record
    // then saves the loop below its turn per word): the lane's range of the run is found first, its words without exempt
    // slots that the range does not reach go without being read, and the one or two it reaches keep what lies inside.
    {
        const const_f4_ptr rr = (const_f4_ptr)(uintptr_t)round_runs + 2 * (slot0 / (32 * kRoundWords));
        const f4v ra = rr[0], rb = rr[1];
        const uint32_t rw = __float_as_uint(rb.x);
        const uint32_t fast = (rw & 0xFFFFu) & ~(rw >> 16) & any;  // (wave-uniform)
        if (fast) {
            about = prune_about(ra.x, ra.y, ra.z, y);
            q = prune_radii(about, y, kBallMarginIn, kBallMarginOut);
            q1 = prune_radii(about, y, 0x1p-15f, 0x1p-15f);
            pcx = ra.x; pcy = ra.y; pcz = ra.z;
            q1_valid = true;
            run = __float_as_uint(ra.w);
            run_survivors(f, q1, (int)(run & 0xFFFFu), (int)(run >> 16), rb.y == 2.0f, sb, se);
            runmask = fast;
            any &= ~fast;
            const int lo_w = max((sb - slot0) >> 5, 0), hi_w = min((se - 1 - slot0) >> 5, kRoundWords - 1);
            const uint32_t reach = se > sb && hi_w >= lo_w ? ((2u << hi_w) - 1u) & ~((1u << lo_w) - 1u) : 0u;
            const uint32_t mine = nz & fast;
            // (a word that goes unread counts like a word the balls strike: kDenseWordBits candidates; see run_before below)
            const int unread = __builtin_popcount(mine & ~reach);
            run_before = unread * ((1 << 16) + kDenseWordBits);
            nz &= ~(mine & ~reach);
            uint32_t todo = mine & reach;
            while (__builtin_amdgcn_ballot_w64(todo != 0u)) {
                const bool live = todo != 0u;
                const int w = live ? __builtin_ctz(todo) : 0;
                todo &= todo - 1;
                const uint32_t m = wave_masks[w * 64 + lane];
                const int base = slot0 + 32 * w;
                const int lo = max(sb - base, 0), hi = min(se - base, 32);
                const uint32_t m2 = hi > lo ? m & (0xFFFFFFFFu >> lo) & (0xFFFFFFFFu << (32 - hi)) : 0u;
                if (live) {
                    run_before += __builtin_popcount(m) + (m2 == 0u ? 1 << 16 : 0);
                    run_after += __builtin_popcount(m2);
                    if (m2 != m) wave_masks[w * 64 + lane] = m2;
                    if (m2 == 0u) nz &= ~(1u << w);
                }
            }
        }
    }
    while (any) {
        const int w = __builtin_ctz(any);
        any &= any - 1;
        const f4v outer = balls[kBoundVec4 * w], inner = balls[kBoundVec4 * w + 1];
        const uint32_t exempt = __float_as_uint(inner.y);  // wave-uniform
        if (!(outer.x == pcx && outer.y == pcy && outer.z == pcz)) {
            about = prune_about(outer.x, outer.y, outer.z, y);
            q = prune_radii(about, y, kBallMarginIn, kBallMarginOut);
            pcx = outer.x; pcy = outer.y; pcz = outer.z;
            q1_valid = false;
        }
        // the spheres the balls leave out get the single-sphere test here, for every lane at once (same slot for every
        // lane): a lane does not keep a word for the sake of one sphere it could strike, and the passes that follow
        // need not look at these slots again
        uint32_t keep = exempt;
        for (uint32_t e = exempt; e; e &= e - 1u) {
            const int p = 31 - __builtin_ctz(e);  // position in the word
            if (cannot_win(f.load1(slot0 + 32 * w + p), y)) keep &= ~(0x80000000u >> p);
        }
        if (inner.z >= 2.0f) {  // (wave-uniform) every slot of the word but the exempt ones belongs to one run
            const uint32_t r = __float_as_uint(inner.w);
            if (!q1_valid) {
                q1 = prune_radii(about, y, 0x1p-15f, 0x1p-15f);
                q1_valid = true;
                run = 0xFFFFFFFFu;
            }
            if (r != run) {
                run = r;
                run_survivors(f, q1, (int)(r & 0xFFFFu), (int)(r >> 16), inner.z == 2.0f, sb, se);
            }
            runmask |= 1u << w;
            if ((nz >> w) & 1u) {
                const uint32_t m = wave_masks[w * 64 + lane];
                const int base = slot0 + 32 * w;
                const int lo = max(sb - base, 0), hi = min(se - base, 32);
                const uint32_t range = hi > lo ? (0xFFFFFFFFu >> lo) & (0xFFFFFFFFu << (32 - hi)) : 0u;
                const uint32_t m2 = m & (keep | (range & ~exempt));
                run_before += __builtin_popcount(m) + (m2 == 0u ? 1 << 16 : 0);  // (upper half: words of runs that went altogether)
                run_after += __builtin_popcount(m2);
                if (m2 != m) wave_masks[w * 64 + lane] = m2;
                if (m2 == 0u) nz &= ~(1u << w);
            }
            continue;
        }
        const bool struck = struck_by(q, inner.x, outer.w);
        if ((nz >> w) & 1u) {
            const uint32_t m = wave_masks[w * 64 + lane];
            const uint32_t m2 = m & (struck ? keep : keep | ~exempt);
            if (m2 != m) wave_masks[w * 64 + lane] = m2;
            if (m2 == 0u) nz &= ~(1u << w);
        }
    }
    return nz;
}
// The same a granule of 8 spheres at a time, every lane going through its own remaining words (the balls come
// through vector loads: the words differ from lane to lane).  When the four granules of every lane's word share
// their centre, the ray's numbers about it are worked out once.
RT_DEV uint32_t cull_granules(const float4 *word_bounds, uint32_t *wave_masks, uint32_t nz, int slot0, int lane, const PruneRay &y,
                              int &before, int &after) {
    const float4 *balls = word_bounds + kBoundVec4 * (slot0 >> 5);
    uint32_t todo = nz;
    before = 0;  // candidates in the words the lane came with, and how many of them the balls left
    after = 0;
    while (__builtin_amdgcn_ballot_w64(todo != 0u)) {
        const bool live = todo != 0u;
        const int w = live ? __builtin_ctz(todo) : 0;  // a lane that is through looks at the round's first word again
        todo &= todo - 1;
        const float4 *g4 = balls + kBoundVec4 * w + 2;
        // the centres first (a granule's record: (C, R2out), (R2in, exempt bits, 0, 0)): when the four granules of every
        // lane's word share theirs, the ray's numbers about it are worked out once; the radii are read two granules at a
        // time, so that the sixteen values of a word are not all in registers at once
        float4 c0 = g4[0];
        const float4 c1 = g4[2], c2 = g4[4], c3 = g4[6];
        const bool same = (c1.x == c0.x) & (c1.y == c0.y) & (c1.z == c0.z) & (c2.x == c0.x) & (c2.y == c0.y) & (c2.z == c0.z) &
                          (c3.x == c0.x) & (c3.y == c0.y) & (c3.z == c0.z);
        uint32_t kill = 0;
        if (__builtin_amdgcn_ballot_w64(!same) == 0) {
            const PruneRadii q = prune_radii(prune_about(c0.x, c0.y, c0.z, y), y, kBallMarginIn, kBallMarginOut);
            const float w1 = c1.w, w2 = c2.w, w3 = c3.w;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 in = g4[2 * g + 1];
                const float r2out = g == 0 ? c0.w : g == 1 ? w1 : g == 2 ? w2 : w3;
                const bool gone = struck_by(q, in.x, r2out);
                kill |= gone ? (0xFF000000u >> (8 * g)) & ~__float_as_uint(in.y) : 0u;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 out = g == 0 ? c0 : g == 1 ? c1 : g == 2 ? c2 : c3;
                const float4 in = g4[2 * g + 1];
                const PruneAbout q = prune_about(out.x, out.y, out.z, y);
                const bool gone = no_root_up_to_t(q, in.x, out.w, kBallMarginIn, kBallMarginOut, y);
                kill |= gone ? (0xFF000000u >> (8 * g)) & ~__float_as_uint(in.y) : 0u;
            }
        }
        if (live) {
            const uint32_t m = wave_masks[w * 64 + lane];
            before += __builtin_popcount(m);
            after += __builtin_popcount(m & ~kill);
            if (m & kill) {
                wave_masks[w * 64 + lane] = m & ~kill;
                if ((m & ~kill) == 0u) nz &= ~(1u << w);
            }
        }
    }
    return nz;
}
// Every lane walks its own candidate words (four candidates per turn, as resolve_word_per_lane takes them), strikes
// what cannot win and writes the words back; returns the lane's non-empty words and, in left, how many candidates
// they hold.
template <typename F>
RT_DEV uint32_t prune_candidates(const F &f, const float4 *word_bounds, uint32_t *wave_masks, uint32_t nz, int slot0, int lane,
                                 const PruneRay &y, int &left) {
    const float4 *balls = word_bounds + kBoundVec4 * (slot0 >> 5);
    uint32_t todo = nz, m = 0, word = 0, gone = 0;
    int w = 0;
    left = 0;  // candidates the lane has left
    // the spheres of the lane's current word, but for those its balls leave out, have one centre (the host says so,
    // TraceParams::word_bounds), and about is the ray's numbers about it, worked out once per word instead of once per
    // candidate
    bool concentric = false;
    PruneRadii about = {};
    for (;;) {
        if (m == 0) {
            if (gone) {
                word &= ~gone;
                wave_masks[w * 64 + lane] = word;
                if (word == 0u) nz &= ~(1u << w);
                gone = 0;
            }
            left += __builtin_popcount(word);
            word = 0;
            if (todo) {
                w = __builtin_ctz(todo);
                todo &= todo - 1;
                word = wave_masks[w * 64 + lane];
                const float4 outer = balls[kBoundVec4 * w], inner = balls[kBoundVec4 * w + 1];
                m = word & ~__float_as_uint(inner.y);  // the slots the balls leave out have had this test (cull_words)
                concentric = inner.z != 0.0f;
                if (concentric) about = prune_radii(prune_about(outer.x, outer.y, outer.z, y), y, 0x1p-15f, 0x1p-15f);
            }
        }
        if (__builtin_amdgcn_ballot_w64(m != 0u) == 0) break;
        // a lane that is through looks at the round's first sphere again and drops the answer
        const bool live = m != 0u;
        const uint32_t mm = live ? m : 1u;
        const int p0 = __builtin_clz(mm), p3 = 31 - __builtin_ctz(mm);
        const uint32_t m1 = mm & ~((0x80000000u >> p0) | (0x80000000u >> p3));
        const int p1 = m1 ? __builtin_clz(m1) : p0, p2 = m1 ? 31 - __builtin_ctz(m1) : p3;
        m = live ? m1 & ~((0x80000000u >> p1) | (0x80000000u >> p2)) : 0u;
        const int base = live ? slot0 + w * 32 : slot0 - 31;
        bool g0, g1, g2, g3;
        if (__builtin_amdgcn_ballot_w64(live && !concentric) == 0) {  // every lane that is still at it is in such a word
            float r0 = f.radius(base + p0), r1 = f.radius(base + p1), r2 = f.radius(base + p2), r3 = f.radius(base + p3);
            asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
            g0 = cannot_win_radius(about, r0); g1 = cannot_win_radius(about, r1);
            g2 = cannot_win_radius(about, r2); g3 = cannot_win_radius(about, r3);
        } else {
            f4v c0 = f.load1(base + p0), c1 = f.load1(base + p1), c2 = f.load1(base + p2), c3 = f.load1(base + p3);
            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
            g0 = cannot_win(c0, y); g1 = cannot_win(c1, y); g2 = cannot_win(c2, y); g3 = cannot_win(c3, y);
        }
        const uint32_t g = (g0 ? 0x80000000u >> p0 : 0u) | (g1 ? 0x80000000u >> p1 : 0u) | (g2 ? 0x80000000u >> p2 : 0u) | (g3 ? 0x80000000u >> p3 : 0u);
        gone |= live ? g : 0u;
    }
    return nz;
}

// The exact pass over what a lane has left, every lane taking its next four candidates wherever its words hold them
// (resolve_word_per_lane goes word by word for the whole wave, which costs a turn per word when a few candidates are
// spread over several).  Empties the words it reads from nz.
template <typename F, bool PROF>
RT_DEV uint32_t resolve_rest_per_lane(const F &f, const uint32_t *wave_masks, uint32_t nz, int slot0, int lane, V3 o, V3 d, const RootDiv &dv,
                                      LaneBest &best, Prof<PROF> &prof) {
    uint32_t m = 0;
    int w = 0;
    for (;;) {
        int s[4];
        bool v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (m == 0u && nz != 0u) {
                w = __builtin_ctz(nz);
                nz &= nz - 1;
                m = wave_masks[w * 64 + lane];
            }
            v[i] = m != 0u;
            const int p = v[i] ? __builtin_clz(m) : 0;
            m = v[i] ? m & ~(0x80000000u >> p) : 0u;
            s[i] = v[i] ? slot0 + w * 32 + p : slot0;  // a lane that is through reads the round's first sphere and accepts nothing
        }
        if (__builtin_amdgcn_ballot_w64(v[0]) == 0) break;
        if constexpr (PROF) {
            prof.count(6);
            prof.count(7, (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(v[0])) + (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(v[1])) +
                              (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(v[2])) + (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(v[3])));
        }
        f4v c0 = f.load1(s[0]), c1 = f.load1(s[1]), c2 = f.load1(s[2]), c3 = f.load1(s[3]);
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
        const float none = -__builtin_inff(), upto = best.prune_above();
        const float n0 = v[0] ? upto : none, n1 = v[1] ? upto : none, n2 = v[2] ? upto : none, n3 = v[3] ? upto : none;
        const LaneRoot r0 = lane_root(c0, o, d, dv, n0), r1 = lane_root(c1, o, d, dv, n1);
        const LaneRoot r2 = lane_root(c2, o, d, dv, n2), r3 = lane_root(c3, o, d, dv, n3);
        lane_offer(best, r0, (uint32_t)s[0]);
        lane_offer(best, r1, (uint32_t)s[1]);
        lane_offer(best, r2, (uint32_t)s[2]);
        lane_offer(best, r3, (uint32_t)s[3]);
        if (__builtin_amdgcn_ballot_w64((v[0] & r0.retry) | (v[1] & r1.retry) | (v[2] & r2.retry) | (v[3] & r3.retry))) {
            if (v[0]) lane_retry(best, r0, (uint32_t)s[0], dv, n0);
            if (v[1]) lane_retry(best, r1, (uint32_t)s[1], dv, n1);
            if (v[2]) lane_retry(best, r2, (uint32_t)s[2], dv, n2);
            if (v[3]) lane_retry(best, r3, (uint32_t)s[3], dv, n3);
        }
    }
    return 0u;
}

// Dense rounds (most lanes hold many candidates: rays among nested shells collect every shell their line crosses).
// (1) Probe: the four candidates next to the sphere the lane's path hit last (two at or before it in slot order, two
// after it) are resolved exactly; among nested shells in radius order the next hit is one of them, elsewhere it is
// merely a bound.  (2) With that bound whole words are struck (cull_words), then single candidates of the remaining
// words (prune_candidates), and the survivors written back.  (3) When no lane has more than kRestPerLane candidates
// left, every lane resolves its own (resolve_rest_per_lane).  Otherwise words that are still dense in most lanes are
// resolved per lane and struck from nz, the others are left to the pooled form; all = true: every word per lane.
// paid: the balls struck two thirds of the candidates of the lanes that came with three or more words (the caller's
// cue to come back).
template <typename F, bool PROF>
RT_DEV uint32_t resolve_dense_words(const F &f, const float4 *word_bounds, const float4 *round_runs, uint32_t *wave_masks, uint32_t nz, int slot0, int lane, V3 o,
                                    V3 d, const RootDiv &dv, bool all, int hint, bool &paid, Prof<PROF> &prof) {
    // the lane's closest (distance, slot) so far lives in LDS beside the pooled form's minima (closest_sphere); it is
    // held in registers only here, so that the sphere filter does not carry it
    unsigned long long *best_lds = (unsigned long long *)(wave_masks + kBestOffset);
    const unsigned long long b0 = best_lds[lane];
    LaneBest best = {__uint_as_float((uint32_t)(b0 >> 32)), (uint32_t)b0, __builtin_inff()};
    {  // (1)
        const int rel = min(max(hint - slot0, 0), 32 * kRoundWords - 1);
        const int hw = rel >> 5, hp = rel & 31;
        const uint32_t mh = (nz >> hw) & 1u ? wave_masks[hw * 64 + lane] : 0u;
        const uint32_t upto = 0xFFFFFFFFu << (31 - hp);  // positions 0 .. hp
        uint32_t below = mh & upto, above = mh & ~upto;
        int wb = hw, wa = hw;
        if (below == 0u) {  // nearest non-empty word before
            const uint32_t nzb = nz & ((1u << hw) - 1u);
            if (nzb) {
                wb = 31 - __builtin_clz(nzb);
                below = wave_masks[wb * 64 + lane];
            }
        }
        if (above == 0u) {  // nearest non-empty word after
            const uint32_t nza = hw < 31 ? nz & ~((2u << hw) - 1u) : 0u;
            if (nza) {
                wa = __builtin_ctz(nza);
                above = wave_masks[wa * 64 + lane];
            }
        }
        int k[4] = {-1, -1, -1, -1};
        if (below) {
            int q = 31 - __builtin_ctz(below);
            k[0] = slot0 + wb * 32 + q;
            below &= ~(0x80000000u >> q);
            if (below) k[1] = slot0 + wb * 32 + 31 - __builtin_ctz(below);
        }
        if (above) {
            int q = __builtin_clz(above);
            k[2] = slot0 + wa * 32 + q;
            above &= ~(0x80000000u >> q);
            if (above) k[3] = slot0 + wa * 32 + __builtin_clz(above);
        }
        const int any_k = k[0] >= 0 ? k[0] : k[2];
        const bool on = any_k >= 0;
        const int s0 = k[0] >= 0 ? k[0] : (on ? any_k : slot0), s1 = k[1] >= 0 ? k[1] : s0, s2 = k[2] >= 0 ? k[2] : s0, s3 = k[3] >= 0 ? k[3] : s2;
        f4v c0 = f.load1(s0), c1 = f.load1(s1), c2 = f.load1(s2), c3 = f.load1(s3);
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
        const float nprune = on ? best.prune_above() : -__builtin_inff();  // a lane without candidates accepts nothing
        const LaneRoot r0 = lane_root(c0, o, d, dv, nprune), r1 = lane_root(c1, o, d, dv, nprune);
        const LaneRoot r2 = lane_root(c2, o, d, dv, nprune), r3 = lane_root(c3, o, d, dv, nprune);
        lane_offer(best, r0, (uint32_t)s0);
        lane_offer(best, r1, (uint32_t)s1);
        lane_offer(best, r2, (uint32_t)s2);
        lane_offer(best, r3, (uint32_t)s3);
        if (__builtin_amdgcn_ballot_w64(on && (r0.retry || r1.retry || r2.retry || r3.retry))) {
            if (on) {
                lane_retry(best, r0, (uint32_t)s0, dv, nprune);
                lane_retry(best, r1, (uint32_t)s1, dv, nprune);
                lane_retry(best, r2, (uint32_t)s2, dv, nprune);
                lane_retry(best, r3, (uint32_t)s3, dv, nprune);
            }
        }
        // these candidates are done with: out of the words, so that the passes below neither test nor resolve them again
        if (on) {
            uint32_t clr_b = 0, clr_a = 0;
            if (k[0] >= 0) clr_b |= 0x80000000u >> ((k[0] - slot0) & 31);
            if (k[1] >= 0) clr_b |= 0x80000000u >> ((k[1] - slot0) & 31);
            if (k[2] >= 0) clr_a |= 0x80000000u >> ((k[2] - slot0) & 31);
            if (k[3] >= 0) clr_a |= 0x80000000u >> ((k[3] - slot0) & 31);
            if (wa == wb) {
                clr_b |= clr_a;
                clr_a = 0;
            }
            if (clr_b) {
                const uint32_t m = wave_masks[wb * 64 + lane] & ~clr_b;
                wave_masks[wb * 64 + lane] = m;
                if (m == 0u) nz &= ~(1u << wb);
            }
            if (clr_a) {
                const uint32_t m = wave_masks[wa * 64 + lane] & ~clr_a;
                wave_masks[wa * 64 + lane] = m;
                if (m == 0u) nz &= ~(1u << wa);
            }
        }
    }
    const PruneRay y = make_prune_ray(o, d, dv.a, best.slot != 0xFFFFFFFFu, best.t);
    // (2) ... and did the balls strike two thirds of the candidates of the lanes that came with three or more words?
    // (a word struck as a whole is taken for kDenseWordBits candidates; before in the upper half, after in the lower)
    const int words0 = __builtin_popcount(nz);
    uint32_t runmask;
    int run_before, run_after;
    nz = cull_words(f, word_bounds, round_runs, wave_masks, nz, slot0, lane, y, runmask, run_before, run_after);
    int before, after;
    // (the words of runs count with their candidates as they were and as they are, not as words struck)
    const int struck = words0 - __builtin_popcount(nz) - (run_before >> 16);
    run_before &= 0xFFFF;
    const uint32_t nz_run = nz & runmask;   // finished: the passes below leave these words alone
    nz = cull_granules(word_bounds, wave_masks, nz & ~runmask, slot0, lane, y, before, after);
    int tally = words0 >= 3 ? ((before + run_before + kDenseWordBits * struck) << 16) | (after + run_after) : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tally += __shfl_xor(tally, off, 64);
    paid = 3 * (tally & 0xFFFF) <= (tally >> 16);
    int left;
    nz = prune_candidates(f, word_bounds, wave_masks, nz, slot0, lane, y, left);
    nz |= nz_run;
    left += run_after;
    // (3) A handful left in the whole wave: the pooled form resolves them in a flush or two, every lane one candidate
    // (a turn of the per-lane form costs four candidates' arithmetic in every lane, however few have any).
    int total = left;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    if (total <= kRestPooled) {
        // nothing here: the caller's resolve_balanced takes what the words still hold
    } else if (__builtin_amdgcn_ballot_w64(left > kRestPerLane) == 0) {
        // (3a) few left in every lane: each lane resolves its own, across its words
        nz = resolve_rest_per_lane(f, wave_masks, nz, slot0, lane, o, d, dv, best, prof);
    } else {
        // (3b) words that are non-empty in some lane (wave-uniform)
        uint32_t any = nz;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) any |= (uint32_t)__shfl_xor((int)any, off, 64);
        any = (uint32_t)__builtin_amdgcn_readfirstlane((int)any);
        while (any) {
            const int w = __builtin_ctz(any);
            any &= any - 1;
            const uint32_t m = (nz >> w) & 1u ? wave_masks[w * 64 + lane] : 0u;
            const uint64_t dense = __builtin_amdgcn_ballot_w64(__builtin_popcount(m) >= kDenseWordBits);
            if (all || __builtin_popcountll(dense) >= 32) {
                resolve_word_per_lane(f, m, slot0 + w * 32, o, d, dv, best, prof);
                nz &= ~(1u << w);
            }
        }
    }
    best_lds[lane] = ((unsigned long long)__float_as_uint(best.t) << 32) | best.slot;
    return nz;
}

// Closest sphere hit of ray (o, d) over all spheres: the sphere part of world_hit (Shaders.metal:13-32) with
// hitSphere (Lib.metal:215-246).  Returns the sphere slot or -1; t in tbest.  Must be called by all 64 lanes of the
// wave (the streaming fetcher loads tiles cooperatively); lanes without a ray pass active = false and get -1.
//
// Conservativeness of pass A.  Let oc = o - c, a = d.d, b = oc.d, c = |oc|^2 - r^2 as the reference evaluates them
// (uncontracted float), u = 2^-24.  Whenever the reference's discriminant b*b - a*c is >= 0, rounding analysis
// gives b^2/a >= |oc|^2 - r^2 - 17 u |oc|^2 in exact arithmetic.
//  General filter: evaluates the same difference with fused operations from dn = d (1 + 2^-18)/|d| (rsq, 1 ulp)
//   and r2q = r (1 + 2^-17) r (two roundings)
//   with error <= ~17 u |oc|^2 + 4 u r^2.  The direction scale adds >= 64 u |oc|^2 of slack when |oc|^2 >= 2 r^2;
//   the 2^-17 = 128 u inflation of r^2 covers |oc|^2 < 2 r^2 (needs <= 72 u r^2).  Margin >= 1.8x.
//  Compact filter: the expanded form cancels terms of size S = |o|^2 + |c|^2: its own error is <= ~50 u S, the
//   reference-side term is <= 34 u S, and it subtracts 2^-16 (S + r^2) = 256 u (S + r^2).  Margin >= 3x.  Because
//   that slack grows with S, a wave uses the compact filter only while every lane's |o|^2 is below the
//   scene-dependent oo_limit; the general filter is translation invariant and is the fallback.
// The bounds assume operands in the normal float range with magnitudes below 2^60.  A NaN or infinity in q has
// an unspecified sign bit; that is harmless because such operands cannot produce a finite root in pass B's
// arithmetic either.
template <typename F, bool PROF, bool DENSE>
RT_DEV int closest_sphere(const TraceParams &P, F &f, uint32_t *wave_masks, bool active, V3 o, V3 d, float &tbest,
                          Prof<PROF> &prof, Work &work) {
    const float a = dot(d, d);
    const bool a_ok = !(__builtin_fabsf(a) < 1e-6f);  // Lib.metal:226-228: such a ray hits no sphere
    const RootDiv dv = root_div(a, active);
    const float s = __builtin_amdgcn_rsqf(a) * kDirScale;
    const V3 dn = v3(d.x * s, d.y * s, d.z * s);
    const int lane = threadIdx.x & 63;
    float oo;
    RayLifted lifted;
    h8 slots[4];
    if constexpr (F::kMatrix) {
        lift_ray_slots(P.lift, o, dn, kMfmaSlack, oo, slots);
    } else {
        oo = __builtin_fmaf(o.z, o.z, __builtin_fmaf(o.y, o.y, o.x * o.x));
    }
    const bool general = __builtin_amdgcn_ballot_w64(active && !(oo <= P.oo_limit)) != 0;  // wave-uniform
    // every lane's closest (distance, slot) so far, kept in LDS where other lanes can lower it (resolve_balanced)
    unsigned long long *best = (unsigned long long *)(wave_masks + kBestOffset);
    best[lane] = ((unsigned long long)__float_as_uint(tbest) << 32) | 0xFFFFFFFFull;
    f.start(general);
    // Matrix kind: the filter at half width when at most 32 rays of the wave can touch the box of the scene's small
    // spheres (filter_round_narrow); those rays are moved into one ray tile (gather_columns).
    uint64_t touching = 0;
    bool narrow = false;
    int owner = lane;
    // (not in the kernels for scenes whose rays cross many spheres, which have no registers to spare for it: the host never
    // sets P.narrow for them; the stamped instantiation, which stands in for both, carries it)
    constexpr bool kNarrow = F::kMatrix && (!DENSE || PROF);
    if constexpr (kNarrow) {
        if (P.narrow != 0 && !general && work.wide_waves < 4u) {
            const bool touch = active && a_ok && (!(oo <= P.narrow_oo) || ray_may_touch_box(o, d, P.cull_mid, P.cull_half));
            touching = __builtin_amdgcn_ballot_w64(touch);
            const int n = __builtin_popcountll(touching);
            narrow = n <= 32;
            work.wide_waves = narrow ? 0u : work.wide_waves + 1u;
            if (narrow) gather_columns(wave_masks, lane, touching, slots, false, lifted.b00, lifted.b01, owner);
        } else if (P.narrow != 0 && !general) {
            // a wave most of whose rays touch the box stops asking for a while
            work.wide_waves = (work.wide_waves + 1u) & 31u;
        }
        if (!narrow) exchange_lifted(slots, lifted);
    } else if constexpr (F::kMatrix) {
        exchange_lifted(slots, lifted);
    }
    for (int g0 = 0; g0 < P.ngroups; g0 += kRoundWords * kWordGroups) {
        const int g1 = min(P.ngroups, g0 + kRoundWords * kWordGroups);
        uint32_t nz;
        if (general) {
            RayGeneral y = {o, dn};
            nz = filter_round(f, g0, g1, P.nsph, wave_masks, lane, y);
        } else if constexpr (F::kMatrix) {
            // The filter is long bursts of matrix and vector issue, everything else in a trip is short dependent
            // chains: a wave outside the filter goes first when it has an instruction ready (+2.8 % measured).
            __builtin_amdgcn_s_setprio(0);
            const int blk0 = g0 / 16, blk1 = (g1 + 15) / 16;
            if (kNarrow && narrow) {
                nz = 0;
                if (touching != 0) {
                    const uint32_t written = filter_round_narrow(f.pm, blk0, blk1, wave_masks, lane, lifted.b00, lifted.b01, owner,
                                                                 (lane & 31) < __builtin_popcountll(touching));
                    nz = scatter_bitmaps(wave_masks, lane, touching, written);
                    work.blocks += (uint32_t)(blk1 - blk0);
                }
                // the spheres the box does not stand for, every lane for its own ray
                const RayGeneral y = {o, dn};
                for (int k = 0; k < P.wide_words; ++k) {
                    const int w = (int)P.wide_word[k] - g0 / kWordGroups;
                    if (w < 0 || w >= kRoundWords) continue;
                    uint32_t m = (nz >> w) & 1u ? wave_masks[w * 64 + lane] : 0u;
                    // (Leaving a sphere no filter can hold -- the cover scene's ground -- out for rays that start outside it and
                    // lead away, b = oc.d > 0 and c = |oc|^2 - r^2 > 0 in the reference's own arithmetic, was measured: half of
                    // the rays qualify, and the twelve instructions cost 0.8 % more than the exact pass saves.)
                    m |= P.wide_always[k];
                    for (uint32_t t = P.wide_test[k]; t != 0; t &= t - 1) {
                        const int j = 31 - __builtin_ctz(t);  // bit 31 - j is slot 32 w + j
                        const float q = sphere_q(f.load1(4 * g0 + 32 * w + j), y);
                        m |= (~__float_as_uint(q) & 0x80000000u) >> j;
                    }
                    wave_masks[w * 64 + lane] = m;
                    nz |= (m != 0u ? 1u : 0u) << w;
                }
            } else {
                nz = filter_round_matrix(f.pm, blk0, blk1, wave_masks, lane, lifted);
                work.blocks += 2u * (uint32_t)(blk1 - blk0);
            }
            __builtin_amdgcn_s_setprio(1);
        } else {
            RayCompact y;
            y.ndnx = -dn.x; y.ndny = -dn.y; y.ndnz = -dn.z;
            y.odn = __builtin_fmaf(o.z, dn.z, __builtin_fmaf(o.y, dn.y, o.x * dn.x));
            y.m2ox = -2.0f * o.x; y.m2oy = -2.0f * o.y; y.m2oz = -2.0f * o.z;
            y.oo1 = oo * (1.0f - kCompactSlack);
            if constexpr (F::kScalarRecords) {
                nz = filter_round(f, g0, g1, P.nsph, wave_masks, lane, pack_compact(y));
            } else {
                nz = filter_round(f, g0, g1, P.nsph, wave_masks, lane, y);
            }
        }
        if (!a_ok || !active) nz = 0;
        prof.lap(1);
        // Which form of pass B: pooled over the wave (the usual case: a couple of candidates per ray, a few lanes with
        // more), or, in rounds with long candidate lists (nested shells), the form that strikes what cannot win and
        // resolves the rest lane by lane (resolve_dense_words).  That code is only part of the kernels launched for
        // scenes in which rays cross many spheres (DENSE; the host decides at scene upload), so that the usual kernels
        // do not pay for it in registers.  A round qualifies when some lane has four or more non-empty words (the
        // pool's loop over layers is as long as the longest list, however few lanes have one).  Whether the form pays
        // is learnt from its own result: it reports whether the bounding balls struck two thirds of the candidates,
        // and a wave that did not see that in four qualifying rounds in a row goes back to the pool and tries again
        // only on every sixteenth of its searches (sending a round of long lists to the pool costs several times what
        // a round that does not pay costs here, hence the patience).  (A wave with a ray whose a = d.d is out of the
        // range of the shared-reciprocal division stays pooled.)
        if constexpr (DENSE) {
            // (dense_kernel: the stamped instantiation carries this code for every scene, and must not take it where the
            // kernels it stands in for would not)
            if (P.dense_kernel != 0 && P.resolve_mode != 1 && !dv.slow) {
                const bool qualifies = __builtin_amdgcn_ballot_w64(__builtin_popcount(nz) >= 4) != 0;
                if (P.resolve_mode == 2 || (qualifies && (work.unpaid < 4u || (work.trips & 15u) == 0u))) {
                    bool paid;
                    nz = resolve_dense_words(f, P.word_bounds, P.round_runs, wave_masks, nz, 4 * g0, lane, o, d, dv, P.resolve_mode == 2, work.last_slot, paid, prof);
                    if (qualifies) work.unpaid = paid ? 0u : min(work.unpaid + 1u, 4u);
                }
            }
        }
        resolve_balanced(f, wave_masks, nz, 4 * g0, lane, o, d, dv, prof);
        prof.lap(2);
    }
    const unsigned long long r = best[lane];
    tbest = __uint_as_float((uint32_t)(r >> 32));
    if constexpr (DENSE) {
        if ((uint32_t)r != 0xFFFFFFFFu) work.last_slot = (int)(uint32_t)r;
    }
    return (int)(uint32_t)r;  // 0xFFFFFFFF = -1: no hit
}

// Asynchronous copy of tile t (count spheres) into LDS buffer b: every wave of the workgroup copies 1 KiB pieces,
// each lane 16 bytes (global_load_lds_dwordx4: per-lane global address, LDS address = wave-uniform base + 16 * lane),
// of the lifted records of the tile's 64-sphere blocks and of its strict records.
RT_DEV void tile_copy(const TraceParams &P, const Fetch<kFetchTiled> &f, int t, int b, int count) {
    const float4 *src = P.tiles + (size_t)t * kTileVec4 + (threadIdx.x & 63);
    f4v *dst = f.buf + b * kTileVec4;
    const int blocks = (count + 63) / 64;
    const int nl = blocks * 256, ns = blocks * 64;  // both multiples of a wave's 64 pieces
    for (int u = (threadIdx.x >> 6) * 64; u < nl + ns; u += (blockDim.x >> 6) * 64) {
        const int v = u < nl ? u : kTileLifted + (u - nl);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + v),
                                         (__attribute__((address_space(3))) void *)(dst + v), 16, 0, 0);
    }
}

// closest_sphere for the tiled kind.  Must be called by every wave of the workgroup the same number of times (it
// contains workgroup barriers), after a workgroup barrier that follows the previous call (the first copy below
// overwrites a buffer the previous call read).  Per tile: wait for the tile (the barrier also says that nobody
// reads the other buffer any more), start the copy of the next tile into the other buffer, lift the rays into the
// tile's frame, matrix-core filter over the tile's 8 blocks (or the general vector filter on the tile's strict
// records when a ray of the wave starts outside the range of the tile's frame), exact pass B on the candidates.
// Tiles hold the spheres in spatial order, not object order: ties in distance go to the lower object index
// (sphere_root_any_order).  Returns the sphere slot in tile order.
template <bool PROF>
RT_DEV int closest_sphere_tiled(const TraceParams &P, Fetch<kFetchTiled> &f, uint32_t *wave_masks, bool active, V3 o, V3 d,
                                float &tbest, Prof<PROF> &prof, Work &work) {
    const float a = dot(d, d);
    const bool a_ok = !(__builtin_fabsf(a) < 1e-6f);  // Lib.metal:226-228
    const RootDiv dv = root_div(a, active);
    const float s = __builtin_amdgcn_rsqf(a) * kDirScale;
    const V3 dn = v3(d.x * s, d.y * s, d.z * s);
    const int lane = threadIdx.x & 63;
    int ibest = -1;
    float nskip = __builtin_inff();
    // scalar loads: a vector load in the loop would make the wave wait for the copy it has just started
    const const_f4_ptr frames = (const_f4_ptr)(uintptr_t)P.tile_frames;
    tile_copy(P, f, 0, 0, (int)frames[1].y);
    for (int t = 0; t < P.ntiles; ++t) {
        // The copy of tile t is this wave's only outstanding vector-memory operation here.  A workgroup barrier orders
        // LDS traffic (lgkmcnt) but does not wait for global -> LDS copies (vmcnt), and another wave reads the piece
        // this wave copied right after the barrier: wait for the copy explicitly.
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0), expcnt and lgkmcnt untouched
        __syncthreads();
        if (t + 1 < P.ntiles) tile_copy(P, f, t + 1, (t + 1) & 1, (int)frames[kTileFrameVec4 * (t + 1) + 1].y);
        const f4v *tile = f.buf + (t & 1) * kTileVec4;
        const f4v fa = frames[kTileFrameVec4 * t], fb = frames[kTileFrameVec4 * t + 1];
        const LiftFrame F = {fa.x, fa.y, fa.z, fa.w};
        const int count = (int)fb.y;
        float oo;
        RayLifted lifted;
        lift_ray_near(F, fb.z, o, dn, kMfmaSlack, kNearSlack, oo, lifted);
        const bool general = __builtin_amdgcn_ballot_w64(active && !(oo <= fb.x)) != 0;  // wave-uniform
        uint32_t nz;
        if (general) {
            RayGeneral y = {o, dn};
            TileStrict ts = {tile + kTileLifted};
            nz = filter_round(ts, 0, (count + 3) / 4, count, wave_masks, lane, y);
        } else {
            __builtin_amdgcn_s_setprio(0);
            if (P.ntiles >= 8)
                nz = filter_round_matrix<true>((const h8 *)tile, 0, (count + 63) / 64, wave_masks, lane, lifted);
            else
                nz = filter_round_matrix<false>((const h8 *)tile, 0, (count + 63) / 64, wave_masks, lane, lifted);
            __builtin_amdgcn_s_setprio(1);
            work.blocks += 2u * (uint32_t)((count + 63) / 64);
        }
        if (!a_ok || !active) nz = 0;
        prof.lap(1);
        uint32_t m = 0;
        int w = 0;
        for (;;) {
            if (m == 0) {
                if (nz == 0) break;
                w = __builtin_ctz(nz);
                nz &= nz - 1;
                m = wave_masks[w * 64 + lane];
            }
            const int j = __builtin_clz(m);
            m &= ~(0x80000000u >> j);
            if constexpr (PROF) {
                prof.count(6);
                prof.count(7, (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)));
            }
            const int kl = w * 32 + j;
            // Lib.metal:216-238 as written
            f4v rec = tile[kTileLifted + kl];  // (center, radius)
            float ocx = o.x - rec.x, ocy = o.y - rec.y, ocz = o.z - rec.z;
            float b = ocx * d.x + ocy * d.y + ocz * d.z;
            float c = (ocx * ocx + ocy * ocy + ocz * ocz) - rec.w * rec.w;
            float disc = b * b - a * c;
            float root, n;
            bool tie;
            if (sphere_root_any_order(disc, b, dv, tbest, nskip, n, root, tie)) {
                const int k = t * kTileSpheres + kl;
                if (!tie || (ibest >= 0 && P.tile_obj[k] < P.tile_obj[ibest])) {
                    tbest = root;
                    ibest = k;
                    nskip = n * (1.0f + 0x1p-22f);
                }
            }
        }
        prof.lap(2);
    }
    return ibest;
}

// closest_sphere for the tiled kind that reads its tiles in place.  The tiles are those of closest_sphere_tiled (spatial
// order, lifted in a frame of their own, ties in distance to the lower object index); what differs is who looks at
// which tile.  Most rays miss most tiles of a large scene: a ray can have a candidate in a tile only if it can touch the
// box around the tile's spheres (ray_may_touch_far_box; see filter_round_narrow for why a ray that fails it has no
// candidate in the box; a tile or group with a sphere that no box can stand for has infinite half extents).  Per group
// of 64 consecutive tiles (consecutive tiles are neighbours in space: they are the leaves of the median splits in order):
//   every lane asks whether its own ray can touch the group's box;
//   lane j takes the box of the group's tile j and is asked about those rays one after the other (or, in a group of
//   fewer tiles than rays, every lane is asked about its own ray, tile after tile);
//   the wave then visits the tiles some ray touches.  For up to kFewRays rays every lane tests its slots of the blocks
//   of 64 that the ray can touch (each block has a box too) exactly, ray after ray; more rays (up to 32) are gathered into one ray tile and the matrix-core filter runs
//   at half width (filter_round_narrow), beyond that at full width, its sphere operands read from global memory --
//   from the L2 and Infinity caches, which hold the whole scene: 196 tiles of 40 KB for 100 000 spheres.
// Nothing is staged and no wave waits for another.
template <bool PROF>
RT_DEV int closest_sphere_direct(const TraceParams &P, Fetch<kFetchTiledDirect> &f, uint32_t *wave_masks, bool active, V3 o, V3 d,
                                 float &tbest, Prof<PROF> &prof, Work &work) {
    const float a = dot(d, d);
    const bool a_ok = !(__builtin_fabsf(a) < 1e-6f);  // Lib.metal:226-228
    const RootDiv dv = root_div(a, active);
    const float s = __builtin_amdgcn_rsqf(a) * kDirScale;
    const V3 dn = v3(d.x * s, d.y * s, d.z * s);
    const int lane = threadIdx.x & 63;
    int ibest = -1;   // slot in tile order of the best hit so far
    float nskip = __builtin_inff();
    const const_f4_ptr frames = (const_f4_ptr)(uintptr_t)P.tile_frames;
    const const_f4_ptr groups = frames + kTileFrameVec4 * P.ntiles;
    const f4v *const block_boxes = (const f4v *)P.tile_frames + kTileFrameVec4 * P.ntiles + 2 * ((P.ntiles + kTileGroup - 1) / kTileGroup);
    const bool live = active && a_ok;
    if (__builtin_amdgcn_ballot_w64(live) == 0) return -1;
    // Where one ray is put to all lanes at once (a lane holding a tile's box, or its slots of a tile's blocks), the ray
    // comes from LDS, every lane reading the same two float4: operands in vector registers (a vector instruction with
    // a scalar-register source issues at half rate), and the broadcast costs no vector instruction.
    f4v *const ray_lds = f.rays;
    if (live) {
        ray_lds[lane] = f4v{o.x, o.y, o.z, dv.a};
        ray_lds[64 + lane] = f4v{d.x, d.y, d.z, dv.r};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t0 = 0; t0 < P.ntiles; t0 += kTileGroup) {
        // the rays that can touch the group's box at all, every lane asking for its own ray
        const uint64_t rays = __builtin_amdgcn_ballot_w64(live && ray_may_touch_far_box(o, d, groups[2 * (t0 / kTileGroup)], groups[2 * (t0 / kTileGroup) + 1]));
        if (rays == 0) continue;
        // which of the group's 64 tiles each of those rays can touch: lane j holds the box of tile t0 + j and is asked about
        // one ray after the other (the ray's numbers broadcast from its lane); it keeps the set of rays that touch its tile
        const int tj = t0 + lane;
        const bool real = tj < P.ntiles;
        const f4v *fr = (const f4v *)P.tile_frames + kTileFrameVec4 * (real ? tj : t0);
        const f4v bm = fr[2], bh = fr[3];
        uint32_t set_lo = 0, set_hi = 0;
        const int ngroup = min(kTileGroup, P.ntiles - t0);
        if (__builtin_popcountll(rays) < ngroup) {
            uint64_t rest = rays;
            int L = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rest));
            f4v ra = ray_lds[L], rb = ray_lds[64 + L];
            while (rest != 0) {
                rest &= rest - 1;
                const int Ln = rest != 0 ? __builtin_amdgcn_readfirstlane(__builtin_ctzll(rest)) : L;
                const f4v na = ray_lds[Ln], nb = ray_lds[64 + Ln];   // the next ray's numbers, asked for before this one's test
                if (real && ray_may_touch_far_box(v3(ra.x, ra.y, ra.z), v3(rb.x, rb.y, rb.z), bm, bh)) {
                    set_lo |= L < 32 ? 1u << L : 0u;
                    set_hi |= L < 32 ? 0u : 1u << (L - 32);
                }
                ra = na;
                rb = nb;
                L = Ln;
            }
        } else {
            // fewer tiles than rays: the other way round, every lane asking about its own ray, tile after tile
            for (int j = 0; j < ngroup; ++j) {
                const uint64_t hit = __builtin_amdgcn_ballot_w64(((rays >> lane) & 1ull) && ray_may_touch_far_box(o, d, frames[kTileFrameVec4 * (t0 + j) + 2], frames[kTileFrameVec4 * (t0 + j) + 3]));
                if (lane == j) {
                    set_lo = (uint32_t)hit;
                    set_hi = (uint32_t)(hit >> 32);
                }
            }
        }
        prof.lap(1);   // (stamped build) the box tests count with the filter
        for (uint64_t tiles = __builtin_amdgcn_ballot_w64((set_lo | set_hi) != 0u); tiles != 0; tiles &= tiles - 1) {
            const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(tiles));
            const int t = t0 + j;
            const uint64_t touching = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)set_hi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)set_lo, j);
            const bool touch = (touching >> lane) & 1ull;
            const f4v fa = frames[kTileFrameVec4 * t], fb = frames[kTileFrameVec4 * t + 1];
            const f4v *tile = f.tiles + (size_t)t * kTileVec4;
            const int n = __builtin_popcountll(touching);
            if (n <= kFewRays) {
                // A few rays against the tile's 512 slots: no filter at all.  Every lane holds the strict records of eight
                // slots (eight coalesced loads, all in flight at once) and evaluates the reference's own discriminant
                // (Lib.metal:216-224, the same operations in the same order) for one ray after the other, the ray's
                // numbers broadcast from its lane; a lane that finds an accepted root offers (root, place of the slot's
                // object index among the tile's, slot) to the ray's 64-bit minimum in LDS, root in the upper half --
                // positive floats order like unsigned integers, so the minimum is the smallest root and, among equal
                // roots, the lowest object index, which is the reference's choice (the places come with the records,
                // 16 bits a slot: no load stands between a root and its offer) (Shaders.metal:17-22; see resolve_balanced for why the root a sphere offers does
                // not depend on the best so far).
                // (Asking for the next visited tile's records while this one is worked on was measured: the second set of
                // registers spills, -29 % with tiles of 512 slots, -6 % with 256.)
                // The tile's slots are ordered so that each block of 64 holds neighbours in space, and each block has a
                // box of its own: lane k holds the box of block k, is asked about the ray, and the wave tests the ray
                // against the blocks it can touch only (one or two of the eight, mostly).
                constexpr int kPerLane = kTileBlocks;
                f4v rec[kPerLane];
#pragma unroll
                for (int i = 0; i < kPerLane; ++i) rec[i] = tile[kTileLifted + 64 * i + lane];
                static_assert(kTileBlocks <= 8, "eight 16-bit places per lane");
                const u4v ranks = ((const u4v *)P.tile_rank)[t * 64 + lane];
                const f4v *kb = block_boxes + 2 * (t * kTileBlocks + (lane & (kTileBlocks - 1)));
                const f4v kbm = kb[0], kbh = kb[1];
                const bool holds_block = lane < ((int)fb.y + 63) / 64;
                unsigned long long *offer = (unsigned long long *)(wave_masks + kBestOffset);
                if (touch) offer[lane] = ~0ull;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (uint64_t rest = touching; rest != 0; rest &= rest - 1) {
                    const int L = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rest));
                    const f4v ra = ray_lds[L], rb = ray_lds[64 + L];
                    const V3 ro = v3(ra.x, ra.y, ra.z), rd = v3(rb.x, rb.y, rb.z);
                    RootDiv q;
                    q.a = ra.w;
                    q.r = rb.w;
                    q.nmin = q.a * 0.0009f;
                    q.slow = dv.slow;
                    const uint32_t blocks = (uint32_t)__builtin_amdgcn_ballot_w64(holds_block && ray_may_touch_far_box(ro, rd, kbm, kbh));
                    if constexpr (PROF) prof.count(7, 64u * (uint32_t)__builtin_popcount(blocks));
#pragma unroll
                    for (int i = 0; i < kPerLane; ++i) {
                        if (((blocks >> i) & 1u) == 0u) continue;   // (wave-uniform)
                        const float ocx = ro.x - rec[i].x, ocy = ro.y - rec[i].y, ocz = ro.z - rec[i].z;
                        const float b = ocx * rd.x + ocy * rd.y + ocz * rd.z;
                        const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - rec[i].w * rec[i].w;
                        const float disc = b * b - q.a * c;
                        if (disc >= 0.0f) {
                            float root, nbest = __builtin_inff();
                            if (sphere_root(disc, b, q, kMaxFloat, nbest, root)) {
                                const uint32_t kl = (uint32_t)(64 * i + lane);   // < 512
                                const uint32_t place = (ranks[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                                atomicMin(&offer[L], ((unsigned long long)__float_as_uint(root) << 32) | (place << 9) | kl);
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (touch) {
                    const unsigned long long key = offer[lane];
                    const float root = __uint_as_float((uint32_t)(key >> 32));
                    const int k = t * kTileSpheres + (int)((uint32_t)key & 511u);
                    // (an equal distance in another tile: the object indices are looked up only then)
                    if (key != ~0ull && (root < tbest || (root == tbest && ibest >= 0 && P.tile_obj[k] < P.tile_obj[ibest]))) {
                        tbest = root;
                        ibest = k;
                        nskip = __builtin_inff();  // (the numerator did not travel: no division is skipped on its account)
                    }
                }
                prof.lap(2);
                continue;
            }
            const LiftFrame F = {fa.x, fa.y, fa.z, fa.w};
            const int count = (int)fb.y;
            uint32_t nz;
            if (count <= kFewSpheres) {
                // a handful of spheres (the tile of the ground sphere, which every ray visits): no filter, every one of them
                // is a candidate of every ray
                wave_masks[lane] = ~(0xFFFFFFFFu >> count);
                nz = count > 0 ? 1u : 0u;
            } else {
            float oo;
            h8 slots[4];
            lift_ray_near_slots(F, fb.z, o, dn, kMfmaSlack, kNearSlack, oo, slots);
            const bool general = __builtin_amdgcn_ballot_w64(touch && !(oo <= fb.x)) != 0;  // wave-uniform
            if (general) {
                RayGeneral y = {o, dn};
                TileStrict ts = {tile + kTileLifted};
                nz = filter_round(ts, 0, (count + 3) / 4, count, wave_masks, lane, y);
            } else {
                __builtin_amdgcn_s_setprio(0);
                const int nblk = (count + 63) / 64;
                if (n <= 32) {
                    h8 b_lo, b_hi;
                    int owner;
                    gather_columns(wave_masks, lane, touching, slots, true, b_lo, b_hi, owner);
                    const uint32_t written = filter_round_narrow<true, true>((const h8 *)tile, 0, nblk, wave_masks, lane, b_lo, b_hi, owner, (lane & 31) < n);
                    nz = scatter_bitmaps(wave_masks, lane, touching, written);
                    work.blocks += (uint32_t)nblk;
                } else {
                    RayLifted lifted;
                    exchange_lifted(slots, lifted);
                    nz = filter_round_matrix<true>((const h8 *)tile, 0, nblk, wave_masks, lane, lifted);
                    work.blocks += 2u * (uint32_t)nblk;
                }
                __builtin_amdgcn_s_setprio(1);
            }
            }
            if (!touch) nz = 0;
            prof.lap(1);
            uint32_t m = 0;
            int w = 0;
            for (;;) {
                if (m == 0) {
                    if (nz == 0) break;
                    w = __builtin_ctz(nz);
                    nz &= nz - 1;
                    m = wave_masks[w * 64 + lane];
                }
                const int j = __builtin_clz(m);
                m &= ~(0x80000000u >> j);
                if constexpr (PROF) {
                    prof.count(6);
                    prof.count(7, (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)));
                }
                const int kl = w * 32 + j;
                // Lib.metal:216-238 as written
                f4v rec = tile[kTileLifted + kl];  // (center, radius)
                float ocx = o.x - rec.x, ocy = o.y - rec.y, ocz = o.z - rec.z;
                float b = ocx * d.x + ocy * d.y + ocz * d.z;
                float c = (ocx * ocx + ocy * ocy + ocz * ocz) - rec.w * rec.w;
                float disc = b * b - a * c;
                float root, n;
                bool tie;
                if (sphere_root_any_order(disc, b, dv, tbest, nskip, n, root, tie)) {
                    const int k = t * kTileSpheres + kl;
                    if (!tie || (ibest >= 0 && P.tile_obj[k] < P.tile_obj[ibest])) {
                        tbest = root;
                        ibest = k;
                        nskip = n * (1.0f + 0x1p-22f);
                    }
                }
            }
            prof.lap(2);
        }
    }
    return ibest;
}

// Closest hit over spheres and quads; fills hit and returns the object index or -1.
template <typename F, bool PROF, bool DENSE = false>
RT_DEV int world_hit(const TraceParams &P, F &f, uint32_t *wave_masks, bool active, V3 o, V3 d, Hit &hit, float &t_out,
                     Prof<PROF> &prof, Work &work) {
    float tbest = kMaxFloat;
    int slot;
    ++work.trips;
    if constexpr (std::is_same_v<F, Fetch<kFetchTiledDirect>>) {
        slot = closest_sphere_direct(P, f, wave_masks, active, o, d, tbest, prof, work);
    } else if constexpr (F::kTiled) {
        slot = closest_sphere_tiled(P, f, wave_masks, active, o, d, tbest, prof, work);
    } else {
        slot = closest_sphere<F, PROF, DENSE>(P, f, wave_masks, active, o, d, tbest, prof, work);
    }
    int obj = -1;
    if (slot >= 0) {
        obj = f.object_of(P, slot);
        f4v s = f.load1(slot);  // (center, radius)
        hit.point = o + tbest * d;                                         // Lib.metal:241
        V3 outward = (hit.point - v3(s.x, s.y, s.z)) / s.w;                // Lib.metal:242
        apply_normals(hit, d, outward);
    }
    for (int q = 0; active && q < P.nquad; ++q) {
        QuadDev quad = P.quads[q];
        float t;
        V3 point, normal;
        if (hit_quad(quad, o, d, tbest, t, point, normal)) {
            // world_hit keeps a later object only when strictly closer (Shaders.metal:21); objects are
            // visited here out of index order, so an equal distance goes to the lower index.
            if (t < tbest || (t == tbest && obj >= 0 && quad.obj < obj)) {
                tbest = t;
                obj = quad.obj;
                hit.point = point;
                apply_normals(hit, d, normal);
            }
        }
    }
    t_out = tbest;
    prof.lap(3);
    return obj;
}

// The parameter block as it lies in the kernel argument segment, behind a pointer the optimiser cannot see
// through: loads of the cold fields stay where they are written instead of being hoisted to kernel entry and
// held in scalar registers across the sphere loop.  The pointer keeps the constant address space, so these loads
// are scalar loads through the constant cache (as generic pointers they were per-lane flat loads, a memory round
// trip each, one after the other).
typedef const __attribute__((address_space(4))) TraceParams *ColdParams;
RT_DEV ColdParams cold_params() {
    ColdParams p = (ColdParams)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
RT_DEV Camera cold_camera(ColdParams C) {
    Camera cam;
    const __attribute__((address_space(4))) float *src = (const __attribute__((address_space(4))) float *)&C->cam;
    float *dst = (float *)&cam;
    for (unsigned i = 0; i < sizeof(Camera) / sizeof(float); ++i) dst[i] = src[i];
    return cam;
}

// Dynamic LDS layout: [candidate words: waves x kMaskWordsPerWave] then, by fetch kind,
//   kFetchLds:    compact records, strict records, each 4 * (ngroups + 1) x 16 B, staged by the workgroup;
//   kFetchStream: 4 waves x 2 tiles x kTileRecords x 16 B;
//   kFetchScalar: nothing;
//   kFetchMatrix: lifted A operands, 4 KiB per 64-sphere block, then strict records 4 * (ngroups + 1) x 16 B, then
//                 (P.stage_mats) the slot -> object table (16-bit) and the material records;
//   kFetchTiled:  two tile buffers of kTileVec4 x 16 B and four spare records (the vector filter reads one group ahead);
//   kFetchTiledDirect: per wave kDirectRayVec4 x 16 B for the rays of its lanes.
template <int KIND>
RT_DEV void stage_scene(const TraceParams &P, uint32_t *lds_words, uint32_t *&wave_masks, Fetch<KIND> &f) {
    const int wave = threadIdx.x >> 6;
    wave_masks = lds_words + wave * kMaskWordsPerWave;
    uint32_t *after_masks = lds_words + (blockDim.x >> 6) * kMaskWordsPerWave;
    if constexpr (KIND == kFetchLds) {
        const int n = 4 * (P.ngroups + 1);
        f4v *c = (f4v *)after_masks;
        f4v *g = c + n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            c[i] = ((const f4v *)P.sphc)[i];
            g[i] = ((const f4v *)P.sphg)[i];
        }
        __syncthreads();
        f.pc = c;
        f.pg = g;
    } else if constexpr (KIND == kFetchMatrix) {
        const int nm = ((P.ngroups + 15) / 16) * 256, n = 4 * (P.ngroups + 1);
        f4v *m = (f4v *)after_masks;
        f4v *g = m + nm;
        for (int i = threadIdx.x; i < nm; i += blockDim.x) m[i] = ((const f4v *)P.sphm)[i];
        for (int i = threadIdx.x; i < n; i += blockDim.x) g[i] = ((const f4v *)P.sphg)[i];
        // sph_obj and the materials ride along when the launch reserved room for them (P.stage_mats)
        uint16_t *ob = (uint16_t *)(g + n);
        MatDev *mt = (MatDev *)(ob + 8 * ((n + 7) / 8));
        if (P.stage_mats) {  // needs nobj <= 65536
            for (int i = threadIdx.x; i < n; i += blockDim.x) ob[i] = (uint16_t)P.sph_obj[i];
            const int nw = P.nobj * (int)(sizeof(MatDev) / 16);
            for (int i = threadIdx.x; i < nw; i += blockDim.x) ((f4v *)mt)[i] = ((const f4v *)P.mats)[i];
        }
        __syncthreads();
        f.pm = (const h8 *)m;
        f.pg = g;
        f.pobj = P.stage_mats ? ob : nullptr;
        f.pmat = P.stage_mats ? mt : nullptr;
    } else if constexpr (KIND == kFetchTiled) {
        f.buf = (f4v *)after_masks;
        f.tiles = (const f4v *)P.tiles;
    } else if constexpr (KIND == kFetchTiledDirect) {
        f.tiles = (const f4v *)P.tiles;
        f.rays = (f4v *)after_masks + wave * kDirectRayVec4;
    } else if constexpr (KIND == kFetchScalar) {
        f.pc = (const_f4_ptr)(uintptr_t)P.sphc;
        f.pg = (const_f4_ptr)(uintptr_t)P.sphg;
        f.strict = (const f4v *)P.sphg;
    } else {
        f.gc = (const f4v *)P.sphc;
        f.gg = (const f4v *)P.sphg;
        f.tile = (f4v *)after_masks + wave * 2 * kTileRecords;
        f.lane = threadIdx.x & 63;
    }
}

}  // namespace rt
