// rt_trace.hip -- the path-tracing kernel for gfx950 (wave64) and its launchers.
//
// Replaces the reference's Metal compute kernel (reference: "Ray Tracer/Shaders.metal":91-161 with the device
// library "Ray Tracer/Lib.metal").
//
// trace_kernel is a persistent kernel: every lane owns one work item at a time -- (pixel, range of sample
// indices) -- drawn from a global counter, traces the item's samples one ray segment per trip of the main
// loop, and draws the next item when done, so lanes never wait for a neighbour's longer path.
//
// One trip = one closest-hit search per lane (the reference's brute-force world_hit loop over all objects):
//   pass A  every sphere against every lane's ray with a conservative sign test (9 or 11 FMA-class vector
//           instructions per sphere); the sphere records are wave-uniform and are read through the scalar
//           cache into scalar registers (or, optionally, from LDS); one result bit per sphere is kept;
//   pass B  the few spheres whose bit is set are resolved per lane with the reference's exact arithmetic
//           (uncontracted discriminant, sqrt, divisions, strict interval test) in index order;
//   quads, hit record, material scatter, and regeneration of finished paths follow.
// Pass A may only over-report; every hit decision is made in pass B, so results are bit-identical to an
// uncontracted IEEE evaluation of the reference's code.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.hpp"
#include "rt_kernels.h"

namespace rt {

constexpr int kWordGroups = 8;    // groups of 4 spheres per 32-bit candidate word
constexpr int kRoundWords = 16;   // candidate words per resolve round (512 spheres)
constexpr int kMaskWordsPerWave = kRoundWords * 64;  // LDS words per wave: one word per lane per round word

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
enum { kFetchLds = 0, kFetchScalar = 1, kFetchStream = 2 };
constexpr int kTileRecords = 128;  // kFetchStream: records per LDS tile (two tiles per wave)

template <int KIND>
struct Fetch;

template <>
struct Fetch<kFetchLds> {  // both arrays staged once per workgroup in LDS, read with a wave-uniform address (broadcast)
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
};

template <>
struct Fetch<kFetchScalar> {  // scalar cache: uniform address -> s_load_dwordx16 into scalar registers
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
};

// Scenes too large for LDS: every wave streams the record array through its own pair of 128-record LDS tiles.
// Each lane loads two records per tile from global memory (coalesced 16-byte loads, served by L2: all waves read
// the same array) two tiles ahead of use, parks them in registers, and writes them to LDS one tile ahead; pass A
// then broadcast-reads them like the staged kind.  No workgroup barrier is involved: a wave only reads what it
// wrote itself, and LDS operations of one wave execute in order.
template <>
struct Fetch<kFetchStream> {
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
};

// Optional in-kernel cycle stamps (diagnostic instantiation only): wave-level s_memtime differences summed
// per phase.  With PROF = false every member function is empty and nothing is emitted.
template <bool PROF>
struct Prof {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};  // 0 regenerate, 1 filter pass, 2 resolve pass, 3 quads+hit record, 4 shade, 5 trips
    uint64_t last = 0;
    RT_DEV void start() { if constexpr (PROF) last = __builtin_amdgcn_s_memtime(); }
    RT_DEV void lap(int i) {
        if constexpr (PROF) {
            uint64_t now = __builtin_amdgcn_s_memtime();
            t[i] += now - last;
            last = now;
        }
    }
    RT_DEV void count(int i) { if constexpr (PROF) t[i] += 1; }
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
template <typename F, bool PROF>
RT_DEV int closest_sphere(const TraceParams &P, F &f, uint32_t *wave_masks, bool active, V3 o, V3 d, float &tbest,
                          Prof<PROF> &prof) {
    const float a = dot(d, d);
    const bool a_ok = !(__builtin_fabsf(a) < 1e-6f);  // Lib.metal:226-228: such a ray hits no sphere
    const float s = __builtin_amdgcn_rsqf(a) * kDirScale;
    const V3 dn = v3(d.x * s, d.y * s, d.z * s);
    const float oo = __builtin_fmaf(o.z, o.z, __builtin_fmaf(o.y, o.y, o.x * o.x));
    const bool general = __builtin_amdgcn_ballot_w64(active && !(oo <= P.oo_limit)) != 0;  // wave-uniform
    const int lane = threadIdx.x & 63;
    int ibest = -1;
    float nbest = __builtin_inff();  // numerator of the current closest root, see sphere_root
    f.start(general);
    for (int g0 = 0; g0 < P.ngroups; g0 += kRoundWords * kWordGroups) {
        const int g1 = min(P.ngroups, g0 + kRoundWords * kWordGroups);
        uint32_t nz;
        if (general) {
            RayGeneral y = {o, dn};
            nz = filter_round(f, g0, g1, P.nsph, wave_masks, lane, y);
        } else {
            RayCompact y;
            y.ndnx = -dn.x; y.ndny = -dn.y; y.ndnz = -dn.z;
            y.odn = __builtin_fmaf(o.z, dn.z, __builtin_fmaf(o.y, dn.y, o.x * dn.x));
            y.m2ox = -2.0f * o.x; y.m2oy = -2.0f * o.y; y.m2oz = -2.0f * o.z;
            y.oo1 = oo * (1.0f - kCompactSlack);
            nz = filter_round(f, g0, g1, P.nsph, wave_masks, lane, y);
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
            const int k = 4 * g0 + w * 32 + j;
            // Lib.metal:216-238 as written
            f4v rec = f.load1(k);  // (center, radius)
            float ocx = o.x - rec.x, ocy = o.y - rec.y, ocz = o.z - rec.z;
            float b = ocx * d.x + ocy * d.y + ocz * d.z;
            float c = (ocx * ocx + ocy * ocy + ocz * ocz) - rec.w * rec.w;
            float disc = b * b - a * c;
            float root;
            if (sphere_root(disc, b, a, tbest, nbest, root)) {
                tbest = root;
                ibest = k;
            }
        }
        prof.lap(2);
    }
    return ibest;
}

// Closest hit over spheres and quads; fills hit and returns the object index or -1.
template <typename F, bool PROF>
RT_DEV int world_hit(const TraceParams &P, F &f, uint32_t *wave_masks, bool active, V3 o, V3 d, Hit &hit, float &t_out,
                     Prof<PROF> &prof) {
    float tbest = kMaxFloat;
    int slot = closest_sphere(P, f, wave_masks, active, o, d, tbest, prof);
    int obj = -1;
    if (slot >= 0) {
        obj = P.sph_obj[slot];
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
// held in scalar registers across the sphere loop.
RT_DEV const TraceParams *cold_params() {
    const TraceParams *p = (const TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// Dynamic LDS layout: [candidate words: 4 waves x kMaskWordsPerWave] then, by fetch kind,
//   kFetchLds:    compact records, strict records, each 4 * (ngroups + 1) x 16 B, staged by the workgroup;
//   kFetchStream: 4 waves x 2 tiles x kTileRecords x 16 B;
//   kFetchScalar: nothing.
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

template <int KIND, bool PROF>
__global__ __launch_bounds__(256) void trace_kernel(const TraceParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    uint32_t *wave_masks;
    Fetch<KIND> fetch;
    stage_scene<KIND>(P, lds_words, wave_masks, fetch);

    bool alive = true;      // this lane still has, or may still draw, work
    bool need_ray = true;   // the lane's current path is finished (or it has none yet)
    bool have_item = false;
    uint32_t px = 0, py = 0, chunk = 0;
    int ray = 0, ray_end = 0;
    V4 acc = v4(0, 0, 0, 0);
    V3 o = v3(0, 0, 0), d = v3(0, 0, 0);
    V4 thr = v4(1, 1, 1, 1), fin = v4(0, 0, 0, 0);
    Seed seed = {0, 0};
    int depth = 0;
    uint32_t nseg = 0;
    Prof<PROF> prof;
    prof.start();

    for (;;) {
        prof.count(5);
        // ---- (re)generate: finish items, draw new ones, start the next camera ray ------------------
        if (__builtin_amdgcn_ballot_w64(alive && need_ray) != 0) {
            const TraceParams *C = cold_params();
            while (alive && need_ray) {
                if (!have_item || ray >= ray_end) {
                    if (have_item) {
                        size_t p = (size_t)py * C->width + px;
                        if (C->mode == kModeAccum) {
                            C->planes[(size_t)chunk * ((size_t)C->width * C->height) + p] = make_float4(acc.x, acc.y, acc.z, acc.w);
                        } else {
                            // Shaders.metal:142-160
                            float n = (float)C->spp_pass;
                            V4 passAcc = v4(acc.x / n, acc.y / n, acc.z / n, acc.w / n);
                            float4 e = C->target[p];
                            V4 finalColor;
                            if (C->pass == 0) {
                                finalColor = passAcc;
                            } else {
                                float ex = e.x * e.x, ey = e.y * e.y, ez = e.z * e.z;
                                float totalRaysSoFar = (float)(C->pass * C->spp_pass + C->spp_pass);
                                float passWeight = n / totalRaysSoFar;
                                finalColor.x = ex * (1.0f - passWeight) + passAcc.x * passWeight;
                                finalColor.y = ey * (1.0f - passWeight) + passAcc.y * passWeight;
                                finalColor.z = ez * (1.0f - passWeight) + passAcc.z * passWeight;
                                finalColor.w = 1.0f;
                            }
                            V4 w = write_color(finalColor);
                            C->target[p] = make_float4(w.x, w.y, w.z, w.w);
                        }
                        have_item = false;
                    }
                    uint32_t item = atomicAdd(C->counter, 1u);
                    if (item >= C->nitems) {
                        alive = false;
                        break;
                    }
                    const uint32_t tile_pixels = (uint32_t)C->tileW * (uint32_t)C->tileH;
                    chunk = item / tile_pixels;
                    uint32_t pix = item - chunk * tile_pixels;
                    uint32_t ty = pix / (uint32_t)C->tileW;
                    px = (uint32_t)C->tileX + (pix - ty * (uint32_t)C->tileW);
                    py = (uint32_t)C->tileY + ty;
                    ray = C->ray_begin + (int)chunk * C->chunk;
                    ray_end = min(ray + C->chunk, C->ray_end);
                    acc = v4(0, 0, 0, 0);
                    have_item = true;
                    if (ray >= ray_end) continue;  // empty range: store zeros on the next trip
                }
                // Shaders.metal:121-132
                seed = sample_seed(px, py, C->pass, ray);
                ++ray;
                get_ray((int)px, (int)py, C->cam, seed, d, o);
                if (length(d) < 1e-6f || __builtin_isnan(d.x) || __builtin_isnan(d.y) || __builtin_isnan(d.z)) continue;
                thr = v4(1, 1, 1, 1);
                fin = v4(0, 0, 0, 0);
                depth = 0;
                if (P.maxDepth <= 0) continue;  // color_ray's loop body never runs: the sample is (0,0,0,0)
                need_ray = false;
            }
        }
        prof.lap(0);
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;

        // ---- one segment: closest hit (Shaders.metal:38-47); every lane takes part, lanes without work idle ---
        Hit hit;
        float t;
        int obj = world_hit(P, fetch, wave_masks, alive, o, d, hit, t, prof);
        if (alive) {
            ++nseg;
            bool done;
            if (obj < 0) {
                fin = fin + thr * P.sky;                       // Shaders.metal:45
                done = true;
            } else {
                MatDev m = P.mats[obj];
                fin = fin + thr * m.emis;                      // Shaders.metal:51-56
                V4 att;
                if (!scatter(m, d, o, hit, att, seed)) {       // Shaders.metal:60-62
                    done = true;
                } else {
                    thr = thr * att;                           // Shaders.metal:64-65
                    ++depth;
                    done = depth >= P.maxDepth;
                }
            }
            if (done) {
                // Shaders.metal:137-139
                fin.x = clampf(fin.x, 0.0f, 10.0f);
                fin.y = clampf(fin.y, 0.0f, 10.0f);
                fin.z = clampf(fin.z, 0.0f, 10.0f);
                acc = acc + fin;
                need_ray = true;
            }
        }
        prof.lap(4);
    }

    // segment counter: one atomic per wave
    const TraceParams *C = cold_params();
    uint64_t total = nseg;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(C->seg_counter, (unsigned long long)total);
    if constexpr (PROF) {
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < 6; ++i) atomicAdd(C->seg_counter + 1 + i, (unsigned long long)prof.t[i]);
    }
}

// First hit of sample (pass, rayIndex 0) per pixel: {t, nx, ny, nz} and the object index.
__global__ __launch_bounds__(256) void first_hit_kernel(const TraceParams P, float4 *__restrict__ out4,
                                                        int *__restrict__ ids) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    uint32_t *wave_masks;
    Fetch<kFetchScalar> fetch;
    stage_scene<kFetchScalar>(P, lds_words, wave_masks, fetch);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inside = i < P.width * P.height;
    if (!inside) i = 0;  // keep every lane in the wave-level votes of closest_sphere
    uint32_t py = (uint32_t)i / (uint32_t)P.width, px = (uint32_t)i - py * (uint32_t)P.width;
    Seed seed = sample_seed(px, py, P.pass, 0);
    V3 o, d;
    get_ray((int)px, (int)py, P.cam, seed, d, o);
    Hit hit;
    float t;
    Prof<false> prof;
    int obj = world_hit(P, fetch, wave_masks, inside, o, d, hit, t, prof);
    if (!inside) return;
    if (obj >= 0) {
        out4[i] = make_float4(t, hit.normal.x, hit.normal.y, hit.normal.z);
    } else {
        out4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    ids[i] = obj;
}

// Self-test of pass A's guarantee: for pseudo-random, mostly grazing (ray, sphere) pairs over a wide range of
// scales, count pairs whose reference discriminant is >= 0 but which a filter fails to flag.  Both counts must
// be zero.  out: [0] pairs, [1] reference candidates, [2] general-filter candidates, [3] compact-filter
// candidates, [4] general misses, [5] compact misses.
__global__ __launch_bounds__(256) void filter_selftest_kernel(uint32_t seed0, int pairs_per_thread,
                                                              unsigned long long *out) {
    uint32_t st = pcg_hash(seed0 ^ pcg_hash(blockIdx.x * 256u + threadIdx.x));
    auto rnd = [&]() -> float { st = pcg_hash(st); return (float)(st >> 8) * (1.0f / 16777216.0f); };  // [0,1)
    auto sym = [&]() -> float { return 2.0f * rnd() - 1.0f; };
    unsigned long long n_ref = 0, n_gen = 0, n_cmp = 0, miss_gen = 0, miss_cmp = 0;
    for (int it = 0; it < pairs_per_thread; ++it) {
        const float S = __builtin_ldexpf(1.0f, (int)(rnd() * 20.0f) - 8);              // overall scale 2^-8 .. 2^11
        const V3 c = v3(S * 16.0f * sym(), S * 16.0f * sym(), S * 16.0f * sym());
        const float r = S * __builtin_ldexpf(1.0f + rnd(), -(int)(rnd() * 10.0f));     // radius S * 2^-9 .. 2S
        V3 u = v3(sym(), sym(), sym());
        float ul = length(u);
        if (!(ul > 1e-3f)) continue;
        u = u / ul;
        const float D = r * __builtin_ldexpf(1.0f + rnd(), (int)(rnd() * 14.0f) - 3);  // distance r/8 .. 2048 r
        const V3 o = c + D * u;
        // a point near the silhouette: c + v * r * (1 + eps), v perpendicular to u
        V3 w = v3(sym(), sym(), sym());
        V3 v = cross(u, w);
        float vl = length(v);
        if (!(vl > 1e-3f)) continue;
        v = v / vl;
        const int ek = (int)(rnd() * 8.0f);
        const float eps = (ek == 0 ? 0.0f : __builtin_ldexpf(1.0f, -24 + 3 * ek)) * (rnd() < 0.5f ? -1.0f : 1.0f);
        const float along = (rnd() < 0.25f) ? 0.0f : sym() * r;                        // slide the target along the ray
        const V3 target = c + (r * (1.0f + eps)) * v + along * u;
        V3 d = target - o;
        const float lam = __builtin_ldexpf(1.0f + rnd(), (int)(rnd() * 12.0f) - 6);    // non-unit directions too
        d = lam * d;
        if (rnd() < 0.3f) d = normalize(d);

        // the reference's discriminant (Lib.metal:216-221)
        const float ocx = o.x - c.x, ocy = o.y - c.y, ocz = o.z - c.z;
        const float a = dot(d, d);
        if (__builtin_fabsf(a) < 1e-6f || !(a < 1e30f)) continue;
        const float b = ocx * d.x + ocy * d.y + ocz * d.z;
        const float cc = (ocx * ocx + ocy * ocy + ocz * ocz) - r * r;
        const float disc = b * b - a * cc;
        const bool ref = !(disc < 0.0f) && !__builtin_isnan(disc);

        // the filters, exactly as closest_sphere sets them up
        const float s = __builtin_amdgcn_rsqf(a) * kDirScale;
        const V3 dn = v3(d.x * s, d.y * s, d.z * s);
        const float oo = __builtin_fmaf(o.z, o.z, __builtin_fmaf(o.y, o.y, o.x * o.x));
        RayGeneral yg = {o, dn};
        f4v rg = {c.x, c.y, c.z, r};
        const bool gen = (sphere_bit(0u, rg, yg) & 1u) == 0u;
        RayCompact yc;
        yc.ndnx = -dn.x; yc.ndny = -dn.y; yc.ndnz = -dn.z;
        yc.odn = __builtin_fmaf(o.z, dn.z, __builtin_fmaf(o.y, dn.y, o.x * dn.x));
        yc.m2ox = -2.0f * o.x; yc.m2oy = -2.0f * o.y; yc.m2oz = -2.0f * o.z;
        yc.oo1 = oo * (1.0f - kCompactSlack);
        const double slack = (double)kCompactSlack;
        const double ccd = (double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z;
        const double kq = ccd * (1.0 - slack) - (double)(r * r) * (1.0 + slack);
        f4v rc = {c.x, c.y, c.z, __double2float_rd(kq)};
        const bool cmp = (sphere_bit(0u, rc, yc) & 1u) == 0u;

        n_ref += ref; n_gen += gen; n_cmp += cmp;
        miss_gen += (ref && !gen);
        miss_cmp += (ref && !cmp);
    }
    atomicAdd(&out[0], (unsigned long long)pairs_per_thread);
    atomicAdd(&out[1], n_ref);
    atomicAdd(&out[2], n_gen);
    atomicAdd(&out[3], n_cmp);
    if (miss_gen) atomicAdd(&out[4], miss_gen);
    if (miss_cmp) atomicAdd(&out[5], miss_cmp);
}

// ---- launchers ---------------------------------------------------------------------------------------------

hipError_t launch_filter_selftest(uint32_t seed, int blocks, int pairs_per_thread, unsigned long long *out,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(filter_selftest_kernel, dim3(blocks), dim3(256), 0, stream, seed, pairs_per_thread, out);
    return hipGetLastError();
}

static size_t trace_lds_bytes(const TraceParams &P, int kind) {
    size_t bytes = (size_t)4 * kMaskWordsPerWave * sizeof(uint32_t);
    if (kind == kFetchLds) bytes += (size_t)8 * (P.ngroups + 1) * sizeof(f4v);
    if (kind == kFetchStream) bytes += (size_t)4 * 2 * kTileRecords * sizeof(f4v);
    return bytes;
}

template <int KIND>
static hipError_t launch_trace_k(const TraceParams &P, bool profile, int blocks, hipStream_t stream) {
    size_t lds = trace_lds_bytes(P, KIND);
    if (profile)
        hipLaunchKernelGGL((trace_kernel<KIND, true>), dim3(blocks), dim3(256), lds, stream, P);
    else
        hipLaunchKernelGGL((trace_kernel<KIND, false>), dim3(blocks), dim3(256), lds, stream, P);
    return hipGetLastError();
}

hipError_t launch_trace(const TraceParams &P, int fetch_kind, bool profile, int blocks, hipStream_t stream) {
    switch (fetch_kind) {
        case kFetchLds: return launch_trace_k<kFetchLds>(P, profile, blocks, stream);
        case kFetchScalar: return launch_trace_k<kFetchScalar>(P, profile, blocks, stream);
        default: return launch_trace_k<kFetchStream>(P, profile, blocks, stream);
    }
}

int trace_blocks_per_cu(const TraceParams &P, int fetch_kind) {
    int n = 0;
    size_t lds = trace_lds_bytes(P, fetch_kind);
    hipError_t e;
    switch (fetch_kind) {
        case kFetchLds: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<kFetchLds, false>, 256, lds); break;
        case kFetchScalar: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<kFetchScalar, false>, 256, lds); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<kFetchStream, false>, 256, lds); break;
    }
    if (e != hipSuccess || n < 1) n = 1;
    return n;
}

hipError_t launch_first_hit(const TraceParams &P, float4 *out4, int *ids, hipStream_t stream) {
    int n = P.width * P.height;
    hipLaunchKernelGGL(first_hit_kernel, dim3((n + 255) / 256), dim3(256), trace_lds_bytes(P, kFetchScalar), stream, P, out4, ids);
    return hipGetLastError();
}

}  // namespace rt
