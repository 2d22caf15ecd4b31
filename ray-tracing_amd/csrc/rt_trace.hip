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

#include "rt_hit.hpp"

namespace rt {

// At least four waves per SIMD: at 138 registers (three waves) the LDS-staged vector kind is 4 % slower than at 128 with
// a dozen spilled.
template <int KIND, bool PROF>
__global__ __launch_bounds__(KIND == kFetchMatrix ? 1024 : 256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void trace_kernel(const TraceParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    uint32_t *wave_masks;
    Fetch<KIND> fetch;
    stage_scene<KIND>(P, lds_words, wave_masks, fetch);

    bool alive = true;      // this lane still has, or may still draw, work
    bool need_ray = true;   // the lane's current path is finished (or it has none yet)
    bool have_item = false;
    uint32_t px = 0, py = 0, chunk = 0;
    int ray = 0, ray_end = 0, queued_end = 0;
    V4 acc = v4(0, 0, 0, 0);
    V3 o = v3(0, 0, 0), d = v3(0, 0, 0);
    V4 thr = v4(1, 1, 1, 1), fin = v4(0, 0, 0, 0);
    Seed seed = {0, 0};
    int depth = 0;
    uint32_t nseg = 0;
    Prof<PROF> prof;
    prof.start();
    const int lane = threadIdx.x & 63;
    // this wave's camera-ray queue (wave-uniform, held in scalar registers)
    float4 *const queue = P.rayq + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 * kQueueRays * 2);

    for (;;) {
        prof.count(5);
        // ---- (re)generate: finish items, draw new ones, start the next camera ray ------------------
        // Camera rays are not generated one at a time by the few lanes that happen to need one: when a lane's next
        // sample has no ray yet, the whole wave generates that lane's next samples together -- lane j the sample
        // with index = j (mod 64), up to the end of the item or the next multiple of 64 -- into the lane's slice of
        // a per-wave queue in global memory, and lanes then take their rays from their own slice.
        if (__builtin_amdgcn_ballot_w64(alive && need_ray) != 0) {
            const TraceParams *C = cold_params();
            for (;;) {
                // (1) per lane: retire a finished item, draw the next one
                if (alive && need_ray && (!have_item || ray >= ray_end)) {
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
                    } else {
                        const uint32_t tile_pixels = (uint32_t)C->tileW * (uint32_t)C->tileH;
                        chunk = item / tile_pixels;
                        uint32_t pix = item - chunk * tile_pixels;
                        uint32_t ty = pix / (uint32_t)C->tileW;
                        px = (uint32_t)C->tileX + (pix - ty * (uint32_t)C->tileW);
                        py = (uint32_t)C->tileY + ty;
                        ray = C->ray_begin + (int)chunk * C->chunk;
                        ray_end = min(ray + C->chunk, C->ray_end);
                        queued_end = ray;  // nothing generated yet
                        acc = v4(0, 0, 0, 0);
                        have_item = true;  // an empty range comes back here on the next turn and stores zeros
                    }
                }
                // (2) the wave: generate for every lane whose next sample has no ray yet
                uint64_t fill = __builtin_amdgcn_ballot_w64(alive && need_ray && have_item && ray < ray_end && ray >= queued_end);
                const bool filled = fill != 0;
                while (fill) {
                    const int L = __builtin_ctzll(fill);
                    fill &= fill - 1;
                    const uint32_t bx = (uint32_t)__builtin_amdgcn_readlane((int)px, L), by = (uint32_t)__builtin_amdgcn_readlane((int)py, L);
                    const int r0 = __builtin_amdgcn_readlane(ray, L), re = __builtin_amdgcn_readlane(ray_end, L);
                    const int lo = r0 & ~(kQueueRays - 1);
                    const int bound = min(re, lo + kQueueRays);
                    const int r = lo + lane;
                    if (r >= r0 && r < bound) {
                        // Shaders.metal:121-132
                        Seed s2 = sample_seed(bx, by, C->pass, r);
                        V3 gd, go;
                        get_ray((int)bx, (int)by, C->cam, s2, gd, go);
                        const bool bad = length(gd) < 1e-6f || __builtin_isnan(gd.x) || __builtin_isnan(gd.y) || __builtin_isnan(gd.z);
                        float4 *slot = queue + ((size_t)L * kQueueRays + lane) * 2;
                        slot[0] = make_float4(go.x, go.y, go.z, s2.x);
                        slot[1] = make_float4(gd.x, gd.y, gd.z, bad ? __builtin_nanf("") : s2.y);  // seeds are in [0, 1)
                    }
                    if (lane == L) queued_end = bound;
                }
                if (filled) {
                    // the records were written by other lanes of this wave: order the stores before the loads below
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                // (3) per lane: take the next ray.  (Fetching it a path ahead into LDS was tried and measured: no gain,
                // the other waves of the SIMD already cover this load.)
                if (alive && need_ray && have_item && ray < queued_end) {
                    const float4 *slot = queue + ((size_t)lane * kQueueRays + (ray & (kQueueRays - 1))) * 2;
                    const float4 ra = slot[0], rb = slot[1];
                    ++ray;
                    if (!__builtin_isnan(rb.w)) {  // else: the reference skips such a sample (it adds nothing)
                        o = v3(ra.x, ra.y, ra.z);
                        d = v3(rb.x, rb.y, rb.z);
                        seed.x = ra.w;
                        seed.y = rb.w;
                        thr = v4(1, 1, 1, 1);
                        fin = v4(0, 0, 0, 0);
                        depth = 0;
                        if (P.maxDepth > 0) need_ray = false;  // else color_ray's loop body never runs: the sample is (0,0,0,0)
                    }
                }
                if (__builtin_amdgcn_ballot_w64(alive && need_ray) == 0) break;
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
                MatDev m = fetch.material(P, obj);
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
            for (int i = 0; i < 8; ++i) atomicAdd(C->seg_counter + 1 + i, (unsigned long long)prof.t[i]);
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

// Self-test of the matrix-core filter's guarantee.  Every wave draws a scene frame (scale, translation) and, per
// round, 64 spheres and 64 rays, ray l aimed at the silhouette of sphere l; it lifts them exactly as the host and
// closest_sphere do, runs filter_round_matrix on the block, and checks all 64 x 64 pairs against the reference's
// discriminant.  out: [0] pairs checked (sphere and ray both inside the filter's range), [1] reference candidates,
// [2] matrix-filter candidates, [3] misses (must be 0), [4] pairs skipped because the sphere or the ray origin is
// outside the range (such spheres are always candidates, such rays use the vector filter), [5] out-of-range
// spheres that an in-range ray did not get as a candidate (must be 0).
__global__ __launch_bounds__(256) void matrix_selftest_kernel(uint32_t seed0, int rounds, float slack,
                                                              unsigned long long *out) {
    __shared__ __attribute__((aligned(16))) _Float16 lds_a[4][64 * kLiftSlots];
    __shared__ uint32_t lds_masks[4][2 * 64];
    __shared__ float4 lds_sph[4][64];
    __shared__ int lds_ok[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t st = pcg_hash(seed0 ^ pcg_hash(blockIdx.x * 256u + threadIdx.x));
    auto rnd = [&]() -> float { st = pcg_hash(st); return (float)(st >> 8) * (1.0f / 16777216.0f); };  // [0,1)
    auto sym = [&]() -> float { return 2.0f * rnd() - 1.0f; };
    // wave-uniform frame: drawn by lane 0
    float S = __builtin_ldexpf(1.0f, (int)(rnd() * 24.0f) - 12);  // scene scale 2^-12 .. 2^11
    V3 T = v3(S * 64.0f * sym(), S * 64.0f * sym(), S * 64.0f * sym());
    if (rnd() < 0.3f) T = v3(0.0f, 0.0f, 0.0f);
    S = __shfl(S, 0, 64);
    T.x = __shfl(T.x, 0, 64); T.y = __shfl(T.y, 0, 64); T.z = __shfl(T.z, 0, 64);
    LiftFrame F;
    F.tx = T.x; F.ty = T.y; F.tz = T.z;
    F.sigma = __builtin_ldexpf(1.0f, (int)__builtin_floorf(__builtin_log2f(42.5f / (16.0f * S))));
    unsigned long long n_pairs = 0, n_ref = 0, n_mat = 0, n_miss = 0, n_skip = 0, n_unflagged = 0;
    for (int it = 0; it < rounds; ++it) {
        // sphere `lane`: centre within 16 S of T (a few far outside), radius S * 2^-9 .. 2 S
        const float ext = (rnd() < 0.05f) ? 400.0f : 16.0f;
        const V3 c = T + v3(S * ext * sym(), S * ext * sym(), S * ext * sym());
        const float r = S * __builtin_ldexpf(1.0f + rnd(), -(int)(rnd() * 10.0f));
        // ray `lane`: from distance r/8 .. 2048 r towards a point near the silhouette of the sphere
        V3 u = v3(sym(), sym(), sym());
        float ul = length(u);
        u = ul > 1e-3f ? u / ul : v3(1.0f, 0.0f, 0.0f);
        const float D = r * __builtin_ldexpf(1.0f + rnd(), (int)(rnd() * 14.0f) - 3);
        const V3 o = c + D * u;
        V3 w = v3(sym(), sym(), sym());
        V3 v = cross(u, w);
        float vl = length(v);
        v = vl > 1e-3f ? v / vl : v3(0.0f, 0.0f, 0.0f);
        const int ek = (int)(rnd() * 8.0f);
        const float eps = (ek == 0 ? 0.0f : __builtin_ldexpf(1.0f, -24 + 3 * ek)) * (rnd() < 0.5f ? -1.0f : 1.0f);
        const float along = (rnd() < 0.25f) ? 0.0f : sym() * r;
        const V3 target = c + (r * (1.0f + eps)) * v + along * u;
        V3 d = target - o;
        d = __builtin_ldexpf(1.0f + rnd(), (int)(rnd() * 12.0f) - 6) * d;
        if (rnd() < 0.3f) d = normalize(d);
        const float a = dot(d, d);
        const bool ray_ok = !(__builtin_fabsf(a) < 1e-6f) && a < 1e30f;

        _Float16 slots[kLiftSlots];
        const bool sph_ok = lift_sphere(c.x, c.y, c.z, r, F, (double)slack, slots);
        for (int k = 0; k < kLiftSlots; ++k) lds_a[wave][lift_index(0, lane, k)] = slots[k];
        lds_sph[wave][lane] = make_float4(c.x, c.y, c.z, r);
        lds_ok[wave][lane] = sph_ok ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const float s = __builtin_amdgcn_rsqf(a) * kDirScale;
        const V3 dn = v3(d.x * s, d.y * s, d.z * s);
        float oo;
        RayLifted L;
        lift_ray(F, o, dn, slack, oo, L);
        const bool origin_ok = oo <= 57000.0f;
        filter_round_matrix((const h8 *)lds_a[wave], 0, 1, lds_masks[wave], lane, L);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int j = 0; j < 64; ++j) {
            const float4 sp = lds_sph[wave][j];
            const bool cand = (lds_masks[wave][(j >> 5) * 64 + lane] >> (31 - (j & 31))) & 1u;
            // the reference's discriminant (Lib.metal:216-221)
            const float ocx = o.x - sp.x, ocy = o.y - sp.y, ocz = o.z - sp.z;
            const float b = ocx * d.x + ocy * d.y + ocz * d.z;
            const float cc = (ocx * ocx + ocy * ocy + ocz * ocz) - sp.w * sp.w;
            const float disc = b * b - a * cc;
            const bool ref = !(disc < 0.0f) && !__builtin_isnan(disc);
            if (!ray_ok) continue;
            if (!origin_ok) {
                ++n_skip;
                continue;
            }
            if (!lds_ok[wave][j]) {
                ++n_skip;
                if (!cand) ++n_unflagged;  // an out-of-range sphere must always be a candidate
                continue;
            }
            ++n_pairs;
            n_ref += ref;
            n_mat += cand;
            n_miss += (ref && !cand);
#ifdef RT_SELFTEST_DEBUG
            if (ref && !cand)
                printf("MISS it %d lane %d j %d S %g T %g %g %g sigma %g | c %.9g %.9g %.9g r %.9g | o %.9g %.9g %.9g d %.9g %.9g %.9g | a %g disc %g oo %g\n",
                       it, lane, j, S, T.x, T.y, T.z, F.sigma, sp.x, sp.y, sp.z, sp.w, o.x, o.y, o.z, d.x, d.y, d.z, a, disc, oo);
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    atomicAdd(&out[0], n_pairs);
    atomicAdd(&out[1], n_ref);
    atomicAdd(&out[2], n_mat);
    if (n_miss) atomicAdd(&out[3], n_miss);
    atomicAdd(&out[4], n_skip);
    if (n_unflagged) atomicAdd(&out[5], n_unflagged);
}

// ---- launchers ---------------------------------------------------------------------------------------------

hipError_t launch_matrix_selftest(uint32_t seed, int blocks, int rounds, float slack, unsigned long long *out,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(matrix_selftest_kernel, dim3(blocks), dim3(256), 0, stream, seed, rounds, slack, out);
    return hipGetLastError();
}

hipError_t launch_filter_selftest(uint32_t seed, int blocks, int pairs_per_thread, unsigned long long *out,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(filter_selftest_kernel, dim3(blocks), dim3(256), 0, stream, seed, pairs_per_thread, out);
    return hipGetLastError();
}

size_t trace_lds_bytes(const TraceParams &P, int kind, int threads) {
    size_t bytes = (size_t)(threads / 64) * kMaskWordsPerWave * sizeof(uint32_t);
    if (kind == kFetchLds) bytes += (size_t)8 * (P.ngroups + 1) * sizeof(f4v);
    if (kind == kFetchStream) bytes += (size_t)(threads / 64) * 2 * kTileRecords * sizeof(f4v);
    if (kind == kFetchMatrix) {
        bytes += (size_t)((P.ngroups + 15) / 16) * 4096 + (size_t)4 * (P.ngroups + 1) * sizeof(f4v);
        if (P.stage_mats) bytes += (size_t)8 * ((4 * (P.ngroups + 1) + 7) / 8) * sizeof(uint16_t) + (size_t)P.nobj * sizeof(MatDev);
    }
    return bytes;
}

template <int KIND>
static hipError_t launch_trace_k(const TraceParams &P, bool profile, int blocks, int threads, hipStream_t stream) {
    size_t lds = trace_lds_bytes(P, KIND, threads);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(profile ? (const void *)trace_kernel<KIND, true> : (const void *)trace_kernel<KIND, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (profile)
        hipLaunchKernelGGL((trace_kernel<KIND, true>), dim3(blocks), dim3(threads), lds, stream, P);
    else
        hipLaunchKernelGGL((trace_kernel<KIND, false>), dim3(blocks), dim3(threads), lds, stream, P);
    return hipGetLastError();
}

hipError_t launch_trace(const TraceParams &P, int fetch_kind, bool profile, int blocks, int threads, hipStream_t stream) {
    if (threads != 256 && !(fetch_kind == kFetchMatrix && (threads == 512 || threads == 1024))) return hipErrorInvalidValue;
    switch (fetch_kind) {
        case kFetchLds: return launch_trace_k<kFetchLds>(P, profile, blocks, threads, stream);
        case kFetchScalar: return launch_trace_k<kFetchScalar>(P, profile, blocks, threads, stream);
        case kFetchMatrix: return launch_trace_k<kFetchMatrix>(P, profile, blocks, threads, stream);
        default: return launch_trace_k<kFetchStream>(P, profile, blocks, threads, stream);
    }
}

static int resident_blocks(const TraceParams &P, int fetch_kind, int threads) {
    int n = 0;
    size_t lds = trace_lds_bytes(P, fetch_kind, threads);
    hipError_t e;
    switch (fetch_kind) {
        case kFetchLds: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<kFetchLds, false>, threads, lds); break;
        case kFetchScalar: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<kFetchScalar, false>, threads, lds); break;
        case kFetchMatrix: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<kFetchMatrix, false>, threads, lds); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<kFetchStream, false>, threads, lds); break;
    }
    if (e != hipSuccess || n < 0) n = 0;
    return n;
}

void trace_launch_shape(const TraceParams &P, int fetch_kind, int *threads, int *blocks_per_cu) {
    int best_t = 256, best_n = resident_blocks(P, fetch_kind, 256);
    if (fetch_kind == kFetchMatrix) {
        // one staged copy of the scene per workgroup: larger workgroups keep more waves resident when the copy is big
        for (int t = 512; t <= 1024; t *= 2) {
            int n = resident_blocks(P, fetch_kind, t);
            if (n * t > best_n * best_t) {
                best_n = n;
                best_t = t;
            }
        }
    }
    *threads = best_t;
    *blocks_per_cu = best_n < 1 ? 1 : best_n;
}

hipError_t launch_first_hit(const TraceParams &P, float4 *out4, int *ids, hipStream_t stream) {
    int n = P.width * P.height;
    hipLaunchKernelGGL(first_hit_kernel, dim3((n + 255) / 256), dim3(256), trace_lds_bytes(P, kFetchScalar, 256), stream, P, out4, ids);
    return hipGetLastError();
}

}  // namespace rt
