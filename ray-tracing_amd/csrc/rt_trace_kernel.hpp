// rt_trace_kernel.hpp -- the persistent path-tracing kernel (a template over the record fetcher and the forms of its
// phases), shared by the two translation units that instantiate it: rt_trace.hip (the kernels most scenes run) and
// rt_trace_dense.hip (the instantiations that carry the pruning exact pass, the stamped diagnostic ones, and the
// vector-filter kinds for scenes of many spheres).  They are
// compiled with different instruction-scheduling strategies (csrc/FLAGS).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_hit.hpp"

namespace rt {

// At least four waves per SIMD: at 138 registers (three waves) the LDS-staged vector kind is 4 % slower than at 128 with
// a dozen spilled.
// SINGLE: every work item is one sample (the progressive flow's tiles, rt_render_tile): each lane generates the camera
// ray of its own item, all lanes that need one at once; the other instantiation has the whole wave generate a lane's
// samples 64 at a time.
// DENSE: the closest-hit search carries the lane-by-lane form of the exact pass for rounds whose candidate words are
// dense (scenes of nested spheres); the kernels without it are the ones every other scene runs.
// REGROUP (staged matrix kind): material scatter is not run by the lane that owns the path.  Every lane whose ray hit a
// scattering surface posts a request -- ray direction, hit record, seed, object -- into its wave's LDS area (the
// candidate words are dead by then) and an entry into one of two index lists of the workgroup: diffuse and metal from
// the bottom, dielectric from the top.  After a workgroup barrier thread i of the workgroup serves list entry i, so a
// wave runs one branch of the material code over 64 requests of one kind instead of every branch over its own mixed
// lanes; the answer (new ray, seed, scattered or absorbed) goes back through the same slot and is taken by its owner
// after a second barrier.  The waves of a workgroup therefore take every trip of the main loop together until none
// of them has work left.  Which lane evaluates a request changes no operation on it: results are identical.
#ifndef RT_SHARE_GIVEUP
#define RT_SHARE_GIVEUP 1
#endif
constexpr int kShareIdle = 16;  // lanes without work from which a sharing wave draws new items whatever its lanes could still give
struct RegroupLds {
    uint32_t alive[2];    // by trip parity: some wave still has, or may still draw, work
    uint32_t lo[2];       // requests appended from the bottom of idx (diffuse, metal)
    uint32_t hi[2];       // requests appended from the top of idx (dielectric)
    uint32_t pad[2];
    uint16_t idx[1024];   // wave * 64 + lane of the request a thread serves
};

// BUILD: nothing in the kernel depends on it; it keeps apart instantiations that two translation units compile with
// different flags (same template arguments would make them one symbol, and one of the two builds would be dropped).
#ifndef RT_DIRECT_WAVES
#define RT_DIRECT_WAVES 4
#endif
template <int KIND, bool PROF, bool SINGLE = false, bool DENSE = false, bool REGROUP = false, int BUILD = 0>
__global__ __launch_bounds__(KIND == kFetchMatrix || KIND == kFetchTiled ? 1024 : 256) __attribute__((amdgpu_waves_per_eu(KIND == kFetchTiledDirect ? RT_DIRECT_WAVES : 4, 8)))
void trace_kernel(const TraceParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    uint32_t *wave_masks;
    Fetch<KIND> fetch;
    stage_scene<KIND>(P, lds_words, wave_masks, fetch);
    static_assert(!REGROUP || (KIND == kFetchMatrix && !SINGLE && !DENSE), "the regrouped form exists for the staged matrix kind");
    [[maybe_unused]] RegroupLds *rg = nullptr;
    [[maybe_unused]] uint32_t par = 0;
    if constexpr (REGROUP) {
        rg = (RegroupLds *)(lds_words + (cold_params()->regroup_offset >> 2));
        if (threadIdx.x < 8) ((uint32_t *)rg)[threadIdx.x] = 0;
        __syncthreads();
    }

    bool alive = true;      // this lane's wave still has, or may still draw, work
    bool need_ray = true;   // the lane's current path is finished (or it has none yet)
    bool have_item = false;
    bool exhausted = false; // (wave-uniform) the work counter has run past the last item: nothing more to draw
    [[maybe_unused]] bool sharing = false;  // (wave-uniform) the wave's lanes work its items off together (2b)
    [[maybe_unused]] uint64_t lent = 0;     // (wave-uniform) slices of the queue that samples of another lane may lie in (5)
    [[maybe_unused]] uint32_t share_turns = 0, share_starved = 0;  // (wave-uniform) turns spent sharing, and with many lanes idle (5)
    [[maybe_unused]] bool gave_up = false;  // (wave-uniform) sharing did not keep the lanes busy: not again in this launch
    // chunk: number of the partial-sum plane the lane is adding into (TraceParams::planes); while a request for an item
    // is under way, the register the counter's answer arrives in
    uint32_t px = 0, py = 0, chunk = 0;
    int ray = 0, ray_end = 0, queued_end = 0;
    // whose slice of the wave's queue the lane's camera rays lie in: its own, or, for samples taken over from
    // another lane once the counter is exhausted, that lane's
    [[maybe_unused]] int qlane = threadIdx.x & 63;
    V4 acc = v4(0, 0, 0, 0);
    V3 o = v3(0, 0, 0), d = v3(0, 0, 0);
    V4 thr = v4(1, 1, 1, 1), fin = v4(0, 0, 0, 0);
    Seed seed = {0, 0};
    int depth = 0;
    // (DENSE) the sphere slot this lane's last camera ray hit first: where the next camera ray's search starts looking
    // (Work::last_slot); the samples of a work item share their pixel
    [[maybe_unused]] int first_slot = 0;
    uint32_t nseg = 0;
    Prof<PROF> prof;
    prof.start();
    Work work;
    const int lane = threadIdx.x & 63;
    // this wave's camera-ray queue (wave-uniform, held in scalar registers)
    float4 *const queue = P.rayq + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 * kQueueRays * 2);

    uint64_t asked = 0;      // (wave-uniform) lanes that asked for an item on the previous turn; the number of the
                             // first of them arrives in `chunk` of the lowest such lane, the others follow on
    bool generated = false;  // (wave-uniform) camera rays were written to the queue during the previous turn

    for (;;) {
        prof.count(5);
        // ---- (re)generate: finish items, draw new ones, start the next camera ray ------------------
        // Nothing here waits for memory it has just asked for: a lane that needs an item asks for its number with
        // an atomic and reads the answer on the next turn; camera rays are generated into a queue on one turn and
        // taken from it on the next.  The lane sits out the turns in between (two per item) while the other lanes
        // of the wave trace, instead of the whole wave stalling on the round trips.
        // Camera rays are not generated one at a time by the few lanes that happen to need one: when a lane's next
        // sample has no ray yet, the whole wave generates that lane's next samples together -- lane j the sample
        // with index = j (mod 64), up to the end of the item or the next multiple of 64 -- into the lane's slice of
        // a per-wave queue in global memory, and lanes then take their rays from their own slice.
        if (__builtin_amdgcn_ballot_w64(alive && need_ray) != 0) {
            const ColdParams C = cold_params();
            // (1) the records written on the previous turn came from other lanes of this wave: order those stores
            // before the loads below
            if (generated) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                generated = false;
            }
            // (2) per lane: the item asked for on the previous turn.  One lane asked for all of them.
            if (asked) {
                const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)chunk, __builtin_ctzll(asked));
                const bool pending = (asked >> lane) & 1;
                const uint32_t item = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(asked >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)asked, 0u));
                asked = 0;
                exhausted = exhausted || __builtin_amdgcn_ballot_w64(pending && item >= C->nitems) != 0;
                if constexpr (!SINGLE) {
                    if (!gave_up) sharing = sharing || base + 64u > C->share_begin;
                }
                if (pending && item < C->nitems) {
                    const uint32_t tile_pixels = (uint32_t)C->tileW * (uint32_t)C->tileH;
                    chunk = item / tile_pixels;
                    uint32_t pix = item - chunk * tile_pixels;
                    uint32_t ty = pix / (uint32_t)C->tileW;
                    px = (uint32_t)C->tileX + (pix - ty * (uint32_t)C->tileW);
                    py = (uint32_t)C->tileY + ty;
                    chunk_range(C, chunk, ray, ray_end);
                    // the items of the last chunks are cut into finer planes: the plane of the item's first sample
                    if (ray >= C->fine_begin) chunk = (uint32_t)C->fine_plane0 + ((uint32_t)(ray - C->fine_begin) >> C->sub_shift);
                    queued_end = ray;  // nothing generated yet
                    acc = v4(0, 0, 0, 0);
                    have_item = true;  // an empty range is retired below and stores zeros
                    if constexpr (!SINGLE) qlane = lane;
                }
            }
            // (2b) Towards the end of a launch (TraceParams::share_begin, and in any case once the counter is exhausted) a
            // lane without work takes over the upper half of the untouched planes of the lane of its wave that has the
            // most of them (the items drawn last are cut into several planes for this, see TraceParams::fine_begin:
            // every plane is a sum of its own, so which lane adds it up changes nothing).  The camera rays are there
            // already, in the slice of the wave's queue that the item was generated into (qlane travels with the
            // samples).  One hand-over per turn.  This is what keeps the last work items of a launch -- up to 64 samples
            // of one pixel, thousands of turns when its paths are long -- from being finished by one lane each while the
            // other 63 of the wave, and soon the whole GPU, have run dry; and since the wave draws new items only when
            // nothing is left to share (5), it holds little work when the counter runs out, whatever its items were.
            [[maybe_unused]] bool shared_now = false;  // (wave-uniform) a lane had planes to give on this turn
            if constexpr (!SINGLE) {
                if ((sharing || exhausted) && C->fine_begin < C->ray_end) {
                    const uint64_t idle = __builtin_amdgcn_ballot_w64(alive && need_ray && !have_item);
                    if (idle) {
                        const int shift = C->sub_shift, rb = C->fine_begin;
                        // first plane the lane has not begun (a lane between two samples is about to begin the one at `ray`)
                        int f = rb + ((((ray - rb) + (1 << shift) - 1) >> shift) << shift);
                        if (need_ray && f == ray) f += 1 << shift;
                        // planes it can give: its camera rays exist, and the item lies in the fine planes
                        const bool gives = have_item && ray >= rb && queued_end >= ray_end && f < ray_end;
                        const int spare = gives ? (((ray_end - 1) - f) >> shift) + 1 : 0;
                        int key = spare ? (spare << 6) | lane : 0;
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) key = max(key, __shfl_xor(key, off, 64));
                        key = __builtin_amdgcn_readfirstlane(key);
                        if (key) {
                            shared_now = true;
                            const int V = key & 63, T = __builtin_ctzll(idle), n = key >> 6;
                            const int v_end = __builtin_amdgcn_readlane(ray_end, V);
                            const int cut = __builtin_amdgcn_readlane(f, V) + ((n - ((n + 1) >> 1)) << shift);  // the giver keeps [.., cut)
                            const uint32_t vx = (uint32_t)__builtin_amdgcn_readlane((int)px, V), vy = (uint32_t)__builtin_amdgcn_readlane((int)py, V);
                            const int vq = __builtin_amdgcn_readlane(qlane, V);
                            if (vq != T) lent |= 1ull << vq;
                            if (lane == V) {
                                ray_end = cut;
                                queued_end = cut;
                            }
                            if (lane == T) {
                                px = vx;
                                py = vy;
                                ray = cut;
                                ray_end = v_end;
                                queued_end = v_end;
                                chunk = (uint32_t)C->fine_plane0 + ((uint32_t)(cut - rb) >> shift);
                                qlane = vq;
                                acc = v4(0, 0, 0, 0);
                                have_item = true;
                            }
                        }
                    }
                }
            }
            // (3) per lane: take the next ray.  (Fetching it a path ahead into LDS was tried and measured: no gain,
            // the other waves of the SIMD already cover this load.)
            if (alive && need_ray && have_item && ray < queued_end) {
                const float4 *slot;
                if constexpr (SINGLE) {
                    slot = queue + ((size_t)lane * kQueueRays + (ray & (kQueueRays - 1))) * 2;
                } else {
                    slot = queue + ((size_t)qlane * kQueueRays + (ray & (kQueueRays - 1))) * 2;
                    // planes finer than items: a sample that begins a new plane sends the finished one off first
                    if (ray >= C->fine_begin) {
                        const uint32_t plane = (uint32_t)C->fine_plane0 + ((uint32_t)(ray - C->fine_begin) >> C->sub_shift);
                        if (plane != chunk) {
                            const size_t q = (size_t)(py - (uint32_t)C->plane_y0) * (uint32_t)C->plane_w + (px - (uint32_t)C->plane_x0);
                            C->planes[(size_t)chunk * C->plane_stride + q] = make_float4(acc.x, acc.y, acc.z, acc.w);
                            acc = v4(0, 0, 0, 0);
                            chunk = plane;
                        }
                    }
                }
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
                    if constexpr (DENSE) work.last_slot = first_slot;
                    if (P.maxDepth > 0) need_ray = false;  // else color_ray's loop body never runs: the sample is (0,0,0,0)
                }
            }
            // (4) the wave: generate for every lane whose next sample has no ray yet
            uint64_t fill = __builtin_amdgcn_ballot_w64(alive && need_ray && have_item && ray < ray_end && ray >= queued_end);
            generated = fill != 0;
            Camera cam;
            int pass = 0;
            if (fill) {
                cam = cold_camera(C);
                pass = C->pass;
            }
            if constexpr (SINGLE) {
                if ((fill >> lane) & 1) {
                    Seed s2 = sample_seed(px, py, pass, ray);
                    V3 gd, go;
                    get_ray((int)px, (int)py, cam, s2, gd, go);
                    const bool bad = length(gd) < 1e-6f || __builtin_isnan(gd.x) || __builtin_isnan(gd.y) || __builtin_isnan(gd.z);
                    float4 *slot = queue + ((size_t)lane * kQueueRays + (ray & (kQueueRays - 1))) * 2;
                    slot[0] = make_float4(go.x, go.y, go.z, s2.x);
                    slot[1] = make_float4(gd.x, gd.y, gd.z, bad ? __builtin_nanf("") : s2.y);
                    queued_end = ray + 1;
                }
            } else {
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
                        Seed s2 = sample_seed(bx, by, pass, r);
                        V3 gd, go;
                        get_ray((int)bx, (int)by, cam, s2, gd, go);
                        const bool bad = length(gd) < 1e-6f || __builtin_isnan(gd.x) || __builtin_isnan(gd.y) || __builtin_isnan(gd.z);
                        float4 *slot = queue + ((size_t)L * kQueueRays + lane) * 2;
                        slot[0] = make_float4(go.x, go.y, go.z, s2.x);
                        slot[1] = make_float4(gd.x, gd.y, gd.z, bad ? __builtin_nanf("") : s2.y);  // seeds are in [0, 1)
                    }
                    if (lane == L) queued_end = bound;
                }
            }
            // (5) per lane: retire a finished item and ask for the next one
            const bool retire = alive && need_ray && have_item && ray >= ray_end;
            bool ask = alive && need_ray && !exhausted && (!have_item || ray >= ray_end);
            if constexpr (!SINGLE) {
                if (sharing && !exhausted) {
                    // While the lanes share, a new item is drawn only when no lane had planes to give on this turn, and by
                    // lanes that were without work then (a lane that retires its samples now looks for planes on the next
                    // turn first).  Hand-overs are one per turn: where paths are so short that the lanes finish planes
                    // faster than that (kShareIdle lanes without work on one turn in eight, after 32 turns), the wave goes
                    // back to drawing an item per lane until the counter is exhausted.
                    ask = ask && !have_item && !shared_now;
                    ++share_turns;
                    if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(alive && need_ray && !have_item)) >= kShareIdle) ++share_starved;
                    if (RT_SHARE_GIVEUP && share_turns >= 32u && share_starved * 8u > share_turns) {
                        sharing = false;
                        gave_up = true;
                    }
                }
                if (lent != 0) {
                    // A new item's camera rays go to the lane's own slice of the queue, so no other lane's samples may
                    // still lie in it: a slice that was never lent is free; of the others, one per turn is looked at.
                    const uint64_t cand = __builtin_amdgcn_ballot_w64(ask);
                    uint64_t askers = cand & ~lent;
                    const uint64_t doubt = cand & lent;
                    if (doubt) {
                        const int L = __builtin_ctzll(doubt);
                        if (__builtin_amdgcn_ballot_w64(have_item && ray < ray_end && qlane == L) == 0) {
                            lent &= ~(1ull << L);
                            askers |= 1ull << L;
                        }
                    }
                    ask = (askers >> lane) & 1;
                }
            }
            if (retire) {
                {
                    size_t p = (size_t)py * C->width + px;
                    if (C->mode == kModeAccum) {
                        const size_t q = (size_t)(py - (uint32_t)C->plane_y0) * (uint32_t)C->plane_w + (px - (uint32_t)C->plane_x0);
                        C->planes[(size_t)chunk * C->plane_stride + q] = make_float4(acc.x, acc.y, acc.z, acc.w);
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
            }
            // ... the lowest of those lanes for all of them, and the answer is not read before the next turn.  (The
            // address is made to look lane-dependent, or the compiler rewrites the atomic into its own
            // one-lane-asks-and-broadcasts form, which waits for the answer on the spot.)
            asked = __builtin_amdgcn_ballot_w64(ask);
            if (asked && lane == __builtin_ctzll(asked)) {
                uint32_t zero = 0;
                asm volatile("" : "+v"(zero));
                chunk = atomicAdd(C->counter + zero, (uint32_t)__builtin_popcountll(asked));
            }
            // the wave is through when the counter is exhausted and no lane holds an item or a path any more
            if (exhausted && asked == 0 && __builtin_amdgcn_ballot_w64(alive && (have_item || !need_ray)) == 0) alive = false;
        }
        prof.lap(0);
        if constexpr (REGROUP) {
            const int wave = threadIdx.x >> 6;
            const int nthreads = blockDim.x;
            if (__builtin_amdgcn_ballot_w64(alive) != 0 && lane == 0) rg->alive[par] = 1;
            // ---- one segment: closest hit (Shaders.metal:38-47) ----
            Hit hit;
            float t;
            int obj = -1;
            const bool tracing = alive && !need_ray;
            if (__builtin_amdgcn_ballot_w64(tracing) != 0)
                obj = world_hit<Fetch<KIND>, PROF, DENSE>(P, fetch, wave_masks, tracing, o, d, hit, t, prof, work);
            // ---- the owner's part before the scatter: miss, emission, and which list the request goes to ----
            bool done = false, wants = false, glass = false;
            V4 att = v4(1.0f, 1.0f, 1.0f, 1.0f);
            if (tracing) {
                ++nseg;
                if (obj < 0) {
                    fin = fin + thr * P.sky;                       // Shaders.metal:45
                    done = true;
                } else {
                    MatDev m = fetch.material(P, obj);
                    fin = fin + thr * m.emis;                      // Shaders.metal:51-56
                    wants = m.type == 0 || m.type == 1 || m.type == 2;   // anything else absorbs (Lib.metal:203-213)
                    glass = m.type == 2;
                    if (!glass) att = m.albedo;
                    done = !wants;                                 // Shaders.metal:60-62
                }
            }
            float4 *const slots = (float4 *)wave_masks;   // this wave's requests: [3][64] float4
            {
                const uint64_t blo = __builtin_amdgcn_ballot_w64(wants && !glass), bhi = __builtin_amdgcn_ballot_w64(wants && glass);
                uint32_t base_lo = 0, base_hi = 0;
                if (lane == 0) {
                    if (blo) base_lo = atomicAdd(&rg->lo[par], (uint32_t)__builtin_popcountll(blo));
                    if (bhi) base_hi = atomicAdd(&rg->hi[par], (uint32_t)__builtin_popcountll(bhi));
                }
                base_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_lo);
                base_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_hi);
                if (wants) {
                    const uint64_t mine = glass ? bhi : blo;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
                    const uint32_t pos = glass ? (uint32_t)nthreads - 1u - (base_hi + rank) : base_lo + rank;
                    rg->idx[pos] = (uint16_t)(wave * 64 + lane);
                    slots[lane] = make_float4(d.x, d.y, d.z, seed.x);
                    slots[64 + lane] = make_float4(hit.normal.x, hit.normal.y, hit.normal.z, seed.y);
                    slots[128 + lane] = make_float4(hit.point.x, hit.point.y, hit.point.z,
                                                    __uint_as_float((uint32_t)obj | (hit.front ? 0x80000000u : 0u)));
                }
            }
            prof.lap(4);   // (stamped build) owner's part before the scatter and the request
            __syncthreads();
            prof.lap(3);   // ... waiting for the other waves counts with "quads + hit record"
            if (rg->alive[par] == 0) break;   // the same answer in every wave
            if (threadIdx.x == 0) {
                rg->alive[par ^ 1] = 0;
                rg->lo[par ^ 1] = 0;
                rg->hi[par ^ 1] = 0;
            }
            // ---- serve: thread i the request of list entry i ----
            {
                const uint32_t nlo = rg->lo[par], nhi = rg->hi[par];
                const uint32_t i = threadIdx.x;
                if (i < nlo || i >= (uint32_t)nthreads - nhi) {
                    const uint32_t e = rg->idx[i];
                    float4 *const rs = (float4 *)(lds_words + (e >> 6) * kMaskWordsPerWave) + (e & 63u);
                    const float4 q0 = rs[0], q1 = rs[64], q2 = rs[128];
                    const uint32_t tag = __float_as_uint(q2.w);
                    const MatDev m = fetch.material(P, (int)(tag & 0x7FFFFFFFu));
                    Hit h;
                    h.point = v3(q2.x, q2.y, q2.z);
                    h.normal = v3(q1.x, q1.y, q1.z);
                    h.front = (tag >> 31) != 0u;
                    V3 rd = v3(q0.x, q0.y, q0.z), ro = v3(0.0f, 0.0f, 0.0f);
                    Seed sd = {q0.w, q1.w};
                    V4 unused;
                    const bool ok = scatter(m, rd, ro, h, unused, sd);
                    rs[0] = make_float4(rd.x, rd.y, rd.z, sd.x);
                    rs[64] = make_float4(ro.x, ro.y, ro.z, sd.y);
                    ((float *)(rs + 128))[0] = ok ? 1.0f : 0.0f;
                }
            }
            prof.lap(4);   // serving requests
            __syncthreads();
            prof.lap(3);
            if (wants) {
                const float4 a0 = slots[lane], a1 = slots[64 + lane];
                const float okf = ((const float *)(slots + 128 + lane))[0];
                d = v3(a0.x, a0.y, a0.z);
                o = v3(a1.x, a1.y, a1.z);
                seed.x = a0.w;
                seed.y = a1.w;
                if (okf == 0.0f) {
                    done = true;
                } else {
                    thr = thr * att;                               // Shaders.metal:64-65
                    ++depth;
                    done = depth >= P.maxDepth;
                }
            }
            if (tracing && done) {
                // Shaders.metal:137-139
                fin.x = clampf(fin.x, 0.0f, 10.0f);
                fin.y = clampf(fin.y, 0.0f, 10.0f);
                fin.z = clampf(fin.z, 0.0f, 10.0f);
                acc = acc + fin;
                need_ray = true;
            }
            par ^= 1u;
            prof.lap(4);
            continue;
        }
        if constexpr (KIND == kFetchTiled) {
            // the waves of a workgroup walk the tiles together (barriers inside world_hit): all of them take every
            // turn, with or without rays, until none of them has work left; this barrier also separates the tile
            // buffers' last readers of the previous turn from the first copy of this one
            if (!__syncthreads_or(alive ? 1 : 0)) break;
        } else {
            if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
            if (__builtin_amdgcn_ballot_w64(alive && !need_ray) == 0) continue;  // no lane has a ray on this turn
        }

        // ---- one segment: closest hit (Shaders.metal:38-47); every lane takes part, lanes without work idle ---
        Hit hit;
        float t;
        const bool tracing = alive && !need_ray;
        int obj = world_hit<Fetch<KIND>, PROF, DENSE>(P, fetch, wave_masks, tracing, o, d, hit, t, prof, work);
        if (tracing) {
            ++nseg;
            if constexpr (DENSE) {
                if (depth == 0) first_slot = work.last_slot;
            }
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
    const ColdParams C = cold_params();
    uint64_t total = nseg;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(C->seg_counter, (unsigned long long)total);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(C->seg_counter + 9, (unsigned long long)work.trips);
        atomicAdd(C->seg_counter + 10, (unsigned long long)work.blocks);
    }
    if constexpr (PROF) {
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(C->seg_counter + 1 + i, (unsigned long long)prof.t[i]);
    }
}

// The instantiations of rt_trace_dense.hip: DENSE (fetch kinds other than the tiled one) and PROF ...
const void *trace_kernel_dense(int fetch_kind, bool profile, bool single);
// ... and the plain kernels of the vector-filter kinds for scenes of kManySpheres spheres or more, where the filter loop
// is most of a trip (the register-minimising schedule of rt_trace.hip costs that loop 11 %, and gains the same kernels
// 3 to 10 % on scenes of a few objects).  Null when there is none.
const void *trace_kernel_many(int fetch_kind, bool single);
constexpr int kManySpheres = 48;

}  // namespace rt
