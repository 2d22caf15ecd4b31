// rt_wavefront.hip -- wavefront formulation of the same path tracer: the work of trace_kernel split into
// generate / extend / shade kernels over a pool of path slots, with ballot- and atomic-compacted queues between them.
//
// A slot is what a lane is in the persistent kernel: it owns one work item (pixel, range of sample indices) at a
// time and sums that item's samples in order, so the output is bit-identical to the persistent kernel's.  Per
// bounce step:
//   extend   every slot with a live ray: closest hit (same filter + exact resolve as rt_hit.hpp); a miss finishes
//            the sample on the spot; a hit stores the hit record and queues the slot by material kind;
//   shade    the three material queues back to back, so a wave shades (almost always) one material kind;
//            continuing paths go to the next extend queue, finished ones to the generate queue;
//   generate finished slots: next sample of the item, or store the item's sum and draw the next item;
//   rotate   one thread resets the consumed counters.
// Queue order varies from run to run; results do not depend on it (slots are independent, every item's sum goes to
// its own plane cell).
#include "rt_hit.hpp"

namespace rt {

// One 128-byte record per slot (eight float4), so that a slot's state is one or two cache lines instead of eight
// 16-byte gathers from eight arrays:
//   [0] origin, seed.x   [1] direction, seed.y   [2] throughput   [3] radiance so far
//   [4] hit point, object index bits   [5] hit normal, front flag   [6] item sum   [7] meta (as int4)
enum { kRo = 0, kRd = 1, kThr = 2, kFin = 3, kHitA = 4, kHitB = 5, kAcc = 6, kMeta = 7, kSlotVec4 = 8 };
struct WfPool {
    float4 *rec;            // [slots][kSlotVec4]
                            // meta: px | py << 16, chunk, next rayIndex, end rayIndex
    int *depth;             // bounces done; < 0: the slot holds no item
    int *hint;              // sphere slot the slot's path hit last (Work::last_slot of the kernels that prune candidates)
    uint32_t *q_ext[2];     // slots with a live ray, ping-pong
    uint32_t *q_mat[3];     // slots to shade: lambertian (and unknown kinds), metal, dielectric
    uint32_t *q_gen;        // slots whose sample finished
    uint32_t *cnt;          // [0],[1] extend counts, [2..4] material counts, [5] generate count
    int slots;
};

// Append `value` to a queue for every lane with pred set: one atomic per wave, positions by ballot prefix.
RT_DEV void wave_push(uint32_t *queue, uint32_t *counter, bool pred, uint32_t value) {
    const uint64_t mask = __builtin_amdgcn_ballot_w64(pred);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __builtin_ctzll(mask);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(mask));
    base = __shfl(base, leader, 64);
    if (pred) queue[base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = value;
}

// Workgroup-level version for up to four queues at once: the waves add their counts to LDS counters, one thread
// per queue draws the workgroup's range with a single global atomic, then every lane writes its entry.  Cuts the
// same-address global atomics by the number of waves per workgroup.  Must be called by all threads of the block.
struct BlockPush {
    uint32_t *queue[4];
    uint32_t *counter[4];
};
RT_DEV void block_push(const BlockPush &bp, int nq, int which /* -1: none */, uint32_t value) {
    __shared__ uint32_t s_cnt[4], s_base[4];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t wave_off = 0, prefix = 0;
    for (int q = 0; q < nq; ++q) {
        const uint64_t mask = __builtin_amdgcn_ballot_w64(which == q);
        if (mask == 0) continue;
        const int leader = __builtin_ctzll(mask);
        uint32_t off = 0;
        if (lane == leader) off = atomicAdd(&s_cnt[q], (uint32_t)__builtin_popcountll(mask));
        off = __shfl(off, leader, 64);
        if (which == q) {
            wave_off = off;
            prefix = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nq && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(bp.counter[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (which >= 0) bp.queue[which][s_base[which] + wave_off + prefix] = value;
}

// Sample finished: clamp (Shaders.metal:137), add to the item's running sum (Shaders.metal:139).
RT_DEV void finish_sample(const WfPool &W, uint32_t slot, V4 fin) {
    fin.x = clampf(fin.x, 0.0f, 10.0f);
    fin.y = clampf(fin.y, 0.0f, 10.0f);
    fin.z = clampf(fin.z, 0.0f, 10.0f);
    float4 a = W.rec[(size_t)slot * kSlotVec4 + kAcc];
    W.rec[(size_t)slot * kSlotVec4 + kAcc] = make_float4(a.x + fin.x, a.y + fin.y, a.z + fin.z, a.w + fin.w);
}

// Store a finished item's sum exactly as trace_kernel does.
RT_DEV void store_item(const TraceParams &Pc, uint32_t px, uint32_t py, uint32_t chunk, float4 acc) {
    const TraceParams *C = &Pc;
    size_t p = (size_t)py * C->width + px;
    if (C->mode == kModeAccum) {
        const size_t q = (size_t)(py - (uint32_t)C->plane_y0) * (uint32_t)C->plane_w + (px - (uint32_t)C->plane_x0);
        C->planes[(size_t)chunk * C->plane_stride + q] = acc;
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
}

// init != 0: first launch, every slot of the pool starts empty.  Otherwise the slots listed in q_gen.
__global__ __launch_bounds__(256) void wf_generate_kernel(const TraceParams P, const WfPool W, int next, int init) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = init ? (uint32_t)W.slots : W.cnt[5];
    bool live = false;
    uint32_t slot = 0;
    if (i < n) {
        slot = init ? i : W.q_gen[i];
        const TraceParams *C = &P;  // every field is read straight from the argument segment
        int4 m = init ? make_int4(0, 0, 0, 0) : ((const int4 *)W.rec)[(size_t)slot * kSlotVec4 + kMeta];
        bool have_item = !init && W.depth[slot] >= 0;
        uint32_t px = (uint32_t)m.x & 0xFFFFu, py = (uint32_t)m.x >> 16, chunk = (uint32_t)m.y;
        int ray = m.z, ray_end = m.w;
        for (;;) {
            if (!have_item || ray >= ray_end) {
                if (have_item) {
                    store_item(P, px, py, chunk, W.rec[(size_t)slot * kSlotVec4 + kAcc]);
                    have_item = false;
                }
                uint32_t item = atomicAdd(C->counter, 1u);
                if (item >= C->nitems) break;
                const uint32_t tile_pixels = (uint32_t)C->tileW * (uint32_t)C->tileH;
                chunk = item / tile_pixels;
                uint32_t pix = item - chunk * tile_pixels;
                uint32_t ty = pix / (uint32_t)C->tileW;
                px = (uint32_t)C->tileX + (pix - ty * (uint32_t)C->tileW);
                py = (uint32_t)C->tileY + ty;
                chunk_range(C, chunk, ray, ray_end);
                W.rec[(size_t)slot * kSlotVec4 + kAcc] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                have_item = true;
                if (ray >= ray_end) continue;
            }
            // Shaders.metal:121-132
            Seed seed = sample_seed(px, py, C->pass, ray);
            ++ray;
            V3 o, d;
            get_ray((int)px, (int)py, C->cam, seed, d, o);
            if (length(d) < 1e-6f || __builtin_isnan(d.x) || __builtin_isnan(d.y) || __builtin_isnan(d.z)) continue;
            if (P.maxDepth <= 0) continue;
            W.rec[(size_t)slot * kSlotVec4 + kRo] = make_float4(o.x, o.y, o.z, seed.x);
            W.rec[(size_t)slot * kSlotVec4 + kRd] = make_float4(d.x, d.y, d.z, seed.y);
            W.rec[(size_t)slot * kSlotVec4 + kThr] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            W.rec[(size_t)slot * kSlotVec4 + kFin] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            live = true;
            break;
        }
        ((int4 *)W.rec)[(size_t)slot * kSlotVec4 + kMeta] = make_int4((int)(px | (py << 16)), (int)chunk, ray, ray_end);
        W.depth[slot] = have_item ? 0 : -1;
    }
    wave_push(W.q_ext[next], &W.cnt[next], live, slot);
}

// Each workgroup stages the scene once and then takes batches of blockDim.x queue entries in a grid-stride loop.
// DENSE: the instantiation with the pruning exact pass (scenes whose rays cross many nested spheres, rt_hit.hpp).
template <int KIND, bool DENSE>
__global__ __launch_bounds__(KIND == kFetchMatrix ? 1024 : 256) void wf_extend_kernel(const TraceParams P, const WfPool W, int cur) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    const uint32_t n = W.cnt[cur];
    if (blockIdx.x * blockDim.x >= n) return;  // whole workgroup idle (uniform)
    uint32_t *wave_masks;
    Fetch<KIND> fetch;
    stage_scene<KIND>(P, lds_words, wave_masks, fetch);
    uint64_t nseg = 0;
    Work work;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const bool active = i < n;
        const uint32_t slot = active ? W.q_ext[cur][i] : 0u;
        float4 ro = W.rec[(size_t)slot * kSlotVec4 + kRo], rd = W.rec[(size_t)slot * kSlotVec4 + kRd];
        V3 o = v3(ro.x, ro.y, ro.z), d = v3(rd.x, rd.y, rd.z);
        Hit hit;
        float t;
        Prof<false> prof;
        if constexpr (DENSE) work.last_slot = W.hint[slot];
        int obj = world_hit<Fetch<KIND>, false, DENSE>(P, fetch, wave_masks, active, o, d, hit, t, prof, work);
        if constexpr (DENSE) {
            if (active) W.hint[slot] = work.last_slot;
        }
        bool miss = active && obj < 0;
        int kind = -1;
        if (active && obj >= 0) {
            W.rec[(size_t)slot * kSlotVec4 + kHitA] = make_float4(hit.point.x, hit.point.y, hit.point.z, __int_as_float(obj));
            W.rec[(size_t)slot * kSlotVec4 + kHitB] = make_float4(hit.normal.x, hit.normal.y, hit.normal.z, hit.front ? 1.0f : 0.0f);
            int type = fetch.material(P, obj).type;
            kind = type == 1 ? 1 : (type == 2 ? 2 : 0);
        }
        if (miss) {
            float4 thr = W.rec[(size_t)slot * kSlotVec4 + kThr], fin = W.rec[(size_t)slot * kSlotVec4 + kFin];
            V4 f = v4(fin.x, fin.y, fin.z, fin.w) + v4(thr.x, thr.y, thr.z, thr.w) * P.sky;  // Shaders.metal:45
            finish_sample(W, slot, f);
        }
        BlockPush bp = {{W.q_mat[0], W.q_mat[1], W.q_mat[2], W.q_gen}, {&W.cnt[2], &W.cnt[3], &W.cnt[4], &W.cnt[5]}};
        block_push(bp, 4, miss ? 3 : kind, slot);
        nseg += active ? 1 : 0;
    }
    // segments and executed-work counters: summed over the workgroup in LDS, one global atomic each per workgroup (a
    // bounce step launches thousands of waves, and their atomics would all go to the same three addresses)
    for (int off = 32; off > 0; off >>= 1) nseg += __shfl_down(nseg, off, 64);
    __shared__ unsigned long long s_sum[3];
    if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        if (nseg) atomicAdd(&s_sum[0], (unsigned long long)nseg);
        atomicAdd(&s_sum[1], (unsigned long long)work.trips);
        atomicAdd(&s_sum[2], (unsigned long long)work.blocks);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_sum[0]) atomicAdd(P.seg_counter, s_sum[0]);
    if (threadIdx.x == 1) atomicAdd(P.seg_counter + 9, s_sum[1]);
    if (threadIdx.x == 2) atomicAdd(P.seg_counter + 10, s_sum[2]);
}

__global__ __launch_bounds__(256) void wf_shade_kernel(const TraceParams P, const WfPool W, int next) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t c0 = W.cnt[2], c1 = W.cnt[3], c2 = W.cnt[4];
    if (blockIdx.x * 256u >= c0 + c1 + c2) return;
    bool cont = false, done = false;
    uint32_t slot = 0;
    if (i < c0 + c1 + c2) {
        slot = i < c0 ? W.q_mat[0][i] : (i < c0 + c1 ? W.q_mat[1][i - c0] : W.q_mat[2][i - c0 - c1]);
        float4 ro = W.rec[(size_t)slot * kSlotVec4 + kRo], rd = W.rec[(size_t)slot * kSlotVec4 + kRd], ha = W.rec[(size_t)slot * kSlotVec4 + kHitA], hb = W.rec[(size_t)slot * kSlotVec4 + kHitB];
        float4 t4 = W.rec[(size_t)slot * kSlotVec4 + kThr], f4 = W.rec[(size_t)slot * kSlotVec4 + kFin];
        V3 o = v3(ro.x, ro.y, ro.z), d = v3(rd.x, rd.y, rd.z);
        Seed seed = {ro.w, rd.w};
        Hit hit;
        hit.point = v3(ha.x, ha.y, ha.z);
        hit.normal = v3(hb.x, hb.y, hb.z);
        hit.front = hb.w != 0.0f;
        const int obj = __float_as_int(ha.w);
        V4 thr = v4(t4.x, t4.y, t4.z, t4.w), fin = v4(f4.x, f4.y, f4.z, f4.w);
        MatDev m = P.mats[obj];
        fin = fin + thr * m.emis;                      // Shaders.metal:51-56
        V4 att;
        int depth = W.depth[slot];
        if (!scatter(m, d, o, hit, att, seed)) {       // Shaders.metal:60-62
            done = true;
        } else {
            thr = thr * att;                           // Shaders.metal:64-65
            ++depth;
            done = depth >= P.maxDepth;
        }
        if (done) {
            finish_sample(W, slot, fin);
        } else {
            cont = true;
            W.rec[(size_t)slot * kSlotVec4 + kRo] = make_float4(o.x, o.y, o.z, seed.x);
            W.rec[(size_t)slot * kSlotVec4 + kRd] = make_float4(d.x, d.y, d.z, seed.y);
            W.rec[(size_t)slot * kSlotVec4 + kThr] = make_float4(thr.x, thr.y, thr.z, thr.w);
            W.rec[(size_t)slot * kSlotVec4 + kFin] = make_float4(fin.x, fin.y, fin.z, fin.w);
            W.depth[slot] = depth;
        }
    }
    BlockPush bp = {{W.q_ext[next], W.q_gen, nullptr, nullptr}, {&W.cnt[next], &W.cnt[5], nullptr, nullptr}};
    block_push(bp, 2, cont ? 0 : (done ? 1 : -1), slot);
}

__global__ void wf_rotate_kernel(uint32_t *cnt, int cur) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        cnt[cur] = 0;
        cnt[2] = 0; cnt[3] = 0; cnt[4] = 0; cnt[5] = 0;
    }
}

// ---- host side -------------------------------------------------------------------------------------------

size_t wavefront_pool_bytes(int slots) {
    // 7 float4 + 1 int4 + 2 int + 6 queues of uint32 per slot, plus the counters
    return (size_t)slots * (8 * 16 + 2 * 4 + 6 * 4) + 256;
}

static WfPool carve_pool(void *base, int slots) {
    WfPool W;
    char *p = (char *)base;
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) / 256 * 256; return (void *)q; };
    W.cnt = (uint32_t *)take(256);
    W.rec = (float4 *)take((size_t)slots * 16 * kSlotVec4);
    W.depth = (int *)take((size_t)slots * 4);
    W.hint = (int *)take((size_t)slots * 4);
    W.q_ext[0] = (uint32_t *)take((size_t)slots * 4);
    W.q_ext[1] = (uint32_t *)take((size_t)slots * 4);
    W.q_mat[0] = (uint32_t *)take((size_t)slots * 4);
    W.q_mat[1] = (uint32_t *)take((size_t)slots * 4);
    W.q_mat[2] = (uint32_t *)take((size_t)slots * 4);
    W.q_gen = (uint32_t *)take((size_t)slots * 4);
    W.slots = slots;
    return W;
}

size_t wavefront_pool_alloc_bytes(int slots) { return wavefront_pool_bytes(slots) + 16 * 256; }

// Runs the whole launch: returns when every item is done (it has to read the queue counter back to know).
// pool: device memory of wavefront_pool_alloc_bytes(slots) bytes.  steps_out: bounce steps executed.
hipError_t run_wavefront(const TraceParams &P, int fetch_kind, int extend_threads, void *pool, int slots, int extend_blocks,
                         hipStream_t stream, uint32_t *host_cnt, int *steps_out) {
    WfPool W = carve_pool(pool, slots);
    const int blocks = (slots + 255) / 256;
    const int ext_need = (slots + extend_threads - 1) / extend_threads;
    const int ext_blocks = extend_blocks > 0 && extend_blocks < ext_need ? extend_blocks : ext_need;
    hipError_t e = hipMemsetAsync(W.cnt, 0, 256, stream);
    if (e != hipSuccess) return e;
    if (e == hipSuccess) e = hipMemsetAsync(W.hint, 0, (size_t)slots * 4, stream);
    if (e != hipSuccess) return e;
    const size_t lds = trace_lds_bytes(P, fetch_kind, extend_threads);
    const bool dense = P.dense_kernel != 0;
    typedef void (*extend_fn)(const TraceParams, const WfPool, int);
    const extend_fn extend = fetch_kind == kFetchMatrix ? (dense ? wf_extend_kernel<kFetchMatrix, true> : wf_extend_kernel<kFetchMatrix, false>)
                             : fetch_kind == kFetchLds  ? (dense ? wf_extend_kernel<kFetchLds, true> : wf_extend_kernel<kFetchLds, false>)
                                                        : (dense ? wf_extend_kernel<kFetchScalar, true> : wf_extend_kernel<kFetchScalar, false>);
    if (lds > 64 * 1024) {
        e = hipFuncSetAttribute((const void *)extend, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int ext_threads = fetch_kind == kFetchMatrix ? extend_threads : 256;
    int cur = 0;
    hipLaunchKernelGGL(wf_generate_kernel, dim3(blocks), dim3(256), 0, stream, P, W, cur, 1);
    int steps = 0;
    if (!host_cnt) return hipErrorInvalidValue;
    for (;;) {
        const int next = cur ^ 1;
        hipLaunchKernelGGL(extend, dim3(ext_blocks), dim3(ext_threads), lds, stream, P, W, cur);
        hipLaunchKernelGGL(wf_shade_kernel, dim3(blocks), dim3(256), 0, stream, P, W, next);
        hipLaunchKernelGGL(wf_generate_kernel, dim3(blocks), dim3(256), 0, stream, P, W, next, 0);
        hipLaunchKernelGGL(wf_rotate_kernel, dim3(1), dim3(64), 0, stream, W.cnt, cur);
        cur = next;
        ++steps;
        if ((steps & 7) == 0) {
            e = hipMemcpyAsync(host_cnt, W.cnt, 32, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) break;
            if (host_cnt[cur] == 0) break;
        }
        if (steps > (1 << 28)) { e = hipErrorUnknown; break; }
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (steps_out) *steps_out = steps;
    return e;
}

}  // namespace rt
