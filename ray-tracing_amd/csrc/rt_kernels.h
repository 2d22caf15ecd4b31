// rt_kernels.h -- kernel parameter block and launcher declarations shared by the .hip files.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.hpp"
#include "rt_lift.hpp"

namespace rt {

enum { kModeAccum = 0, kModeProgressive = 1 };
constexpr int kBoundVec4 = 10;  // float4 of bounding balls per word of 32 sphere slots (TraceParams::word_bounds)
constexpr int kWideWords = 4;  // words of 32 sphere slots that may hold spheres outside the box of the half-width filter (TraceParams::wide_word)
constexpr int kQueueRays = 64;  // camera rays per lane in the trace kernel's per-wave queue (two float4 each)

// Tiled matrix-core kind (TraceParams::tiles, rt_hit.hpp closest_sphere_tiled).
#ifndef RT_TILE_SPHERES
#define RT_TILE_SPHERES 512
#endif
constexpr int kTileSpheres = RT_TILE_SPHERES;          // sphere slots per tile (a power of two from 64 to 512: at most one resolve round)
static_assert(kTileSpheres >= 64 && kTileSpheres <= 512 && (kTileSpheres & (kTileSpheres - 1)) == 0, "tile size");
constexpr int kTileLifted = kTileSpheres * 4;          // float4 of lifted A operands per tile (64 bytes per sphere)
constexpr int kTileVec4 = kTileLifted + kTileSpheres;  // ... followed by the strict records
constexpr int kTileFrameVec4 = 4;                      // float4 per tile in TraceParams::tile_frames
constexpr int kTileGroup = 64;                         // consecutive tiles that share a box of their own (rt_hit.hpp closest_sphere_direct)
constexpr int kTileBlocks = kTileSpheres / 64;         // blocks of 64 consecutive slots of a tile, each with a box of its own

// Everything one trace launch needs.  Passed by value in the kernel argument segment.  The first block is
// read on every trip of the kernel's main loop and lives in scalar registers; the kernel reads the second
// block (camera, work-item bookkeeping, output buffers) from the argument segment only where it is used, so
// it does not occupy registers during the sphere loop.
struct TraceParams {
    // ---- hot ----
    const float4 *sphg;     // [nslots] strict records (cx, cy, cz, r): exact hit test and the general filter
                            //                 (padding: r = NaN, which can never hit)
    const float4 *sphc;     // [nslots] compact filter records (cx, cy, cz, |c|^2 * (1 - 2^-16) - r*r * (1 + 2^-16))
    const int *sph_obj;     // [nslots] object index of each sphere slot; nslots = 4 * ngroups rounded up to a
                            //          multiple of 128, plus 256 spare never-hit slots (read-ahead of the fetchers)
    const QuadDev *quads;   // [nquad]
    const MatDev *mats;     // [nobj] indexed by object index
    int ngroups;            // sphere slots / 4 (slots are padded to a multiple of 4 with never-hit records)
    int nsph;               // real spheres
    int nquad;
    float oo_limit;         // rays with |origin|^2 above this use the general filter (see rt_hit.hpp); for the
                            // matrix-core filter the origin is measured in the lifted frame
    const void *sphm;       // matrix-core filter: lifted A operands, 4 KiB per 64-sphere block (rt_lift.hpp)
    LiftFrame lift;         // translation and scale of the lifted frame
    const float4 *tiles;    // tiled matrix-core kind: per tile of 512 sphere slots (up to 512 spheres that lie close
                            // together), 2048 float4 of lifted A operands in the tile's own frame, then the 512
                            // strict records
    const float4 *tile_frames;  // per tile (tx, ty, tz, sigma), (range limit in the tile's frame, spheres,
                                // largest |c'| of a filtered sphere, 0), (centre of the box around the tile's spheres,
                                // length of its half diagonal), (its half extents x (1 + 2^-12), 0); a tile with a sphere
                                // no box can stand for: infinite half extents.  After the tiles: the same two float4 (box
                                // centre and half diagonal, half extents) for every group of kTileGroup consecutive tiles,
                                // and after those for each of the kTileBlocks blocks of 64 slots of every tile (a tile's
                                // slots are ordered so that a block's spheres are neighbours in space)
    const int *tile_obj;    // [ntiles * 512] object index of each slot in tile order (-1: padding)
    const uint16_t *tile_rank;  // [ntiles * 512] per tile 512 entries, for lane l the entries 8 l + i: the place of the
                                // object index of slot 64 i + l among the tile's object indices in ascending order
                                // (what equal distances inside a tile are decided by; closest_sphere_direct)
    int ntiles;
    int stage_mats;         // matrix-core kind: sph_obj and mats are staged in LDS too (the launch reserved the room)
    int nobj;               // objects (length of mats)
    int resolve_mode;       // exact pass over the candidates: 0 chosen per wave and round, 1 always pooled over the wave,
                            // 2 always per lane (rt_hit.hpp closest_sphere) ...
    int dense_kernel;       // ... in the kernel instantiation that carries the per-lane form (launch_trace picks it when
                            // this is set: scenes whose rays cross many spheres, or resolve_mode 2)
    const float4 *word_bounds;  // kBoundVec4 float4 per word of 32 sphere slots: balls around and inside the word's spheres and
                                // around and inside each of its four granules of 8 (rt_api.hip group_bounds)
    const float4 *round_runs;   // two float4 per resolve round of 512 slots: (centre, bits of first slot | end slot << 16 of the run)
                                // and (bits of the round's words of that run | its words with exempt slots << 16, 2 if the radii
                                // fall along the slots else 3, 0, 0); no words when the round's words of runs are not all of one run
    // Staged matrix kind at half width (rt_hit.hpp filter_round_narrow; rt_api.hip narrow_setup).  narrow != 0: a ray whose
    // origin lies within narrow_oo (|o'|^2 in the lifted frame) and that cannot touch the box (cull_mid, cull_half) has
    // no candidate among the spheres the box stands for.  The others -- the spheres of bits wide_test[k] (tested with the
    // general vector filter) and wide_always[k] (candidates for every ray) of word wide_word[k] = slot / 32, bit 31 - j
    // for slot 32 w + j -- are looked at for every ray.
    int narrow;
    float narrow_oo;
    V3 cull_mid, cull_half;
    int wide_words;
    uint32_t wide_word[kWideWords], wide_test[kWideWords], wide_always[kWideWords];
    float4 *rayq;           // per-wave queues of generated camera rays: grid lanes x kQueueRays x 2 float4
    int maxDepth;           // Uniforms.maxRayDepth
    V4 sky;                 // Uniforms.globalIllumation
    // ---- cold ----
    Camera cam;
    int width, height;      // render target size
    int tileX, tileY, tileW, tileH;
    int pass;               // Uniforms.currentSample
    int ray_begin, ray_end; // rayIndex range traced by this launch
    int chunk;              // samples per work item of the first `nbig` chunks of a pixel's sample range ...
    int nbig, chunk_small;  // ... and of the chunks after them: items are drawn chunk by chunk, so the launch ends on
                            // short items and the persistent grid drains evenly
    uint32_t nitems;        // tileW * tileH * number of chunks
    uint32_t *counter;      // work-item counter, zeroed before the launch
    unsigned long long *seg_counter;  // [0] += number of closest-hit searches; [1..8] += phase cycle stamps and
                                      // counts of the diagnostic instantiation; [9] += closest-hit searches counted
                                      // per wave (idle lanes ride along); [10] += 64-sphere blocks the waves put
                                      // through the matrix-core filter
    float4 *planes;         // kModeAccum: [nchunks][plane_stride] partial sums, one plane per chunk; pixel (x, y) of a
                            // plane is at (y - plane_y0) * plane_w + (x - plane_x0): planes cover the whole target
                            // (0, 0, width) or just the tile of the launch
    int plane_x0, plane_y0, plane_w;
    unsigned long long plane_stride;
    float4 *target;         // kModeProgressive: rgba32Float render target
    int mode;
    int spp_pass;           // kModeProgressive: Uniforms.sampleCount
    // kModeAccum: plane of a sample.  Work items before sample index fine_begin have a plane each, numbered like their
    // chunks; from fine_begin on (the items that are drawn last) a plane holds 2^sub_shift consecutive samples, plane
    // fine_plane0 + ((sample - fine_begin) >> sub_shift), several per item, so that lanes without work can take whole
    // planes over from other lanes once the work counter is exhausted.  fine_begin >= ray_end: no such planes.
    int sub_shift, fine_begin, fine_plane0;
    // Work items from number share_begin on (inside the fine planes, about the last item per resident lane) are worked
    // off by the lanes of a wave together (RT_OPT_SHARE 0): a lane without work takes the upper half of another lane's
    // untouched planes before the wave draws a new item, so that a wave holds a few items' worth of samples when the
    // counter runs out instead of half an item per lane.  0xFFFFFFFF (the default): only once the counter is exhausted.
    uint32_t share_begin;
    int regroup;            // staged matrix kind: material scatter regrouped over the workgroup (trace_kernel<..., REGROUP>)
    int regroup_offset;     // ... and where in the dynamic LDS its lists lie (bytes; set by launch_trace)
};

// Sample range [ray, ray_end) of chunk number `chunk` of a pixel.
template <class ParamsPtr>
__device__ __forceinline__ void chunk_range(ParamsPtr C, uint32_t chunk, int &ray, int &ray_end) {
    const bool big = (int)chunk < C->nbig;
    const int start = big ? (int)chunk * C->chunk : C->nbig * C->chunk + ((int)chunk - C->nbig) * C->chunk_small;
    ray = C->ray_begin + start;
    ray_end = min(ray + (big ? C->chunk : C->chunk_small), C->ray_end);
}

// fetch_kind: 0 records staged in LDS per workgroup, 1 scalar cache, 2 streamed through per-wave LDS tiles,
// 3 staged in LDS with the matrix-core filter, 4 lifted tiles streamed through LDS per workgroup with the
// matrix-core filter.  threads: workgroup size (256; kind 3 also 512 or 1024; kind 4 only 1024).
hipError_t launch_trace(const TraceParams &P, int fetch_kind, bool profile, int blocks, int threads, hipStream_t stream);
// Workgroup size and resident workgroups per compute unit that give the most resident waves.
void trace_launch_shape(const TraceParams &P, int fetch_kind, int *threads, int *blocks_per_cu);
size_t trace_lds_bytes(const TraceParams &P, int fetch_kind, int threads);
// Wavefront formulation (rt_wavefront.hip): blocks until the launch is complete.
size_t wavefront_pool_alloc_bytes(int slots);
// fetch_kind 0, 1 or 3; extend_threads: workgroup size of the extend kernel (256, or what trace_launch_shape gives for 3).
// host_cnt: at least 32 bytes of pinned host memory for the queue counters read back between bounce steps.
hipError_t run_wavefront(const TraceParams &P, int fetch_kind, int extend_threads, void *pool, int slots, int extend_blocks,
                         hipStream_t stream, uint32_t *host_cnt, int *steps_out);
hipError_t launch_first_hit(const TraceParams &P, float4 *out4, int *ids, hipStream_t stream);
hipError_t launch_filter_selftest(uint32_t seed, int blocks, int pairs_per_thread, unsigned long long *out,
                                  hipStream_t stream);
hipError_t launch_matrix_selftest(uint32_t seed, int blocks, int rounds, float slack, int far, unsigned long long *out,
                                  hipStream_t stream);
// sph: nsph strict records (x, y, z, radius) in slot order; bounds: their bounding balls (TraceParams::word_bounds)
hipError_t launch_prune_selftest(uint32_t seed, int blocks, int rays_per_thread, const float4 *sph, const float4 *bounds, int nsph,
                                 unsigned long long *out, hipStream_t stream);
// the box of the half-width filter (P.narrow_oo, cull_mid, cull_half, wide_*, lift) against the exact hit test of every sphere
hipError_t launch_box_selftest(uint32_t seed, int blocks, int rays_per_thread, const float4 *sph, int nsph, const TraceParams &P,
                               unsigned long long *out, hipStream_t stream);
// the boxes of the tiles and tile groups (P.tiles, P.tile_frames, P.ntiles), their allowances scaled by `scale`
hipError_t launch_tile_box_selftest(uint32_t seed, int blocks, int rays_per_thread, const TraceParams &P, float scale,
                                    unsigned long long *out, hipStream_t stream);
hipError_t launch_fold(float4 *accum, const float4 *planes, int nplanes, int width, int height, int tileX, int tileY,
                       int tileW, int tileH, hipStream_t stream);
// The reference's per-pixel epilogue for one pass over a tile (Shaders.metal:142-160) on tile-sized planes that hold one
// sample each: sequential sum in sample order, mean, progressive blend with the target, write_color.
hipError_t launch_blend_tile(float4 *target, const float4 *planes, int nplanes, int width, int tileX, int tileY, int tileW,
                             int tileH, int pass, int spp_pass, hipStream_t stream);
hipError_t launch_resolve(float4 *target, const float4 *accum, int npix, int spp, hipStream_t stream);
hipError_t launch_fill_rect(float4 *target, int width, int height, int x0, int y0, int w, int h, float4 value,
                            hipStream_t stream);
hipError_t launch_error_pixels(float4 *target, int width, int height, int gw, int gh, int tileX, int tileY, int tileW,
                               int tileH, int all_fail, hipStream_t stream);
hipError_t launch_present_bgra8(const float4 *target, int width, int height, uint32_t *out, int out_w, int out_h,
                                hipStream_t stream);

}  // namespace rt
