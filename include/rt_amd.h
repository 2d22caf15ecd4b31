/*
 * rt_amd.h -- C-ABI of the MI355X path-tracing library (librt_amd.so).
 *
 * The reference has no plugin or FFI interface: its boundary is the Metal binding
 * contract of one compute kernel (texture(0) = rgba32Float render target,
 * buffer(0) = one `Uniforms`, buffer(1) = `Object[objCount]`; reference:
 * "Ray Tracer/Shaders.metal":91-94) and the Swift methods of `MetalRenderer`
 * that drive it (reference: "Ray Tracer/ContentView.swift":232-752).  Each entry
 * point below names the reference call it replaces.  Struct layouts are those of
 * the reference's bridging header (see rt_abi.h).
 *
 * All functions return RT_OK (0) or a negative RT_ERR_* code; rt_last_error()
 * gives a message for the last failure on that context.  Host pointers passed in
 * are copied before the call returns (the reference re-sends uniforms and objects
 * with setBytes on every dispatch, ContentView.swift:689-696); nothing is retained.
 * A context is used from one host thread at a time, like the reference's renderer,
 * which is driven from the main run loop only (ContentView.swift:617-619).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define RT_API
#else
#define RT_API __attribute__((visibility("default")))
#endif

typedef struct rt_ctx rt_ctx;

enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,    /* bad argument */
    RT_ERR_NO_DEVICE = -2,  /* no usable GPU / HIP runtime failure at create */
    RT_ERR_HIP = -3,        /* a HIP call failed; see rt_last_error */
    RT_ERR_NO_TARGET = -4,  /* rt_set_target has not been called */
    RT_ERR_NO_SCENE = -5,   /* no objects given and none resident */
    RT_ERR_TOO_LARGE = -6   /* size beyond what the entry point supports */
};

/* Option keys for rt_set_option. */
enum {
    RT_OPT_CHUNK = 1,       /* samples per work item of the batch path; 0 = all samples of a call in one
                               item, which reproduces the reference's sequential per-pixel sum order;
                               -1 (default) = automatic: 1 to 64, from the samples per lane of the GPU and the
                               scene size */
    RT_OPT_KERNEL = 3,      /* 0 = persistent kernel (default); 1 = wavefront formulation: generate / extend / shade
                               kernels over a pool of path slots with compacted queues (same results; rt_render then
                               returns only when the launch has finished); 2 = the persistent kernel with in-kernel
                               cycle stamps per phase (diagnostic; fills rt_stats.phase_cycles) */
    RT_OPT_SCENE_FETCH = 4, /* where the sphere filter reads its records:
                               0 = automatic (default): staged in LDS once per workgroup and broadcast-read up to
                                   1024 spheres (1792 with the vector filters); beyond that, 3 where the scene allows
                                   it, else 1,
                               1 = through the scalar cache into scalar registers (vector filter),
                               2 = streamed from global memory through per-wave LDS tiles (vector filter),
                               3 = spheres sorted in space and cut into tiles of 512, each lifted in a frame of its
                                   own (any scene size).  With RT_OPT_NARROW 1 (default) the tiles are read where they
                                   lie: a wave visits only the tiles its rays can touch (a box per tile and per 64
                                   tiles), tests a few rays per tile exactly and runs the matrix-core filter for more;
                                   with RT_OPT_NARROW 0 every workgroup streams all tiles through two LDS buffers and
                                   runs the matrix-core filter on them */
    RT_OPT_BLOCKS_PER_CU = 5, /* persistent grid size = this x compute units; 0 = occupancy query */
    RT_OPT_FILTER = 6,      /* which conservative sphere filter precedes the exact hit test; results are identical:
                               0 = automatic (default): the matrix-core filter for LDS-staged scenes of 48 to 1024
                                   spheres that its FP16 range covers and, tile by tile, for larger scenes
                                   (RT_OPT_SCENE_FETCH 3); else the compact 9-instruction vector filter,
                               1 = always the translation-invariant 12-instruction vector filter,
                               2 = vector filters only (compact where the scene allows it),
                               3 = the matrix-core filter for any staged scene it covers, however small */
    RT_OPT_WAVEFRONT_SLOTS = 7, /* path slots of the wavefront pool (default 2^20; 168 bytes of device memory each) */
    RT_OPT_TAIL_CHUNK = 8,  /* the last samples of every pixel are handed out in items of this many samples, after all
                               longer items, so that the persistent grid drains evenly; -1 (default) and 0 = no such
                               tail (measured on the cover scene: it costs more in camera-ray generation than the
                               even drain gains) */
    RT_OPT_TAIL_SAMPLES = 9, /* samples per pixel in that tail; -1 (default) = automatic */
    RT_OPT_RESOLVE = 10,    /* how the exact hit test walks the candidates the sphere filter left (LDS-staged scenes;
                               results are identical): 0 = automatic (default): pooled over the wave, and for scenes
                               whose lines cross many spheres (nested shells; decided at scene upload) candidates that
                               cannot win are struck first (bounding balls per 32 and 8 consecutive spheres, then a
                               test per sphere) and dense candidate words go lane by lane, four candidates in flight;
                               1 = always pooled; 2 = always struck and lane by lane */
    RT_OPT_PLANE_SAMPLES = 12, /* rt_render / rt_render_noalloc sum the samples of a work item one after the other into a
                               partial sum (a "plane" per item) and add the planes in order.  The items that are drawn
                               last are cut into planes of this many samples instead: once the work counter is exhausted,
                               lanes without work take whole planes over from the lane of their wave that has the most
                               left, so the launch does not end on a few lanes finishing 64-sample items.  -1 (default)
                               = automatic (4 when RT_OPT_CHUNK is automatic and gives items of 8 samples or more, else 0); 0 = never; else a power of two that
                               divides the item size.  rt_stats reports the resulting schedule of the sums */
    RT_OPT_SHARE = 13,      /* when the lanes of a wave begin to take planes over from each other (RT_OPT_PLANE_SAMPLES):
                               1 (default) = once the work counter is exhausted; 0 = from about one work item per resident
                               lane before that -- a wave then draws a new item only when none of its lanes has planes left
                               to give, and holds a few items' worth of samples at the end of the launch instead of half an
                               item per lane; a wave whose lanes finish planes faster than one hand-over per turn can serve
                               them (short paths) goes back to an item per lane.  Results are identical.  Measured: one
                               rank's share of the benchmark frame at 8 ranks +0.4 %, nested shells +1.3 %, Cornell box -4 %,
                               hence not the default */
    RT_OPT_NARROW = 14,     /* 1 (default) = rays that cannot touch the box around a group of spheres are left out of its sphere
                               filter.  Staged matrix-core kind (scenes of 128 spheres or more with at most 8 spheres much
                               larger than the rest): where at most 32 rays of a wave can touch the box around the ordinary
                               spheres, the filter runs on those rays alone, at half width.  Tiled kind: see
                               RT_OPT_SCENE_FETCH 3.  0 = every ray against every sphere.  Results are identical */
    RT_OPT_REGROUP = 11     /* material scatter of the staged matrix kind's batch kernels (results are identical):
                               0 = automatic (default), 1 = every lane scatters its own path, 2 = requests are sorted by
                               material over the workgroup through LDS, so that a wave runs one branch of the material
                               code; the waves of a workgroup then take every trip of the main loop together */
};

/* setupMetal (ContentView.swift:328-356): device, queue, pipelines.  `device` is the HIP device ordinal.
 * `stream` is a hipStream_t the library should enqueue on, or NULL to create its own (non-blocking) stream.
 * The legacy default stream has the null handle and therefore cannot be passed.  Whatever else touches the
 * buffers handed to rt_render / rt_resolve (a collective over the accumulation buffers, say) must run on, or be
 * ordered against, the context's stream: pass a stream you created and enqueue that work on it. */
RT_API int rt_create(rt_ctx **out, int device, void *stream);
RT_API void rt_destroy(rt_ctx *ctx);
RT_API const char *rt_last_error(const rt_ctx *ctx);
RT_API int rt_set_option(rt_ctx *ctx, int key, int value);
RT_API int rt_get_option(const rt_ctx *ctx, int key, int *value);

/* setupTexture(width:height:) (ContentView.swift:358-378): allocate the rgba32Float render target and
 * fill it with (0,0,0,1). */
RT_API int rt_set_target(rt_ctx *ctx, int width, int height);
/* clearTexture() (ContentView.swift:661-664). */
RT_API int rt_clear(rt_ctx *ctx);
/* updateTexture() (ContentView.swift:509-518): replace the whole target with host pixels (rgba float). */
RT_API int rt_write_target(rt_ctx *ctx, const float *rgba);
/* No counterpart (the reference never reads the texture back): copy the target to host, rgba float,
 * row 0 first. */
RT_API int rt_read_target(rt_ctx *ctx, float *rgba);

/* runComputeShaderForTile(tile:sampleIndex:) (ContentView.swift:666-709) + computeShader
 * (Shaders.metal:91-161).  The caller fills currentSample, objCount and the tile rectangle of *u exactly
 * as the Swift method does (:678-687).  One pass of u->sampleCount samples per pixel of the tile is traced
 * and blended into the target with the reference's gamma-encoded running mean (Shaders.metal:144-160).
 * Blocks until the GPU is done, like commit + waitUntilCompleted (:707-708).  The reference's yellow
 * error pixels (Shaders.metal:99-114: threads of the 8x8 groups outside the tile, pixels outside the
 * target, currentSample > 10000) are written at the same texels; the return value is then 1 instead of 0. */
RT_API int rt_render_tile(rt_ctx *ctx, const struct Uniforms *u, const struct Object *objs, int objCount);

/* draw(in:) (ContentView.swift:575-598) + Present.metal:16-44: nearest-sample the target into an
 * out_w x out_h bgra8Unorm image (row 0 = top), returned to host memory. */
RT_API int rt_present_bgra8(rt_ctx *ctx, uint8_t *out, int out_w, int out_h);

/* ---- batch path: same kernel semantics, whole image per call, linear accumulation ---------------- */

/* Upload and keep the object array on the device (the reference's `objects`, ContentView.swift:239);
 * later calls may pass objs = NULL to use it. */
RT_API int rt_set_scene(rt_ctx *ctx, const struct Object *objs, int objCount);

/* Trace samples rayIndex in [ray_begin, ray_end) of pass u->currentSample for every pixel of the tile
 * rectangle in *u (normally the whole target), each sample exactly as one trip of the loop at
 * Shaders.metal:120-140, and ADD their linear sums into `accum_dev`, a device buffer of
 * width*height float4 (row-major, rgba), or into the context's own buffer when accum_dev is NULL.
 * Enqueued on the context's stream; returns without waiting. */
RT_API int rt_render(rt_ctx *ctx, const struct Uniforms *u, const struct Object *objs, int objCount, int ray_begin,
                     int ray_end, float *accum_dev);
/* No counterpart in the reference.  rt_render for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments and results as
 * rt_render(ctx, u, NULL, 0, ray_begin, ray_end, accum_dev), bit for bit, with two differences: the scene is always
 * the one kept by rt_set_scene (uploading objects allocates), and the launch's scratch -- work counters, camera-ray
 * queues, partial-sum planes -- lies in a static device array of RT_NOALLOC_SCRATCH_BYTES that every GPU the library
 * runs on holds until the process ends, instead of in buffers the context grows on demand.  The call allocates and
 * frees nothing.  Limits: (number of sample chunks per pixel) x width x height x 16 bytes of the render target must
 * not exceed RT_NOALLOC_PLANE_BYTES, else RT_ERR_TOO_LARGE (raise RT_OPT_CHUNK, or trace fewer samples per call);
 * the persistent grid is cut to the 32 768 lanes the queue part holds, so large frames trace slower than with
 * rt_render; RT_OPT_KERNEL 1 (wavefront) is refused.  All contexts on one GPU share that GPU's array: calls that
 * overlap on one GPU (two contexts, two streams) race, and rt_last_stats must be read before the next such call. */
#define RT_NOALLOC_PLANE_BYTES ((size_t)64 << 20)
#define RT_NOALLOC_SCRATCH_BYTES (256 + ((size_t)64 << 20) + RT_NOALLOC_PLANE_BYTES)
RT_API int rt_render_noalloc(rt_ctx *ctx, const struct Uniforms *u, int ray_begin, int ray_end, float *accum_dev);
/* Zero an accumulation buffer (the context's own when NULL).  Enqueued on the stream. */
RT_API int rt_accum_clear(rt_ctx *ctx, float *accum_dev);
/* Device address of the context's own accumulation buffer (width*height*4 floats). */
RT_API int rt_accum_ptr(rt_ctx *ctx, float **accum_dev);
/* accum / spp -> NaN scrub, clamp to [0,1], sqrt (write_color, Shaders.metal:71-85; Lib.metal:144-150)
 * -> target.  This is what computeShader stores for currentSample == 0.  Enqueued on the stream. */
RT_API int rt_resolve(rt_ctx *ctx, const float *accum_dev, int spp);
/* Wait for everything enqueued on the context's stream. */
RT_API int rt_synchronize(rt_ctx *ctx);

/* First-hit buffer of sample (u->currentSample, rayIndex 0): camera ray (Lib.metal:99-106) + closest hit
 * (Shaders.metal:13-32).  out4 = width*height x {t, nx, ny, nz}; ids = object index or -1.  Host buffers.
 * Blocks. */
RT_API int rt_first_hit(rt_ctx *ctx, const struct Uniforms *u, const struct Object *objs, int objCount, float *out4,
                        int32_t *ids);

/* Counters of the last rt_render / rt_render_tile call: segments = closest-hit searches (world_hit calls),
 * tests = ray-object tests the reference's brute-force loop performs for them (segments x objCount),
 * kernel_ms = device time of the trace kernel(s) from HIP events on the context's stream.  Blocks until
 * that call has finished. */
struct rt_stats {
    uint64_t segments;
    uint64_t tests;
    uint64_t samples;
    float kernel_ms;
    int grid_blocks;
    int block_threads;
    int scene_path;  /* how the last trace launch read and filtered the spheres: 0 staged in LDS + vector filter,
                        1 scalar cache + vector filter, 2 streamed + vector filter, 3 staged in LDS + matrix-core filter,
                        4 lifted tiles streamed through LDS + matrix-core filter, 5 lifted tiles read in place by the
                        waves whose rays can touch them */
    /* Work items of the last rt_render launch: the sample indices of a pixel were cut into chunk_count chunks of
     * chunk samples followed by chunks of tail_chunk samples (the last one shorter if need be).  Each chunk is
     * summed sample by sample, and the chunk sums are added in order. */
    int chunk, chunk_count, tail_chunk;
    /* RT_OPT_KERNEL = 2 only, summed over waves, in shader clock cycles: [0] generate rays / draw work,
     * [1] sphere filter pass, [2] exact resolve of candidates, [3] quads + hit record, [4] shading,
     * [5] number of main-loop trips, [6] trips of the resolve loop, [7] candidates resolved, summed over lanes
     * (the last three are counts, not cycles). */
    uint64_t phase_cycles[8];
    /* Executed work of the last launch, counted per wave: wave_trips = closest-hit searches a wave of 64 lanes took
     * part in (lanes without a ray ride along), matrix_blocks = 64-sphere blocks the waves put through the
     * matrix-core filter, each 8 v_mfma_f32_32x32x16_f16 (0 for the vector filters). */
    uint64_t wave_trips, matrix_blocks;
};
RT_API int rt_last_stats(rt_ctx *ctx, struct rt_stats *out);

/* Device durations (ms, HIP events on the context's stream) of the trace launches enqueued since the
 * previous call of this function, oldest first; at most the 64 most recent are kept.  Writes up to `cap`
 * values and stores the number written in *count.  Blocks until those launches have finished. */
RT_API int rt_trace_times(rt_ctx *ctx, float *ms, int cap, int *count);

/* No counterpart in the reference.  Self-test of the sphere filter that precedes the exact hit test: runs
 * blocks x 256 x pairs_per_thread pseudo-random, mostly grazing ray/sphere pairs and reports out[0] pairs,
 * out[1] pairs whose reference discriminant is >= 0, out[2] / out[3] pairs flagged by the general / compact
 * filter, out[4] / out[5] pairs with discriminant >= 0 that the general / compact filter failed to flag.
 * out[4] and out[5] must be 0: the filters may only over-report. */
RT_API int rt_selftest_filter(rt_ctx *ctx, uint32_t seed, int blocks, int pairs_per_thread, uint64_t out[6]);

/* No counterpart in the reference.  The same guarantee for the matrix-core filter: every wave of blocks x 4 draws
 * a scene frame and, `rounds` times, 64 spheres and 64 rays aimed at their silhouettes, and checks all 64 x 64
 * pairs.  slack_log2 = -16 is the slack the library renders with; larger negative values probe the margin.
 * out[0] pairs checked, out[1] pairs whose reference discriminant is >= 0, out[2] pairs the filter flagged,
 * out[3] misses (must be 0 at -16), out[4] pairs outside the filter's range (not filtered when rendering),
 * out[5] out-of-range spheres that were not handed on as candidates (must be 0). */
RT_API int rt_selftest_matrix_filter(rt_ctx *ctx, uint32_t seed, int blocks, int rounds, int slack_log2, uint64_t out[6]);

/* No counterpart in the reference.  The same check for the form the tiled kind (RT_OPT_SCENE_FETCH 3) lifts its rays
 * in: the frame is local to the 64 spheres while the rays start up to 2^20 radii away.  slack_log2 scales both the
 * relative slack and the distance-dependent one; -16 is what the library renders with.  Same outputs. */
RT_API int rt_selftest_tiled_filter(rt_ctx *ctx, uint32_t seed, int blocks, int rounds, int slack_log2, uint64_t out[6]);

/* No counterpart in the reference.  Self-test of the prune that precedes the exact hit test in scenes whose rays cross
 * many spheres (RT_OPT_RESOLVE), on the scene set with rt_set_scene: blocks x 256 threads build rays_per_thread rays
 * each the way paths produce them, with the hard cases made frequent (origins on or a few ulps off a surface,
 * directions tangent to it or at another sphere's silhouette, bounds equal to another sphere's root or a few ulps
 * off it), and check every sphere of the scene against every level of the prune.  out[0] (ray, sphere) pairs,
 * out[1] pairs whose sphere has an accepted root, out[2] pairs the single-sphere test strikes, out[3] pairs the
 * bounding balls of a word or granule strike, out[4] / out[5] pairs struck by the one / the other although the
 * root the reference computes is not above the ray's bound: both must be 0. */
RT_API int rt_selftest_prune(rt_ctx *ctx, uint32_t seed, int blocks, int rays_per_thread, uint64_t out[6]);

/* No counterpart in the reference.  Host-only (no device, no context): the box the library would build at scene upload
 * for the half-width sphere filter (RT_OPT_NARROW) of `count` spheres given as (x, y, z, radius) in slot order.  Returns
 * 1 and fills out when the scene gets one, 0 when it does without (fewer than 128 spheres, more than 8 spheres outside
 * the box or in more than 4 words of 32 slots, or a scene the lifted form does not cover).  out[0..2] the box's centre,
 * out[3..5] its half extents (allowances included), out[6] the distance from the scene's median centre up to which the
 * box test speaks for a ray's origin, out[7] the number of words that hold spheres outside the box, then per such word
 * three 32-bit patterns stored in the floats' bytes: the word's index (slot / 32), the spheres tested for every ray with
 * the vector filter, the spheres that are candidates for every ray (bit 31 - j for slot 32 w + j). */
#define RT_FILTER_BOX_FLOATS 20
RT_API int rt_filter_box(const float *spheres_xyzr, int count, float out[RT_FILTER_BOX_FLOATS]);

/* No counterpart in the reference.  Self-test of the box test that lets rays skip the matrix-core sphere filter
 * (RT_OPT_NARROW), on the scene set with rt_set_scene (RT_ERR_INVALID when the scene has no such box): blocks x 256
 * threads build rays_per_thread rays each -- origins on or a few ulps off a sphere's surface or a face of the box,
 * directions tangent to a sphere, at another sphere's silhouette or along a face of the box -- and check every ray the
 * box test leaves out against the exact hit test (Lib.metal:215-246) of every sphere the box stands for.
 * padding_log2 = 0 tests the box as rendered with; -1 .. -39 scale its allowances by that power of two, -40 removes them.
 * out[0] rays, out[1] rays left out, out[2] (ray, sphere) pairs checked for them, out[3] accepted roots over all pairs
 * of all rays, out[4] accepted roots among the pairs of out[2]: must be 0. */
RT_API int rt_selftest_box(rt_ctx *ctx, uint32_t seed, int blocks, int rays_per_thread, int padding_log2, uint64_t out[6]);

/* No counterpart in the reference.  The same for the boxes of the tiled kind that reads its tiles in place (scenes beyond
 * 1024 spheres, or RT_OPT_SCENE_FETCH 3): a ray that the box of a tile, of its group of 64 tiles, or of one of the tile's
 * blocks of 64 sphere slots leaves out must have no accepted root on any sphere of that tile or block.  Origins on, near
 * or up to 2^12 radii away from a sphere, or on a face of a tile's box.  padding_log2 scales the allowances that grow with
 * the ray's distance (0: as rendered with; -40: none).  out as rt_selftest_box, out[1] counting (ray, tile) pairs left
 * out, out[5] (ray, block) pairs left out. */
RT_API int rt_selftest_tile_boxes(rt_ctx *ctx, uint32_t seed, int blocks, int rays_per_thread, int padding_log2, uint64_t out[6]);

/* No counterpart in the reference.  Host-only (no device, no context): the bounding balls the library builds at scene
 * upload for the kernels that prune candidates a group at a time, for `count` spheres given as (x, y, z, radius) in slot
 * order.  out receives RT_BOUND_FLOATS floats per word of 32 slots ((count + 31) / 32 words): (Cx, Cy, Cz, R2out),
 * (R2in, exempt mask bits, 0, 0) for the word, then the same pair for each of its four granules of 8 slots.  Every
 * sphere of a group that is not exempt lies inside the ball (C, sqrt(R2out / (1 + 2^-6))) and contains the ball
 * (C, sqrt(R2in / (1 - 2^-6))); R2in < 0: no inner ball; mask bit of slot p of the word: 0x80000000 >> p.
 * Returns the number of words, or RT_ERR_INVALID. */
#define RT_BOUND_FLOATS 40
RT_API int rt_group_bounds(const float *spheres_xyzr, int count, float *out);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
