// rt_kernels.hip -- gfx950 kernels of the path tracer and their launchers.
//
// Replaces the reference's Metal compute kernel (reference: "Ray Tracer/Shaders.metal":91-161 with the
// device library "Ray Tracer/Lib.metal") and its present pass ("Ray Tracer/Present.metal":16-44).
//
// trace_kernel is a persistent wave64 kernel: every lane owns one work item at a time -- (pixel, range of
// sample indices) -- drawn from a global counter, traces the item's samples one ray segment per trip of
// the main loop, and draws the next item when done, so lanes never wait for a neighbour's longer path.
// Each trip tests the lane's ray against every sphere (the reference's brute-force world_hit loop),
// reading the compact sphere records either from LDS (staged once per workgroup) or through the scalar
// cache; only the sign of the discriminant is kept per sphere (one bit), and the few spheres with a
// non-negative discriminant are then resolved exactly (sqrt, divides, interval test) in index order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.hpp"
#include "rt_kernels.h"

namespace rt {

// Spheres per 32-bit candidate mask, and masks per resolve round (kGroup * kChunk spheres are filtered,
// then the candidates among them are resolved, then the next kGroup * kChunk spheres, ...).
constexpr int kChunk = 32;
constexpr int kGroup = 16;
// LDS words each wave needs for its candidate masks: kGroup masks x 64 lanes.
constexpr int kMaskWordsPerWave = kGroup * 64;

typedef const f4v __attribute__((address_space(4))) *const_f4_ptr;

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

template <bool LDS_SCENE>
struct SphereFetch;
template <>
struct SphereFetch<true> {   // filter records staged in LDS by the workgroup
    const f4v *lds;
    RT_DEV f4v get(int i) const { return lds[i]; }
};
template <>
struct SphereFetch<false> {  // filter records read through the scalar cache (uniform index -> s_load)
    const_f4_ptr g;
    RT_DEV f4v get(int i) const { return g[i]; }
};

// Closest sphere hit of ray (o, d) over all spheres: the sphere part of world_hit (Shaders.metal:13-32)
// with hitSphere (Lib.metal:215-246).  Returns the sphere slot or -1; t in tbest.
//
// Pass A (every sphere, every lane): a conservative sign test.  With dn = d * (1 + 2^-18) / |d| and the
// filter record (c, r2q), r2q = r*r * (1 + 2^-17), it evaluates q = (oc.dn)^2 - (|oc|^2 - r2q) with fused
// multiply-adds: 11 vector instructions per sphere.  q >= 0 holds whenever the reference's uncontracted
// discriminant b*b - a*c is >= 0: the two inflations exceed, by a factor of two, the worst-case rounding
// difference between the two evaluations (about 33 ulp of |oc|^2 on the b side, 56 ulp of r*r on the c
// side) for operands within the normal float range.  Only the sign bit of q is kept.
// Pass B (the few candidates, in index order): the reference's arithmetic exactly -- uncontracted
// discriminant, sqrt, divisions, strict interval test -- so hit decisions and distances are bit-identical
// to the reference's; a false positive of pass A costs time, never correctness.
template <typename Fetch, bool PROF>
RT_DEV int closest_sphere(const TraceParams &P, const Fetch &fetch, uint32_t *wave_masks, V3 o, V3 d, float &tbest,
                          Prof<PROF> &prof) {
    const float a = dot(d, d);
    const bool a_ok = !(__builtin_fabsf(a) < 1e-6f);  // Lib.metal:226-228: such a ray hits no sphere
    const float s = __builtin_amdgcn_rsqf(a) * (1.0f + 0x1p-18f);
    const V3 dn = v3(d.x * s, d.y * s, d.z * s);
    const int lane = threadIdx.x & 63;
    int ibest = -1;
    for (int gbase = 0; gbase < P.nsph_pad; gbase += kGroup * kChunk) {
        const int nchunk = min(kGroup, (P.nsph_pad - gbase) / kChunk);
        uint32_t nz = 0;  // bit ch set: this lane has a candidate in chunk ch
        for (int ch = 0; ch < nchunk; ++ch) {
            const int base = gbase + ch * kChunk;
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < kChunk; ++j) {
                f4v r = fetch.get(base + j);
                float ocx = o.x - r.x, ocy = o.y - r.y, ocz = o.z - r.z;
                float bn = __builtin_fmaf(ocz, dn.z, __builtin_fmaf(ocy, dn.y, ocx * dn.x));
                float cq = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, -r.w)));
                float q = __builtin_fmaf(bn, bn, -cq);
                // shift q's sign bit into the mask: sphere base+j ends up at bit (31 - j)
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(q), 31);
            }
            uint32_t cand = ~m;
            wave_masks[ch * 64 + lane] = cand;
            nz |= (cand != 0u ? 1u : 0u) << ch;
        }
        if (!a_ok) nz = 0;
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
            const int k = gbase + w * kChunk + j;
            // Lib.metal:216-238 as written
            f4v rec = fetch.get(k);
            float radius = P.rad[k];
            float ocx = o.x - rec.x, ocy = o.y - rec.y, ocz = o.z - rec.z;
            float b = ocx * d.x + ocy * d.y + ocz * d.z;
            float c = (ocx * ocx + ocy * ocy + ocz * ocz) - radius * radius;
            float disc = b * b - a * c;
            float root;
            if (sphere_root(disc, b, a, tbest, root)) {
                tbest = root;
                ibest = k;
            }
        }
        prof.lap(2);
    }
    return ibest;
}

// Closest hit over spheres and quads; fills hit and returns the object index or -1.
template <typename Fetch, bool PROF>
RT_DEV int world_hit(const TraceParams &P, const Fetch &fetch, uint32_t *wave_masks, V3 o, V3 d, Hit &hit, float &t_out,
                     Prof<PROF> &prof) {
    float tbest = kMaxFloat;
    int slot = closest_sphere(P, fetch, wave_masks, o, d, tbest, prof);
    int obj = -1;
    if (slot >= 0) {
        obj = P.sph_obj[slot];
        f4v s = fetch.get(slot);
        float r = P.rad[slot];
        hit.point = o + tbest * d;                                         // Lib.metal:241
        V3 outward = (hit.point - v3(s.x, s.y, s.z)) / r;                  // Lib.metal:242
        apply_normals(hit, d, outward);
    }
    for (int q = 0; q < P.nquad; ++q) {
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

// Dynamic LDS layout of trace_kernel and first_hit_kernel: [candidate masks: 4 waves x kMaskWordsPerWave
// words][filter records: nsph_pad x 16 B when staged in LDS].
template <bool LDS_SCENE>
RT_DEV SphereFetch<LDS_SCENE> stage_scene(const TraceParams &P, uint32_t *lds_words, uint32_t *&wave_masks) {
    wave_masks = lds_words + (threadIdx.x >> 6) * kMaskWordsPerWave;
    SphereFetch<LDS_SCENE> fetch;
    if constexpr (LDS_SCENE) {
        f4v *lds_sph = (f4v *)(lds_words + (blockDim.x >> 6) * kMaskWordsPerWave);
        for (int i = threadIdx.x; i < P.nsph_pad; i += blockDim.x) lds_sph[i] = ((const f4v *)P.sphq)[i];
        __syncthreads();
        fetch.lds = lds_sph;
    } else {
        fetch.g = (const_f4_ptr)(uintptr_t)P.sphq;
    }
    return fetch;
}

template <bool LDS_SCENE, bool PROF, int WAVES>
__global__ __launch_bounds__(256, WAVES) void trace_kernel(const TraceParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    uint32_t *wave_masks;
    SphereFetch<LDS_SCENE> fetch = stage_scene<LDS_SCENE>(P, lds_words, wave_masks);

    const uint32_t tile_pixels = (uint32_t)P.tileW * (uint32_t)P.tileH;

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
        while (alive && need_ray) {
            if (!have_item || ray >= ray_end) {
                if (have_item) {
                    size_t p = (size_t)py * P.width + px;
                    if (P.mode == kModeAccum) {
                        P.planes[(size_t)chunk * ((size_t)P.width * P.height) + p] = make_float4(acc.x, acc.y, acc.z, acc.w);
                    } else {
                        // Shaders.metal:142-160
                        float n = (float)P.spp_pass;
                        V4 passAcc = v4(acc.x / n, acc.y / n, acc.z / n, acc.w / n);
                        float4 e = P.target[p];
                        V4 finalColor;
                        if (P.pass == 0) {
                            finalColor = passAcc;
                        } else {
                            float ex = e.x * e.x, ey = e.y * e.y, ez = e.z * e.z;
                            float totalRaysSoFar = (float)(P.pass * P.spp_pass + P.spp_pass);
                            float passWeight = n / totalRaysSoFar;
                            finalColor.x = ex * (1.0f - passWeight) + passAcc.x * passWeight;
                            finalColor.y = ey * (1.0f - passWeight) + passAcc.y * passWeight;
                            finalColor.z = ez * (1.0f - passWeight) + passAcc.z * passWeight;
                            finalColor.w = 1.0f;
                        }
                        V4 w = write_color(finalColor);
                        P.target[p] = make_float4(w.x, w.y, w.z, w.w);
                    }
                    have_item = false;
                }
                uint32_t item = atomicAdd(P.counter, 1u);
                if (item >= P.nitems) {
                    alive = false;
                    break;
                }
                chunk = item / tile_pixels;
                uint32_t pix = item - chunk * tile_pixels;
                uint32_t ty = pix / (uint32_t)P.tileW;
                px = (uint32_t)P.tileX + (pix - ty * (uint32_t)P.tileW);
                py = (uint32_t)P.tileY + ty;
                ray = P.ray_begin + (int)chunk * P.chunk;
                ray_end = min(ray + P.chunk, P.ray_end);
                acc = v4(0, 0, 0, 0);
                have_item = true;
                if (ray >= ray_end) continue;  // empty range: store zeros on the next trip
            }
            // Shaders.metal:121-132
            seed = sample_seed(px, py, P.pass, ray);
            ++ray;
            get_ray((int)px, (int)py, P.cam, seed, d, o);
            if (length(d) < 1e-6f || __builtin_isnan(d.x) || __builtin_isnan(d.y) || __builtin_isnan(d.z)) continue;
            thr = v4(1, 1, 1, 1);
            fin = v4(0, 0, 0, 0);
            depth = 0;
            if (P.maxDepth <= 0) continue;  // color_ray's loop body never runs: the sample is (0,0,0,0)
            need_ray = false;
        }
        prof.lap(0);
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
        if (!alive) continue;

        // ---- one segment: closest hit (Shaders.metal:38-47) ------------------------------------------
        ++nseg;
        Hit hit;
        float t;
        int obj = world_hit(P, fetch, wave_masks, o, d, hit, t, prof);
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
        prof.lap(4);
    }

    // segment counter: one atomic per wave
    uint64_t total = nseg;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(P.seg_counter, (unsigned long long)total);
    if constexpr (PROF) {
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < 6; ++i) atomicAdd(P.seg_counter + 1 + i, (unsigned long long)prof.t[i]);
    }
}

// accum[p] += planes[0][p] + planes[1][p] + ... (sequential, in plane order) over the tile rectangle.
__global__ void fold_kernel(float4 *__restrict__ accum, const float4 *__restrict__ planes, int nplanes, int width,
                            int height, int tileX, int tileY, int tileW, int tileH) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tileW * tileH) return;
    int ty = i / tileW;
    size_t p = (size_t)(tileY + ty) * width + (tileX + (i - ty * tileW));
    size_t stride = (size_t)width * height;
    float4 a = accum[p];
    for (int k = 0; k < nplanes; ++k) {
        float4 v = planes[(size_t)k * stride + p];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    accum[p] = a;
}

// target[p] = write_color(accum[p] / spp)   (Shaders.metal:142, 150, 160)
__global__ void resolve_kernel(float4 *__restrict__ target, const float4 *__restrict__ accum, int npix, float spp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float4 a = accum[i];
    V4 w = write_color(v4(a.x / spp, a.y / spp, a.z / spp, a.w / spp));
    target[i] = make_float4(w.x, w.y, w.z, w.w);
}

__global__ void fill_rect_kernel(float4 *__restrict__ target, int width, int height, int x0, int y0, int w, int h,
                                 float4 value) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    int y = y0 + i / w, x = x0 + i % w;
    if (x < width && y < height) target[(size_t)y * width + x] = value;
}

// The reference's yellow error texels for one dispatch (Shaders.metal:99-114, error_write :87-89): every
// thread (gx, gy) of the gw x gh grid that fails a guard writes (1,1,0,1) at texel (gx, gy).
__global__ void error_pixels_kernel(float4 *__restrict__ target, int width, int height, int gw, int gh, int tileX,
                                    int tileY, int tileW, int tileH, int all_fail) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= gw * gh) return;
    int gy = i / gw, gx = i - gy * gw;
    bool fail = all_fail || gx >= tileW || gy >= tileH || (uint32_t)(tileX + gx) >= (uint32_t)width ||
                (uint32_t)(tileY + gy) >= (uint32_t)height;
    if (fail && gx < width && gy < height) target[(size_t)gy * width + gx] = make_float4(1.0f, 1.0f, 0.0f, 1.0f);
}

// Present.metal:39-44 + bgra8Unorm store: nearest texel, clamp to [0,1], x*255 rounded to nearest even.
__global__ void present_bgra8_kernel(const float4 *__restrict__ target, int width, int height, uint32_t *__restrict__ out,
                                     int out_w, int out_h) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_w * out_h) return;
    int y = i / out_w, x = i - y * out_w;
    int ty = (int)(((int64_t)(2 * y + 1) * height) / (2 * (int64_t)out_h));
    int tx = (int)(((int64_t)(2 * x + 1) * width) / (2 * (int64_t)out_w));
    float4 c = target[(size_t)ty * width + tx];
    auto cvt = [](float v) -> uint32_t {
        v = __builtin_isnan(v) ? 0.0f : __builtin_fminf(__builtin_fmaxf(v, 0.0f), 1.0f);
        return (uint32_t)__builtin_rintf(v * 255.0f);
    };
    out[i] = cvt(c.z) | (cvt(c.y) << 8) | (cvt(c.x) << 16) | (cvt(c.w) << 24);
}

// First hit of sample (pass, rayIndex 0) per pixel: {t, nx, ny, nz} and the object index.
__global__ __launch_bounds__(256) void first_hit_kernel(const TraceParams P, float4 *__restrict__ out4,
                                                        int *__restrict__ ids) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    uint32_t *wave_masks;
    SphereFetch<false> fetch = stage_scene<false>(P, lds_words, wave_masks);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.width * P.height) return;
    uint32_t py = (uint32_t)i / (uint32_t)P.width, px = (uint32_t)i - py * (uint32_t)P.width;
    Seed seed = sample_seed(px, py, P.pass, 0);
    V3 o, d;
    get_ray((int)px, (int)py, P.cam, seed, d, o);
    Hit hit;
    float t;
    Prof<false> prof;
    int obj = world_hit(P, fetch, wave_masks, o, d, hit, t, prof);
    if (obj >= 0) {
        out4[i] = make_float4(t, hit.normal.x, hit.normal.y, hit.normal.z);
    } else {
        out4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    ids[i] = obj;
}

}  // namespace rt

// ---- launchers (called from rt_api.hip) --------------------------------------------------------------
namespace rt {

size_t trace_lds_bytes(const TraceParams &P, bool lds_scene) {
    return (size_t)4 * kMaskWordsPerWave * sizeof(uint32_t) + (lds_scene ? (size_t)P.nsph_pad * sizeof(f4v) : 0);
}

template <int WAVES>
static hipError_t launch_trace_w(const TraceParams &P, bool lds_scene, bool profile, int blocks, hipStream_t stream) {
    size_t lds = trace_lds_bytes(P, lds_scene);
    if (profile) {
        if (lds_scene)
            hipLaunchKernelGGL((trace_kernel<true, true, WAVES>), dim3(blocks), dim3(256), lds, stream, P);
        else
            hipLaunchKernelGGL((trace_kernel<false, true, WAVES>), dim3(blocks), dim3(256), lds, stream, P);
    } else if (lds_scene) {
        hipLaunchKernelGGL((trace_kernel<true, false, WAVES>), dim3(blocks), dim3(256), lds, stream, P);
    } else {
        hipLaunchKernelGGL((trace_kernel<false, false, WAVES>), dim3(blocks), dim3(256), lds, stream, P);
    }
    return hipGetLastError();
}

hipError_t launch_trace(const TraceParams &P, bool lds_scene, bool profile, int waves, int blocks, hipStream_t stream) {
    switch (waves) {
        case 5: return launch_trace_w<5>(P, lds_scene, profile, blocks, stream);
        case 6: return launch_trace_w<6>(P, lds_scene, profile, blocks, stream);
        case 8: return launch_trace_w<8>(P, lds_scene, profile, blocks, stream);
        default: return launch_trace_w<4>(P, lds_scene, profile, blocks, stream);
    }
}

int trace_blocks_per_cu(const TraceParams &P, bool lds_scene, int waves) {
    int n = 0;
    size_t lds = trace_lds_bytes(P, lds_scene);
    hipError_t e;
    switch (waves) {
        case 5: e = lds_scene ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<true, false, 5>, 256, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<false, false, 5>, 256, lds); break;
        case 6: e = lds_scene ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<true, false, 6>, 256, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<false, false, 6>, 256, lds); break;
        case 8: e = lds_scene ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<true, false, 8>, 256, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<false, false, 8>, 256, lds); break;
        default: e = lds_scene ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<true, false, 4>, 256, lds)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<false, false, 4>, 256, lds); break;
    }
    if (e != hipSuccess || n < 1) n = 1;
    return n;
}

hipError_t launch_fold(float4 *accum, const float4 *planes, int nplanes, int width, int height, int tileX, int tileY,
                       int tileW, int tileH, hipStream_t stream) {
    int n = tileW * tileH;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fold_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, planes, nplanes, width, height,
                       tileX, tileY, tileW, tileH);
    return hipGetLastError();
}

hipError_t launch_resolve(float4 *target, const float4 *accum, int npix, int spp, hipStream_t stream) {
    hipLaunchKernelGGL(resolve_kernel, dim3((npix + 255) / 256), dim3(256), 0, stream, target, accum, npix, (float)spp);
    return hipGetLastError();
}

hipError_t launch_fill_rect(float4 *target, int width, int height, int x0, int y0, int w, int h, float4 value,
                            hipStream_t stream) {
    int n = w * h;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_rect_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, target, width, height, x0, y0, w, h,
                       value);
    return hipGetLastError();
}

hipError_t launch_error_pixels(float4 *target, int width, int height, int gw, int gh, int tileX, int tileY, int tileW,
                               int tileH, int all_fail, hipStream_t stream) {
    int n = gw * gh;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(error_pixels_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, target, width, height, gw, gh,
                       tileX, tileY, tileW, tileH, all_fail);
    return hipGetLastError();
}

hipError_t launch_present_bgra8(const float4 *target, int width, int height, uint32_t *out, int out_w, int out_h,
                                hipStream_t stream) {
    int n = out_w * out_h;
    hipLaunchKernelGGL(present_bgra8_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, target, width, height, out, out_w,
                       out_h);
    return hipGetLastError();
}

hipError_t launch_first_hit(const TraceParams &P, float4 *out4, int *ids, hipStream_t stream) {
    int n = P.width * P.height;
    hipLaunchKernelGGL(first_hit_kernel, dim3((n + 255) / 256), dim3(256), trace_lds_bytes(P, false), stream, P, out4, ids);
    return hipGetLastError();
}

}  // namespace rt
