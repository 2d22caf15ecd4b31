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

// Number of spheres whose discriminant signs are gathered into one 32-bit mask before the candidates
// among them are resolved.
constexpr int kChunk = 32;

typedef const f4v __attribute__((address_space(4))) *const_f4_ptr;

template <bool FAST, bool LDS_SCENE>
struct SphereFetch;

template <bool FAST>
struct SphereFetch<FAST, true> {
    const f4v *lds;
    RT_DEV f4v get(int i) const { return lds[i]; }
};
template <bool FAST>
struct SphereFetch<FAST, false> {
    const_f4_ptr g;
    RT_DEV f4v get(int i) const { return g[i]; }
};

// Closest sphere hit of ray (o, d) over all spheres: the sphere part of world_hit (Shaders.metal:13-32)
// with hitSphere (Lib.metal:215-246).  Returns the sphere slot or -1; t in tbest.
template <bool FAST, typename Fetch>
RT_DEV int closest_sphere(const Fetch &fetch, int nsph_pad, V3 o, V3 d, float &tbest) {
    const float a = dot(d, d);
    const bool a_ok = !(__builtin_fabsf(a) < 1e-6f);  // Lib.metal:226-228
    int ibest = -1;
    for (int base = 0; base < nsph_pad; base += kChunk) {
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            float b;
            float disc = sphere_disc<FAST>(fetch.get(base + j), o, d, a, b);
            // shift the sign bit of disc into the mask: bit (kChunk-1-j) ends up = (disc is negative)
            m = __builtin_amdgcn_alignbit(m, __float_as_uint(disc), 31);
        }
        // b*b - a*c is never -0.0 and a NaN discriminant never hits, so "sign bit clear" is a superset of
        // the reference's !(disc < 0) that the exact test below filters again.
        uint32_t cand = a_ok ? ~m : 0u;
        while (cand) {
            int j = __builtin_clz(cand);
            cand &= ~(0x80000000u >> j);
            float b;
            float disc = sphere_disc<FAST>(fetch.get(base + j), o, d, a, b);
            float root;
            if (sphere_root(disc, b, a, tbest, root)) {
                tbest = root;
                ibest = base + j;
            }
        }
    }
    return ibest;
}

// Closest hit over spheres and quads; fills hit and returns the object index or -1.
template <bool FAST, typename Fetch>
RT_DEV int world_hit(const TraceParams &P, const Fetch &fetch, V3 o, V3 d, Hit &hit, float &t_out) {
    float tbest = kMaxFloat;
    int slot = closest_sphere<FAST>(fetch, P.nsph_pad, o, d, tbest);
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
            if (t < tbest || (t == tbest && (obj < 0 || quad.obj < obj))) {
                tbest = t;
                obj = quad.obj;
                hit.point = point;
                apply_normals(hit, d, normal);
            }
        }
    }
    t_out = tbest;
    return obj;
}

template <bool FAST, bool LDS_SCENE>
__global__ __launch_bounds__(256) void trace_kernel(const TraceParams P) {
    extern __shared__ f4v lds_sph[];
    SphereFetch<FAST, LDS_SCENE> fetch;
    if constexpr (LDS_SCENE) {
        for (int i = threadIdx.x; i < P.nsph_pad; i += blockDim.x) lds_sph[i] = ((const f4v *)P.sph)[i];
        __syncthreads();
        fetch.lds = lds_sph;
    } else {
        fetch.g = (const_f4_ptr)(uintptr_t)P.sph;
    }

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

    for (;;) {
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
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
        if (!alive) continue;

        // ---- one segment: closest hit (Shaders.metal:38-47) ------------------------------------------
        ++nseg;
        Hit hit;
        float t;
        int obj = world_hit<FAST>(P, fetch, o, d, hit, t);
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

    // segment counter: one atomic per wave
    uint64_t total = nseg;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(P.seg_counter, (unsigned long long)total);
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
template <bool FAST>
__global__ __launch_bounds__(256) void first_hit_kernel(const TraceParams P, float4 *__restrict__ out4,
                                                        int *__restrict__ ids) {
    SphereFetch<FAST, false> fetch;
    fetch.g = (const_f4_ptr)(uintptr_t)P.sph;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.width * P.height) return;
    uint32_t py = (uint32_t)i / (uint32_t)P.width, px = (uint32_t)i - py * (uint32_t)P.width;
    Seed seed = sample_seed(px, py, P.pass, 0);
    V3 o, d;
    get_ray((int)px, (int)py, P.cam, seed, d, o);
    Hit hit;
    float t;
    int obj = world_hit<FAST>(P, fetch, o, d, hit, t);
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

template <bool FAST, bool LDS>
static hipError_t launch_trace_t(const TraceParams &P, int blocks, hipStream_t stream) {
    size_t lds = LDS ? (size_t)P.nsph_pad * sizeof(float4) : 0;
    hipLaunchKernelGGL((trace_kernel<FAST, LDS>), dim3(blocks), dim3(256), lds, stream, P);
    return hipGetLastError();
}

hipError_t launch_trace(const TraceParams &P, bool fast, bool lds_scene, int blocks, hipStream_t stream) {
    if (fast) return lds_scene ? launch_trace_t<true, true>(P, blocks, stream) : launch_trace_t<true, false>(P, blocks, stream);
    return lds_scene ? launch_trace_t<false, true>(P, blocks, stream) : launch_trace_t<false, false>(P, blocks, stream);
}

int trace_blocks_per_cu(bool fast, bool lds_scene, size_t lds_bytes) {
    int n = 0;
    hipError_t e;
    if (fast)
        e = lds_scene ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<true, true>, 256, lds_bytes)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<true, false>, 256, 0);
    else
        e = lds_scene ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<false, true>, 256, lds_bytes)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<false, false>, 256, 0);
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

hipError_t launch_first_hit(const TraceParams &P, bool fast, float4 *out4, int *ids, hipStream_t stream) {
    int n = P.width * P.height;
    if (fast)
        hipLaunchKernelGGL(first_hit_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, stream, P, out4, ids);
    else
        hipLaunchKernelGGL(first_hit_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, stream, P, out4, ids);
    return hipGetLastError();
}

}  // namespace rt
