// rt_kernels.hip -- the small HBM-bound kernels around the trace kernel: fold of per-chunk partial sums,
// resolve (mean -> NaN scrub -> clamp -> sqrt; reference: "Ray Tracer/Shaders.metal":71-85, 142-160),
// target fills, the reference's yellow error texels (Shaders.metal:87-114) and the present pass
// (reference: "Ray Tracer/Present.metal":16-44 plus the bgra8Unorm drawable).  The trace kernel itself is
// in rt_trace.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.hpp"
#include "rt_kernels.h"

namespace rt {

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

// One pass of the reference's kernel over a tile, from per-sample planes of the tile (plane k = sample k, pixel (x, y) of
// the tile at y * tileW + x): passAccumulator += sample in sample order (Shaders.metal:120-140; a skipped sample's plane
// holds zeros), /= sampleCount (:142), blend with the gamma-encoded target (:144-158), write_color (:160).
__global__ void blend_tile_kernel(float4 *__restrict__ target, const float4 *__restrict__ planes, int nplanes, int width,
                                  int tileX, int tileY, int tileW, int tileH, int pass, int spp_pass) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tileW * tileH) return;
    int ty = i / tileW;
    size_t p = (size_t)(tileY + ty) * width + (tileX + (i - ty * tileW));
    size_t stride = (size_t)tileW * tileH;
    V4 acc = v4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int k = 0; k < nplanes; ++k) {
        float4 v = planes[(size_t)k * stride + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float n = (float)spp_pass;
    V4 passAcc = v4(acc.x / n, acc.y / n, acc.z / n, acc.w / n);
    V4 finalColor;
    if (pass == 0) {
        finalColor = passAcc;
    } else {
        float4 e = target[p];
        float ex = e.x * e.x, ey = e.y * e.y, ez = e.z * e.z;
        float totalRaysSoFar = (float)(pass * spp_pass + spp_pass);
        float passWeight = n / totalRaysSoFar;
        finalColor.x = ex * (1.0f - passWeight) + passAcc.x * passWeight;
        finalColor.y = ey * (1.0f - passWeight) + passAcc.y * passWeight;
        finalColor.z = ez * (1.0f - passWeight) + passAcc.z * passWeight;
        finalColor.w = 1.0f;
    }
    V4 w = write_color(finalColor);
    target[p] = make_float4(w.x, w.y, w.z, w.w);
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

}  // namespace rt

namespace rt {

hipError_t launch_fold(float4 *accum, const float4 *planes, int nplanes, int width, int height, int tileX, int tileY,
                       int tileW, int tileH, hipStream_t stream) {
    int n = tileW * tileH;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fold_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, planes, nplanes, width, height,
                       tileX, tileY, tileW, tileH);
    return hipGetLastError();
}

hipError_t launch_blend_tile(float4 *target, const float4 *planes, int nplanes, int width, int tileX, int tileY, int tileW,
                             int tileH, int pass, int spp_pass, hipStream_t stream) {
    int n = tileW * tileH;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(blend_tile_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, target, planes, nplanes, width, tileX,
                       tileY, tileW, tileH, pass, spp_pass);
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

}  // namespace rt
