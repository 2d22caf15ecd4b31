// cvt_probe.hip -- can the sign bits of the matrix-core filter's 32 results per lane be taken with the gfx950 instruction
// that converts 32 binary32 values to FP6 at once (v_cvt_scalef32_2xpk16_fp6_f32), instead of 32 v_alignbit_b32?
// (1) what the instruction does with the signs of awkward inputs and where the 32 results lie in its six output
// registers; (2) what it costs to issue, at 4 waves per SIMD, next to v_alignbit_b32.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u6v __attribute__((ext_vector_type(6)));

__global__ void semantics(const float *in, unsigned *out, float scale) {
    f16v a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[i]; b[i] = in[16 + i]; }
    u6v r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    if (threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) out[i] = r[i];
}

#define ITERS 2048
__global__ __launch_bounds__(256) void rate_cvt(unsigned *out, float seed) {
    f16v a, b;
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed - i - threadIdx.x; }
    u6v acc = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            u6v t;
            asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, 1.0" : "=v"(t) : "v"(a), "v"(b));
            acc ^= t;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ acc[4] ^ acc[5];
}
__global__ __launch_bounds__(256) void rate_align(unsigned *out, unsigned seed) {
    unsigned v[8], w = seed + threadIdx.x;
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(v[i]) : "v"(w));
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the 6-instruction gather that would follow the conversion: three words -> 16 sign bits (mask, multiply, shift-merge)
__global__ __launch_bounds__(256) void rate_xor(unsigned *out, unsigned seed) {
    unsigned v[8], w = seed + threadIdx.x;
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(w));
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K, typename A>
static double time_kernel(K k, A arg, unsigned *dev, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dev, arg);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dev, arg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float host[32];
    const float tiny = 1e-30f, den = 1e-40f;
    const float vals[32] = {1.0f, -1.0f, 0.0f, -0.0f, tiny, -tiny, den, -den, 1e30f, -1e30f, INFINITY, -INFINITY, NAN, -NAN, 0.3f, -0.3f,
                            2.0f, -2.0f, 3.0f, -3.0f, 1e-10f, -1e-10f, 1e10f, -1e10f, 0.5f, -0.5f, 7.0f, -7.0f, 0.1f, -0.1f, 100.0f, -100.0f};
    memcpy(host, vals, sizeof host);
    float *din;
    unsigned *dout;
    hipMalloc((void **)&din, sizeof host);
    hipMalloc((void **)&dout, 4096 * 256 * 4);
    hipMemcpy(din, host, sizeof host, hipMemcpyHostToDevice);
    const float scales[4] = {1.0f, 0x1p-100f, 0x1p100f, 0x1p-126f};
    for (int s = 0; s < 4; ++s) {
        unsigned r[6];
        hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, din, dout, scales[s]);
        hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
        printf("scale %g: words %08x %08x %08x %08x %08x %08x\n  fp6 of input k (bit 5 = sign), k = 0..31, taken as bits 6k..6k+5 of the 192:", scales[s], r[0], r[1], r[2], r[3], r[4], r[5]);
        for (int k = 0; k < 32; ++k) {
            const int bit = 6 * k;
            unsigned long long two = (unsigned long long)r[bit / 32] | ((bit / 32 + 1 < 6 ? (unsigned long long)r[bit / 32 + 1] : 0ull) << 32);
            printf(" %02x", (unsigned)((two >> (bit % 32)) & 63u));
        }
        printf("\n");
    }
    // issue cost: blocks = 256 CUs x 4 workgroups of 256 threads = 4 waves per SIMD
    const int blocks = 256 * 4;
    const double clock_ghz = 2.4;
    double t = time_kernel(rate_cvt, 1.5f, dout, blocks);
    printf("v_cvt_scalef32_2xpk16_fp6_f32: %.2f clocks per instruction and SIMD (at %.1f GHz nominal)\n", t * 1e-3 * clock_ghz * 1e9 / (4.0 * ITERS * 8), clock_ghz);
    t = time_kernel(rate_align, 7u, dout, blocks);
    printf("v_alignbit_b32:                 %.2f clocks per instruction and SIMD\n", t * 1e-3 * clock_ghz * 1e9 / (4.0 * ITERS * 64));
    t = time_kernel(rate_xor, 7u, dout, blocks);
    printf("v_xor_b32:                      %.2f clocks per instruction and SIMD\n", t * 1e-3 * clock_ghz * 1e9 / (4.0 * ITERS * 64));
    return 0;
}
