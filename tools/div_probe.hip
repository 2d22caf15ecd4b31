// Is the division sequence with the scaling steps left out (valid when neither operand needs scaling) bit-identical
// to the compiler's correctly rounded n / a?  And the square root without the rescue of tiny arguments?  Counts
// mismatches over random operands in the ranges the trace kernel uses them in (rt_device.hpp, RootDiv: the division
// is used there; the square root variant saved too little to keep).  make tools && tools/div_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ inline uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

struct DivBy { float a, r; };
__device__ inline DivBy div_prepare(float a) {
    float r0 = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, r0, 1.0f);
    return {a, __builtin_fmaf(e, r0, r0)};
}
__device__ inline float div_fast(float n, DivBy d) {
    float q = n * d.r;
    float e2 = __builtin_fmaf(-d.a, q, n);
    float q1 = __builtin_fmaf(e2, d.r, q);
    float e3 = __builtin_fmaf(-d.a, q1, n);
    return __builtin_fmaf(e3, d.r, q1);
}
// correctly rounded square root of a normal x >= 2^-100: hardware estimate (1 ulp), then the neighbour whose
// residual changes sign -- the compiler's sequence without its rescaling of tiny arguments
__device__ inline float sqrt_fast(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    float rd = __builtin_fmaf(-dn, s, x), ru = __builtin_fmaf(-up, s, x);
    float t = rd <= 0.0f ? dn : s;
    return ru > 0.0f ? up : t;
}

__global__ void probe(uint32_t seed, int iters, unsigned long long *out) {
    uint32_t st = pcg(seed ^ pcg(blockIdx.x * blockDim.x + threadIdx.x));
    unsigned long long bad_div = 0, bad_sqrt = 0, n_div = 0, n_sqrt = 0;
    for (int it = 0; it < iters; ++it) {
        st = pcg(st); uint32_t ma = st;
        st = pcg(st); uint32_t mn = st;
        st = pcg(st); uint32_t ex = st;
        // a: mantissa random, exponent -40..40 (mostly near 0); n: exponent of a plus -12..+90
        int ea = (ex & 1) ? (int)((ex >> 1) % 81) - 40 : (int)((ex >> 1) % 5) - 2;
        int en = ea + (int)((ex >> 12) % 103) - 12;
        // both signs: the quotients of a vector by a (possibly negative) scalar use the same sequence
        float a = __uint_as_float(((uint32_t)(ea + 127) << 23) | (ma >> 9) | ((ex & 0x400u) << 21));
        float n = __uint_as_float(((uint32_t)(en + 127) << 23) | (mn >> 9) | ((ex & 0x800u) << 20));
        DivBy d = div_prepare(a);
        float q0 = n / a, q1 = div_fast(n, d);
        ++n_div;
        if (__float_as_uint(q0) != __float_as_uint(q1)) ++bad_div;
        // sqrt over exponents -100..120
        int es = (int)((ex >> 20) % 221) - 100;
        float x = __uint_as_float(((uint32_t)(es + 127) << 23) | (mn >> 9));
        float s0 = __builtin_sqrtf(x), s1 = sqrt_fast(x);
        ++n_sqrt;
        if (__float_as_uint(s0) != __float_as_uint(s1)) ++bad_sqrt;
    }
    atomicAdd(&out[0], n_div); atomicAdd(&out[1], bad_div); atomicAdd(&out[2], n_sqrt); atomicAdd(&out[3], bad_sqrt);
}

int main() {
    unsigned long long *dev, host[4];
    hipMalloc((void **)&dev, 32);
    hipMemset(dev, 0, 32);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, 12345u, 4096, dev);
    hipMemcpy(host, dev, 32, hipMemcpyDeviceToHost);
    printf("division: %llu pairs, %llu differ from n / a\n", host[0], host[1]);
    printf("sqrt:     %llu values, %llu differ from sqrtf\n", host[2], host[3]);
    return (host[1] || host[3]) ? 1 : 0;
}
