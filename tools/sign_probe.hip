// sign_probe.hip -- cheaper ways to take the sign bits of the matrix-core filter's results (32 binary32 values per lane
// and 32-sphere group) than one v_alignbit_b32 per value.  gfx950 converts two binary32 values to two FP4 values
// (sign + 3 bits) and writes them into one selectable byte of its destination, the other bytes kept
// (v_cvt_scalef32_pk_fp4_f32): four of them fill a register with eight sign bits, at bits 3, 7, ..., 31.
// (1) what the conversion does with the signs of awkward inputs; (2) issue cost of the candidates and of the two
// complete extraction sequences, at 4 waves per SIMD, in nominal 2.4 GHz clocks per instruction and SIMD.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

__global__ void semantics(const float *in, unsigned *out, float scale) {
    unsigned r[4] = {0, 0, 0, 0};
    for (int w = 0; w < 4; ++w) {
        r[w] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r[w], in[8 * w + 0], in[8 * w + 1], scale, 0);
        r[w] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r[w], in[8 * w + 2], in[8 * w + 3], scale, 1);
        r[w] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r[w], in[8 * w + 4], in[8 * w + 5], scale, 2);
        r[w] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r[w], in[8 * w + 6], in[8 * w + 7], scale, 3);
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < 4; ++i) out[i] = r[i];
}

#define ITERS 2048
#define RATE_KERNEL(NAME, DECL, BODY)                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {            \
        unsigned v[8], w = seed + threadIdx.x, x = seed * 3 + threadIdx.x;                  \
        for (int i = 0; i < 8; ++i) v[i] = seed * (i + 3) + threadIdx.x;                    \
        DECL;                                                                               \
        for (int it = 0; it < ITERS; ++it) {                                                \
            _Pragma("unroll") for (int r = 0; r < 8; ++r)                                   \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) { BODY; }                         \
        }                                                                                   \
        unsigned s = 0;                                                                     \
        for (int i = 0; i < 8; ++i) s ^= v[i];                                              \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                            \
    }

RATE_KERNEL(rate_align, , asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(v[i]) : "v"(w)))
RATE_KERNEL(rate_xor, , asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
RATE_KERNEL(rate_fp4_const, , asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, 1.0 op_sel:[0,0,1,0]" : "+v"(v[i]) : "v"(w), "v"(x)))
RATE_KERNEL(rate_fp4_vscale, float sc = 1.0f; asm volatile("" : "+v"(sc)), asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(v[i]) : "v"(w), "v"(x), "v"(sc)))
RATE_KERNEL(rate_fp8_scale, , asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, 1.0 op_sel:[0,0,0,1]" : "+v"(v[i]) : "v"(w), "v"(x)))
RATE_KERNEL(rate_fp8, , asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(v[i]) : "v"(w), "v"(x)))
RATE_KERNEL(rate_pkrtz, , asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(v[i]) : "v"(w), "v"(x)))
RATE_KERNEL(rate_bf16, , asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[i]) : "v"(w), "v"(x)))
RATE_KERNEL(rate_perm, , asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(w), "v"(x)))
RATE_KERNEL(rate_bfi, , asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(w), "v"(x)))
RATE_KERNEL(rate_bfi_lit, unsigned mk = 0x88888888u; asm volatile("" : "+s"(mk)), asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(v[i]) : "v"(w), "s"(mk)))
RATE_KERNEL(rate_shr, , asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(v[i])))
RATE_KERNEL(rate_and_lit, , asm volatile("v_and_b32 %0, 0x88888888, %0" : "+v"(v[i])))
RATE_KERNEL(rate_lshl_or, , asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(v[i]) : "v"(w)))
RATE_KERNEL(rate_and_or, , asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(w), "v"(x)))

// The two complete sequences for 32 values per lane (values in 32 registers, as the matrix instruction leaves them).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void seq_align(unsigned *out, float seed) {
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed - i - threadIdx.x; }
    unsigned acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("" : "+v"(a), "+v"(b));
        unsigned w0 = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) w0 = __builtin_amdgcn_alignbit(w0, __float_as_uint(a[k]), 31);
#pragma unroll
        for (int k = 0; k < 16; ++k) w0 = __builtin_amdgcn_alignbit(w0, __float_as_uint(b[k]), 31);
        acc ^= w0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void seq_fp4(unsigned *out, float seed) {
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed - i - threadIdx.x; }
    unsigned acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("" : "+v"(a), "+v"(b));
        unsigned r[4] = {0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x16 &q = w < 2 ? a : b;
            const int o = (w & 1) * 8;
            r[w] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r[w], q[o + 0], q[o + 1], 1.0f, 0);
            r[w] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r[w], q[o + 2], q[o + 3], 1.0f, 1);
            r[w] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r[w], q[o + 4], q[o + 5], 1.0f, 2);
            r[w] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r[w], q[o + 6], q[o + 7], 1.0f, 3);
        }
        // bits 3 mod 4 from r0, 2 mod 4 from r1 >> 1, 1 mod 4 from r2 >> 2, 0 mod 4 from r3 >> 3
        unsigned t;
        asm volatile("v_lshrrev_b32 %0, 3, %1\n\tv_lshrrev_b32 %2, 2, %2\n\tv_lshrrev_b32 %3, 1, %3\n\t"
                     "v_bfi_b32 %0, %5, %2, %0\n\tv_bfi_b32 %0, %6, %3, %0\n\tv_bfi_b32 %0, %7, %4, %0"
                     : "=&v"(t), "+v"(r[3]), "+v"(r[2]), "+v"(r[1]) : "v"(r[0]), "s"(0x22222222u), "s"(0x44444444u), "s"(0x88888888u));
        acc ^= t;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
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
    const float tiny = 1e-30f, den = 1e-40f;
    const float vals[32] = {1.0f, -1.0f, 0.0f, -0.0f, tiny, -tiny, den, -den, 1e30f, -1e30f, INFINITY, -INFINITY, NAN, -NAN, 0.3f, -0.3f,
                            2.0f, -2.0f, 3.0f, -3.0f, 1e-10f, -1e-10f, 1e10f, -1e10f, 0.5f, -0.5f, 7.0f, -7.0f, 0.1f, -0.1f, 100.0f, -100.0f};
    float *din;
    unsigned *dout;
    hipMalloc((void **)&din, sizeof vals);
    hipMalloc((void **)&dout, 4096 * 256 * 4);
    hipMemcpy(din, vals, sizeof vals, hipMemcpyHostToDevice);
    const float scales[4] = {1.0f, 0x1p-100f, 0x1p100f, 0x1p-126f};
    for (int s = 0; s < 4; ++s) {
        unsigned r[4];
        hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, din, dout, scales[s]);
        hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
        printf("scale %g: words %08x %08x %08x %08x\n  fp4 of input k (bit 3 = sign), k = 0..31:", scales[s], r[0], r[1], r[2], r[3]);
        int wrong = 0;
        for (int k = 0; k < 32; ++k) {
            const unsigned n = (r[k / 8] >> (4 * (k % 8))) & 15u;
            printf(" %x", n);
            // the one property the filter needs: a value that is not negative (sign bit clear, NaN aside) never gets the sign
            if (!std::isnan(vals[k]) && !std::signbit(vals[k]) && (n & 8u)) ++wrong;
        }
        printf("\n  non-negative inputs that came out with a sign: %d\n", wrong);
    }
    const int blocks = 256 * 4;  // 4 workgroups of 256 threads per CU = 4 waves per SIMD
    const double ghz = 2.4;
#define RATE(NAME, LABEL) { double t = time_kernel(NAME, 7u, dout, blocks); printf("%-44s %6.2f clocks per instruction and SIMD\n", LABEL, t * 1e-3 * ghz * 1e9 / (4.0 * ITERS * 64)); }
    RATE(rate_align, "v_alignbit_b32");
    RATE(rate_xor, "v_xor_b32");
    RATE(rate_fp4_const, "v_cvt_scalef32_pk_fp4_f32 (scale 1.0)");
    RATE(rate_fp4_vscale, "v_cvt_scalef32_pk_fp4_f32 (scale in a VGPR)");
    RATE(rate_fp8_scale, "v_cvt_scalef32_pk_fp8_f32");
    RATE(rate_fp8, "v_cvt_pk_fp8_f32");
    RATE(rate_pkrtz, "v_cvt_pkrtz_f16_f32");
    RATE(rate_bf16, "v_cvt_pk_bf16_f32");
    RATE(rate_perm, "v_perm_b32");
    RATE(rate_bfi, "v_bfi_b32");
    RATE(rate_bfi_lit, "v_bfi_b32 with the mask in an SGPR");
    RATE(rate_shr, "v_lshrrev_b32 by a constant");
    RATE(rate_and_lit, "v_and_b32 with a literal");
    RATE(rate_lshl_or, "v_lshl_or_b32");
    RATE(rate_and_or, "v_and_or_b32");
    double t = time_kernel(seq_align, 1.5f, dout, blocks);
    printf("32 values -> one word, 32 v_alignbit_b32:                      %7.1f clocks per 32 values and SIMD\n", t * 1e-3 * ghz * 1e9 / (4.0 * ITERS));
    t = time_kernel(seq_fp4, 1.5f, dout, blocks);
    printf("32 values -> one word, 16 fp4 conversions + 3 shifts + 3 bfi: %7.1f clocks per 32 values and SIMD\n", t * 1e-3 * ghz * 1e9 / (4.0 * ITERS));
    return 0;
}
