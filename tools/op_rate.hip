// op_rate.hip -- issue cost of single vector instructions on this GPU, in shader clocks per wave64 instruction and SIMD,
// at 4 waves per SIMD with 8 independent dependency chains per wave.  Used to pick the instructions of the sign extraction
// behind the matrix-core filter.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
#define CH 8
#define REP 8

#define KERNEL(name, T, INIT, BODY)                                             \
    __global__ __launch_bounds__(256) void name(unsigned *out, unsigned seed) {  \
        T v[CH];                                                                 \
        T w = INIT(seed * 3u + threadIdx.x);                                     \
        T x = INIT(seed * 5u + threadIdx.x * 7u);                                \
        for (int i = 0; i < CH; ++i) v[i] = INIT(seed + i * 977u + threadIdx.x); \
        for (int it = 0; it < ITERS; ++it) {                                     \
            _Pragma("unroll") for (int r = 0; r < REP; ++r)                      \
                _Pragma("unroll") for (int i = 0; i < CH; ++i) { BODY; }         \
        }                                                                        \
        unsigned s = 0;                                                          \
        for (int i = 0; i < CH; ++i) s ^= __builtin_bit_cast(unsigned, v[i]);    \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                 \
    }
#define UI(x) (x)
#define FL(x) ((float)(x) * 1e-9f + 1.0f)

KERNEL(k_fma, float, FL, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_fma3, float, FL, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w), "v"(x)))
KERNEL(k_fma3b, float, FL, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(w), "v"(x)))
KERNEL(k_fmac3, float, FL, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(w), "v"(x)))
KERNEL(k_fmacs, float, FL, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "s"(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w))), "v"(x)))
KERNEL(k_max2, float, FL, asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[i]) : "v"(w)))
KERNEL(k_mul2, float, FL, asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(w), "v"(x)))
KERNEL(k_align3, unsigned, UI, asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w), "v"(x)))
KERNEL(k_alignbit, unsigned, UI, asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(v[i]) : "v"(w)))
KERNEL(k_lshl_or, unsigned, UI, asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_and_or, unsigned, UI, asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_bfi, unsigned, UI, asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_perm, unsigned, UI, asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_lshr, unsigned, UI, asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(v[i])))
KERNEL(k_or, unsigned, UI, asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_or3, unsigned, UI, asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_lshl_add, unsigned, UI, asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_add3, unsigned, UI, asm volatile("v_add3_u32 %0, %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_add, unsigned, UI, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_bfe, unsigned, UI, asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(v[i])))
KERNEL(k_mad24, unsigned, UI, asm volatile("v_mad_u32_u24 %0, %0, 2, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_mul_lo, unsigned, UI, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_cmp_addc, unsigned, UI, asm volatile("v_cmp_gt_f32 vcc, 0, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(v[i]) : "v"(w) : "vcc"))
KERNEL(k_pkrtz, unsigned, UI, asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_pknorm, unsigned, UI, asm volatile("v_cvt_pknorm_i16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_max3, float, FL, asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_med3, float, FL, asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_mulclamp, float, FL, asm volatile("v_mul_f32 %0, %0, %1 clamp" : "+v"(v[i]) : "v"(w)))
KERNEL(k_accread, unsigned, UI, asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(v[i])::"a0"))
KERNEL(k_fmac, float, FL, asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_fmaak, float, FL, asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7fbe77" : "+v"(v[i]) : "v"(w)))
KERNEL(k_fma_c, float, FL, asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(v[i]) : "v"(w)))
KERNEL(k_mul, float, FL, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_addf, float, FL, asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_maxf, float, FL, asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_sqrt, float, FL, asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_rcp, float, FL, asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_cvt, unsigned, UI, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[i])))
KERNEL(k_cvth, unsigned, UI, asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_cmp32, unsigned, UI, asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(v[i]), "v"(w) : "vcc"))
KERNEL(k_cmp64, unsigned, UI, asm volatile("v_cmp_gt_f32 s[4:5], %0, %1" ::"v"(v[i]), "v"(w) : "s4", "s5"))
KERNEL(k_cnd64, unsigned, UI, asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w), "s"(__builtin_amdgcn_read_exec())))
KERNEL(k_accrd, unsigned, UI, asm volatile("v_accvgpr_read_b32 %0, a1" : "=v"(v[i])::"a1"))
KERNEL(k_xor, unsigned, UI, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_lshl, unsigned, UI, asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(v[i]) : "v"(w)))
// packed FP32: two chains per 64-bit register pair
#define KERNEL2(name, BODY)                                                              \
    __global__ __launch_bounds__(256) void name(unsigned *out, unsigned seed) {           \
        typedef float f2 __attribute__((ext_vector_type(2)));                             \
        f2 v[CH], w = {FL(seed * 3u + threadIdx.x), FL(seed * 7u + threadIdx.x)};         \
        f2 su = {FL(seed * 11u), FL(seed * 13u)};                                         \
        for (int i = 0; i < CH; ++i) v[i] = f2{FL(seed + i * 977u + threadIdx.x), FL(seed + i * 31u)}; \
        for (int it = 0; it < ITERS; ++it) {                                              \
            _Pragma("unroll") for (int r = 0; r < REP; ++r)                               \
                _Pragma("unroll") for (int i = 0; i < CH; ++i) { BODY; }                  \
        }                                                                                 \
        unsigned s = 0;                                                                   \
        for (int i = 0; i < CH; ++i) s ^= __builtin_bit_cast(unsigned, v[i][0]) ^ __builtin_bit_cast(unsigned, v[i][1]); \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                          \
    }
KERNEL2(k_pkfma, asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(w), "v"(v[(i + 1) % CH])))
KERNEL2(k_pkfma_s, asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "s"(su), "v"(w)))
KERNEL2(k_pkfma_sb, asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(v[i]) : "s"(su), "v"(w)))
KERNEL2(k_pkmul, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL2(k_pkadd, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_mov, unsigned, UI, asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_sad, unsigned, UI, asm volatile("v_sad_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w)))
KERNEL(k_xad, unsigned, UI, asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w)))

typedef void (*kern_t)(unsigned *, unsigned);
static double run(kern_t k, unsigned *out, int blocks) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0 * 1e-3;
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, wps = 4;
    unsigned *out;
    (void)hipMalloc(&out, (size_t)cus * wps * 256 * 4);
    struct { const char *name; kern_t k; int ops; } tab[] = {
        {"v_fma_f32 d, d, w, w", k_fma, 1}, {"v_fma_f32 d, d, w, x", k_fma3, 1}, {"v_fma_f32 d, w, x, d", k_fma3b, 1},
        {"v_fmac_f32 d, w, x", k_fmac3, 1}, {"v_fmac_f32 d, s, x", k_fmacs, 1}, {"v_max_f32 d, w, d", k_max2, 1},
        {"v_mul_f32 d, w, x", k_mul2, 1}, {"v_alignbit_b32 d, d, w, x", k_align3, 1}, {"v_alignbit_b32 d, d, w, 31", k_alignbit, 1}, {"v_lshl_or_b32", k_lshl_or, 1}, {"v_and_or_b32", k_and_or, 1},
        {"v_bfi_b32", k_bfi, 1}, {"v_perm_b32", k_perm, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_or_b32", k_or, 1}, {"v_or3_b32", k_or3, 1},
        {"v_lshl_add_u32", k_lshl_add, 1}, {"v_add3_u32", k_add3, 1}, {"v_add_u32", k_add, 1}, {"v_bfe_u32", k_bfe, 1},
        {"v_mad_u32_u24", k_mad24, 1}, {"v_mul_lo_u32", k_mul_lo, 1},
        {"v_cmp_gt_f32 + v_addc_co_u32", k_cmp_addc, 2}, {"v_cvt_pkrtz_f16_f32", k_pkrtz, 1}, {"v_cvt_pknorm_i16_f32", k_pknorm, 1},
        {"v_max3_f32", k_max3, 1}, {"v_med3_f32", k_med3, 1}, {"v_mul_f32 clamp", k_mulclamp, 1},
        {"v_accvgpr_write + read", k_accread, 2}, {"v_mov_b32", k_mov, 1}, {"v_sad_u32", k_sad, 1}, {"v_xad_u32", k_xad, 1},
        {"v_fmac_f32 d, w, w", k_fmac, 1}, {"v_fmaak_f32 (VOP2 + literal)", k_fmaak, 1}, {"v_fma_f32 v, v, v, 1.0", k_fma_c, 1},
        {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_addf, 1}, {"v_max_f32", k_maxf, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_rcp_f32", k_rcp, 1},
        {"v_cvt_f32_u32", k_cvt, 1}, {"v_cvt_f16_f32", k_cvth, 1}, {"v_cmp_gt_f32 -> vcc", k_cmp32, 1},
        {"v_cmp_gt_f32 -> s[4:5]", k_cmp64, 1}, {"v_cndmask_b32 (sgpr pair)", k_cnd64, 1},
        {"v_accvgpr_read_b32", k_accrd, 1}, {"v_pk_fma_f32 (2 fma)", k_pkfma, 1}, {"v_pk_fma_f32 s[pair], v, d", k_pkfma_s, 1},
        {"v_pk_fma_f32 s(broadcast), v, d", k_pkfma_sb, 1}, {"v_pk_mul_f32", k_pkmul, 1}, {"v_pk_add_f32", k_pkadd, 1}, {"v_xor_b32", k_xor, 1}, {"v_lshlrev_b32", k_lshl, 1}};
    printf("%d waves per SIMD, %d chains per wave; clocks at a nominal 2.4 GHz\n", wps, CH);
    for (auto &t : tab) {
        double s = run(t.k, out, cus * wps);
        double n = (double)ITERS * REP * CH * wps;  // sequences per SIMD
        printf("%-32s %6.2f clocks per %s\n", t.name, s * 2.4e9 / n, t.ops == 1 ? "instruction" : "pair");
    }
    return 0;
}
