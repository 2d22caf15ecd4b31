// mfma_hazard_probe.hip -- how far behind v_mfma_f32_32x32x16_f16 must a vector instruction be before it may read
// the result, and is there a hardware interlock if it is not?  (There is none on MI355X.)
// The compiler keeps such a read 11 "wait states" behind the matrix instruction and counts every instruction in
// between as one.  Each test issues, in one asm block,
//     v_mfma v[64:79], a, b, 0 ; filler ; v_mov_b32 early, v[64 + R] ; long wait ; v_mov_b32 late, v[64 + R]
// and counts early != late.  One wave per SIMD (64 KiB of LDS per 256-thread workgroup and one workgroup per CU), so
// no other wave's instructions stretch the distance.  Fillers: one s_nop N, N + 1 independent v_mul_f32, N + 1
// independent v_alignbit_b32.  Prints, per result register, which filler lengths gave stale reads.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define MFMA "v_mfma_f32_32x32x16_f16 v[64:79], %7, %8, 0\n"
#define TAIL "v_mov_b32 %0, v[%c9]\n s_nop 15\n s_nop 15\n s_nop 15\n v_mov_b32 %1, v[%c9]\n"
#define OPS                                                                                                        \
    : "=&v"(early), "=&v"(late), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(u)                                   \
    : "v"(a), "v"(b), "n"(64 + R), "n"(N)                                                                          \
    : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"
#define M2 "v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %6\n"
#define M4 M2 "v_mul_f32 %4, %4, %6\n v_mul_f32 %5, %5, %6\n"
#define A2 "v_alignbit_b32 %2, %2, %6, 31\n v_alignbit_b32 %3, %3, %6, 31\n"
#define A4 A2 "v_alignbit_b32 %4, %4, %6, 31\n v_alignbit_b32 %5, %5, %6, 31\n"

// KIND 0: s_nop N   1: N + 1 v_mul_f32   2: N + 1 v_alignbit_b32
template <int KIND, int R, int N>
__global__ __launch_bounds__(256) void k(const float *in, int iters, unsigned *bad) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(in[threadIdx.x % 64] + i);
        b[i] = (_Float16)(in[64 + threadIdx.x % 64] * 0.5f + i);
    }
    float t0 = in[threadIdx.x], t1 = t0 + 1, t2 = t0 + 2, t3 = t0 + 3, u = 1.0001f;
    unsigned stale = 0;
    for (int it = 0; it < iters; ++it) {
        float early, late;
        if constexpr (KIND == 0) {
            asm volatile(MFMA "s_nop %c10\n" TAIL OPS);
        } else if constexpr (KIND == 1) {
            if constexpr (N == 3) asm volatile(MFMA M4 TAIL OPS);
            if constexpr (N == 7) asm volatile(MFMA M4 M4 TAIL OPS);
            if constexpr (N == 9) asm volatile(MFMA M4 M4 M2 TAIL OPS);
            if constexpr (N == 11) asm volatile(MFMA M4 M4 M4 TAIL OPS);
            if constexpr (N == 13) asm volatile(MFMA M4 M4 M4 M2 TAIL OPS);
            if constexpr (N == 15) asm volatile(MFMA M4 M4 M4 M4 TAIL OPS);
        } else {
            if constexpr (N == 3) asm volatile(MFMA A4 TAIL OPS);
            if constexpr (N == 7) asm volatile(MFMA A4 A4 TAIL OPS);
            if constexpr (N == 9) asm volatile(MFMA A4 A4 A2 TAIL OPS);
            if constexpr (N == 11) asm volatile(MFMA A4 A4 A4 TAIL OPS);
            if constexpr (N == 13) asm volatile(MFMA A4 A4 A4 A2 TAIL OPS);
            if constexpr (N == 15) asm volatile(MFMA A4 A4 A4 A4 TAIL OPS);
        }
        stale += __float_as_uint(early) != __float_as_uint(late);
        b[it & 7] = (_Float16)((float)((it * 2654435761u + threadIdx.x * 40503u) >> 20) * (1.0f / 4096.0f) - 0.5f);
        if (KIND != 0) {
            t0 = in[it & 63];
            t1 = t0 + 1;
            t2 = t0 + 2;
            t3 = t0 + 3;
        }
    }
    if (t0 + t1 + t2 + t3 == 1234.5f) stale += 1000000;
    atomicAdd(bad, stale);
}

static const float *g_in;
static unsigned *g_bad;

template <int KIND, int R, int N>
unsigned run() {
    const int iters = 4000, blocks = 256;
    (void)hipMemset(g_bad, 0, 4);
    hipLaunchKernelGGL((k<KIND, R, N>), dim3(blocks), dim3(256), 65536, 0, g_in, iters, g_bad);
    unsigned n = 0;
    (void)hipMemcpy(&n, g_bad, 4, hipMemcpyDeviceToHost);
    return n;
}

template <int KIND, int R>
void row(const char *what) {
    unsigned n[16] = {0};
    bool have[16] = {false};
#define CASE(N)              \
    n[N] = run<KIND, R, N>(); \
    have[N] = true;
    if constexpr (KIND == 0) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
    } else {
        CASE(3) CASE(7) CASE(9) CASE(11) CASE(13) CASE(15)
    }
    printf("result register %2d, fillers of %-20s:", R, what);
    for (int i = 0; i < 16; ++i)
        if (have[i]) printf(" %d:%s", i + 1, n[i] ? "STALE" : "ok");
    printf("\n");
}

int main() {
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 0.01f * (float)((i * 37) % 101) - 0.3f;
    float *in;
    (void)hipMalloc(&in, sizeof h);
    (void)hipMalloc(&g_bad, 4);
    (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    g_in = in;
    row<0, 0>("s_nop wait states");
    row<0, 3>("s_nop wait states");
    row<0, 4>("s_nop wait states");
    row<0, 7>("s_nop wait states");
    row<0, 8>("s_nop wait states");
    row<0, 11>("s_nop wait states");
    row<0, 12>("s_nop wait states");
    row<0, 15>("s_nop wait states");
    row<1, 0>("v_mul_f32");
    row<1, 7>("v_mul_f32");
    row<1, 15>("v_mul_f32");
    row<2, 0>("v_alignbit_b32");
    row<2, 7>("v_alignbit_b32");
    row<2, 15>("v_alignbit_b32");
    return 0;
}
