// filter_probe.hip -- how much of the matrix-core sphere filter's two halves (the matrix instructions and the sign
// extraction that follows them) run concurrently on a SIMD, in isolation from the rest of the trace kernel.
// One 1024-thread workgroup per CU (4 waves per SIMD, as the trace kernel runs); every wave loops over the 8 blocks
// of a 512-sphere scene staged in LDS.  Variants:
//   0  filter_round_matrix exactly as the trace kernel calls it
//   1  its matrix instructions only (results kept alive, not read)
//   2  its sign extraction only (on registers that nothing writes)
//   3  waves 0-7 run variant 1, waves 8-15 variant 2 (two of each per SIMD): the co-execution the hardware allows
//   4  variant 0 with waves 8-15 started half a group late
//   5  as 3, the matrix waves issuing only instructions with a zero C operand (no accumulator read)
//   6  as 3, the matrix waves issuing only accumulating instructions
//   8  the filter with every matrix instruction followed by 8 sign extractions of the other accumulator (pipelined)
//   7  as 3, the matrix waves issuing v_mfma_f32_16x16x32_f16 with a zero C operand, 8 per 32-sphere group (same flops)
// Reported: shader clocks (s_memtime) and wall time per block and wave, and the shader clock rate under that load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../ray-tracing_amd/csrc/rt_hit.hpp"

using namespace rt;

#define BLOCKS 8

__device__ __forceinline__ uint64_t realtime() {
    uint64_t t;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

// Pass A with the matrix pipe and the vector ALU busy at the same time inside one wave: the two accumulators of a
// 32-sphere group (rays 0-31 and rays 32-63) are produced half a group apart, and while the two matrix instructions
// of one run, the 16 results of the other are turned into sign bits, 8 behind each matrix instruction.  The order
// is fixed with volatile asm statements (the compiler does not reorder them against each other).  The hardware has
// no interlock on matrix results: register r of a result may be read no sooner than 5 + r / 2 (rounded up)
// instructions after its matrix instruction; here the first read comes 9 instructions after it and each further
// register one instruction later.
RT_DEV void mfma_zero(f32x16 &q, const h8 a, const h8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(q) : "v"(a), "v"(b));
}
RT_DEV void mfma_acc(f32x16 &q, const h8 a, const h8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(q) : "v"(a), "v"(b));
}
template <int K0>
RT_DEV void signs8(uint32_t &w, const f32x16 &q) {
    asm volatile(
        "v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %2, 31\n v_alignbit_b32 %0, %0, %3, 31\n"
        "v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %0, %0, %5, 31\n v_alignbit_b32 %0, %0, %6, 31\n"
        "v_alignbit_b32 %0, %0, %7, 31\n v_alignbit_b32 %0, %0, %8, 31"
        : "+v"(w)
        : "v"(q[K0]), "v"(q[K0 + 1]), "v"(q[K0 + 2]), "v"(q[K0 + 3]), "v"(q[K0 + 4]), "v"(q[K0 + 5]), "v"(q[K0 + 6]), "v"(q[K0 + 7]));
}
RT_DEV uint32_t filter_round_matrix_pipelined(const h8 *pm, int blk0, int blk1, uint32_t *masks, int lane, const RayLifted &L) {
    uint32_t nz = 0;
    const h8 *A = pm + (size_t)blk0 * 256 + lane;
    f32x16 q0, q1;
    uint32_t w0 = 0, w1 = 0;
    const int groups = 2 * (blk1 - blk0);
    h8 a_lo = A[0], a_hi = A[64];
    // group 0: nothing to extract beside its first two matrix instructions
    mfma_zero(q0, a_lo, L.b00);
    mfma_acc(q0, a_hi, L.b01);
    asm volatile("s_nop 7");  // in the loop the first read of q0 is 9 or more instructions behind this; here, make it so
    for (int g = 0;;) {
        A += 128;
        mfma_zero(q1, a_lo, L.b10);
        a_lo = A[0];
        signs8<0>(w0, q0);
        mfma_acc(q1, a_hi, L.b11);
        a_hi = A[64];
        signs8<8>(w0, q0);
        if (++g == groups) break;
        mfma_zero(q0, a_lo, L.b00);
        signs8<0>(w1, q1);
        mfma_acc(q0, a_hi, L.b01);
        signs8<8>(w1, q1);
        if ((g & 1) == 0) {  // block (g - 2) / 2 is complete
            uint32_t c0 = w0, c1 = w1;
            swap_halves(c0, c1);
            c0 = ~c0;
            c1 = ~c1;
            const int w = g - 2;
            masks[w * 64 + lane] = c0;
            masks[(w + 1) * 64 + lane] = c1;
            nz |= ((c0 != 0u ? 1u : 0u) | (c1 != 0u ? 2u : 0u)) << w;
        }
    }
    // q1's last matrix instruction is 8 instructions back, and each further register is read one instruction later
    signs8<0>(w1, q1);
    signs8<8>(w1, q1);
    {
        uint32_t c0 = w0, c1 = w1;
        swap_halves(c0, c1);
        c0 = ~c0;
        c1 = ~c1;
        const int w = groups - 2;
        masks[w * 64 + lane] = c0;
        masks[(w + 1) * 64 + lane] = c1;
        nz |= ((c0 != 0u ? 1u : 0u) | (c1 != 0u ? 2u : 0u)) << w;
    }
    return nz;
}

template <int V>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k(const h8 *A_global, int trips, unsigned long long *out, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    h8 *A = (h8 *)lds;                                   // BLOCKS * 256 h8 = 32 KiB
    uint32_t *masks_all = lds + BLOCKS * 256 * 4;        // 16 waves * 16 words * 64 lanes
    for (int i = threadIdx.x; i < BLOCKS * 256; i += blockDim.x) A[i] = A_global[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *masks = masks_all + wave * (2 * BLOCKS * 64);
    RayLifted L;
    for (int i = 0; i < 8; ++i) {
        L.b00[i] = (_Float16)(0.01f * (float)((lane * 7 + i) % 13) - 0.05f);
        L.b01[i] = (_Float16)(0.02f * (float)((lane * 5 + i) % 11) - 0.1f);
        L.b10[i] = (_Float16)(0.01f * (float)((lane * 3 + i) % 17) - 0.07f);
        L.b11[i] = (_Float16)(0.03f * (float)((lane * 11 + i) % 7) - 0.09f);
    }
    uint32_t acc = 0;
    const bool role_matrix = V == 1 || (V == 3 && wave < 8);
    const bool role_signs = V == 2 || (V >= 3 && V != 4 && wave >= 8);
    const bool role_matrix_zero = V == 5 && wave < 8, role_matrix_acc = V == 6 && wave < 8, role_matrix_16 = V == 7 && wave < 8;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 r4[8];
    for (int i = 0; i < 8; ++i) r4[i] = f32x4{0, 0, 0, 0};
    f32x16 q0 = {0}, q1 = {0};
    for (int i = 0; i < 16; ++i) { q0[i] = (float)(lane - 3 * i); q1[i] = (float)(i * 5 - lane); }
    if (V == 4 && wave >= 8) {
        // half a group of sign extraction as the delay
        uint32_t w0 = 0;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) w0 = push_sign(w0, q0[k2]);
        acc ^= w0;
    }
    __syncthreads();
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = realtime();
    for (int t = 0; t < trips; ++t) {
        if (V == 0 || V == 4) {
            acc ^= filter_round_matrix(A, 0, BLOCKS, masks, lane, L);
        } else if (V == 8) {
            acc ^= filter_round_matrix_pipelined(A, 0, BLOCKS, masks, lane, L);
        } else if (role_matrix) {
            const h8 *P = A + lane;
            for (int blk = 0; blk < BLOCKS; ++blk, P += 256) {
                matrix_group(P[0], P[64], L, q0, q1);
                asm volatile("" : "+v"(q0), "+v"(q1));
                matrix_group(P[128], P[192], L, q0, q1);
                asm volatile("" : "+v"(q0), "+v"(q1));
            }
        } else if (role_matrix_zero) {
            const h8 *P = A + lane;
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int blk = 0; blk < 2 * BLOCKS; ++blk, P += 128) {
                const h8 a_lo = P[0], a_hi = P[64];
                __builtin_amdgcn_sched_barrier(0);
                q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, L.b00, zero, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, L.b10, zero, 0, 0, 0);
                asm volatile("" : "+v"(q0), "+v"(q1));
                q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, L.b01, zero, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, L.b11, zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+v"(q0), "+v"(q1));
            }
        } else if (role_matrix_acc) {
            const h8 *P = A + lane;
            for (int blk = 0; blk < 2 * BLOCKS; ++blk, P += 128) {
                const h8 a_lo = P[0], a_hi = P[64];
                __builtin_amdgcn_sched_barrier(0);
                q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, L.b00, q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, L.b10, q1, 0, 0, 0);
                q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, L.b01, q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, L.b11, q1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (role_matrix_16) {
            const h8 *P = A + lane;
            const f32x4 zero = {0, 0, 0, 0};
            for (int blk = 0; blk < 2 * BLOCKS; ++blk, P += 128) {
                const h8 a_lo = P[0], a_hi = P[64];
                __builtin_amdgcn_sched_barrier(0);
                r4[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, L.b00, zero, 0, 0, 0);
                r4[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, L.b10, zero, 0, 0, 0);
                r4[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, L.b01, zero, 0, 0, 0);
                r4[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, L.b11, zero, 0, 0, 0);
                r4[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, L.b00, zero, 0, 0, 0);
                r4[5] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, L.b10, zero, 0, 0, 0);
                r4[6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, L.b01, zero, 0, 0, 0);
                r4[7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, L.b11, zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]), "+v"(r4[4]), "+v"(r4[5]), "+v"(r4[6]), "+v"(r4[7]));
            }
        } else if (role_signs) {
            for (int blk = 0; blk < BLOCKS; ++blk) {
                uint32_t w0 = 0, w1 = 0;
                asm volatile("" : "+v"(q0), "+v"(q1));
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) { w0 = push_sign(w0, q0[k2]); w1 = push_sign(w1, q1[k2]); }
                asm volatile("" : "+v"(q0), "+v"(q1));
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) { w0 = push_sign(w0, q0[k2]); w1 = push_sign(w1, q1[k2]); }
                swap_halves(w0, w1);
                masks[(2 * blk) * 64 + lane] = ~w0;
                masks[(2 * blk + 1) * 64 + lane] = ~w1;
            }
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = realtime();
    for (int i = 0; i < 16; ++i) acc ^= __float_as_uint(q0[i]) ^ __float_as_uint(q1[i]);
    for (int i = 0; i < 8; ++i) acc ^= __float_as_uint(r4[i][0]) ^ __float_as_uint(r4[i][3]);
    if (acc == 0x12345u) sink[0] = acc;
    if (V == 0 || V == 8) {  // the candidate words of the last trip, for comparing the two forms
        __syncthreads();
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < 16 * 2 * BLOCKS * 64; i += blockDim.x) sink[16 + (V == 8 ? 16 * 2 * BLOCKS * 64 : 0) + i] = masks_all[i];
    }
    if (lane == 0) {
        atomicAdd(&out[0], (unsigned long long)(c1 - c0));
        atomicAdd(&out[1], (unsigned long long)(r1 - r0));
        atomicMax(&out[2], (unsigned long long)(c1 - c0));
        atomicAdd(&out[wave < 8 ? 3 : 4], (unsigned long long)(c1 - c0));
    }
}

template <int V>
void run(const h8 *A, int cus, int trips, unsigned long long *out, uint32_t *sink, const char *what) {
    const size_t lds = BLOCKS * 4096 + 16 * 2 * BLOCKS * 64 * 4;
    (void)hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V>), dim3(cus), dim3(1024), lds, 0, A, trips, out, sink);
    (void)hipDeviceSynchronize();
    (void)hipMemset(out, 0, 64);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), dim3(cus), dim3(1024), lds, 0, A, trips, out, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[5];
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    const double waves = (double)cus * 16, per = (double)trips * BLOCKS;
    const double clk = (double)h[0] / waves / per, rt_ticks = (double)h[1] / waves;
    // s_memrealtime ticks at 100 MHz
    const double mhz = (double)h[0] / waves / (rt_ticks / 100.0);
    printf("%-58s %8.1f clk/block/wave (slowest %8.1f; waves 0-7 %8.1f, waves 8-15 %8.1f)  %7.3f ms  shader clock %6.0f MHz\n",
           what, clk, (double)h[2] / per, (double)h[3] / (waves / 2) / per, (double)h[4] / (waves / 2) / per, ms, mhz);
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    // optional argument: number of workgroups (default one per CU); a few workgroups keep the chip far from its power limit
    const int cus = argc > 1 ? atoi(argv[1]) : p.multiProcessorCount;
    printf("%d workgroups of 1024 threads\n", cus);
    std::vector<_Float16> hA(BLOCKS * 256 * 8);
    srand(7);
    for (auto &x : hA) x = (_Float16)((rand() / (double)RAND_MAX - 0.5) * 4.0);
    h8 *A;
    unsigned long long *out;
    uint32_t *sink;
    (void)hipMalloc(&A, hA.size() * 2);
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&sink, 64 + 2 * 16 * 2 * BLOCKS * 64 * 4);
    (void)hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    const int trips = 4096;
    run<0>(A, cus, trips, out, sink, "0 filter as in the trace kernel");
    run<1>(A, cus, trips, out, sink, "1 matrix instructions only");
    run<2>(A, cus, trips, out, sink, "2 sign extraction only");
    run<3>(A, cus, trips, out, sink, "3 half the waves matrix only, half signs only");
    run<4>(A, cus, trips, out, sink, "4 filter, waves 8-15 half a group late");
    run<8>(A, cus, trips, out, sink, "8 filter, pipelined inside the wave");
    run<5>(A, cus, trips, out, sink, "5 as 3, matrix waves: zero C operand only");
    run<6>(A, cus, trips, out, sink, "6 as 3, matrix waves: accumulating only");
    run<7>(A, cus, trips, out, sink, "7 as 3, matrix waves: 16x16x32, zero C, 8 per group");
    {
        const int n = 16 * 2 * BLOCKS * 64;
        std::vector<uint32_t> h(2 * n);
        (void)hipMemcpy(h.data(), sink + 16, h.size() * 4, hipMemcpyDeviceToHost);
        int diff = 0, set = 0;
        for (int i = 0; i < n; ++i) { diff += h[i] != h[n + i]; set += __builtin_popcount(h[i]); }
        printf("candidate words of variant 8 against variant 0: %d of %d differ (%d bits set)\n", diff, n, set);
    }
    return 0;
}
