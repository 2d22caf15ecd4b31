// mfma_split_probe.hip -- do a matrix-only wave and a vector-only wave that share a SIMD run concurrently?
// 512-thread workgroups put waves w and w + 4 on the same SIMD.  Waves 0-3 loop over v_mfma_f32_32x32x16_f16,
// waves 4-7 over independent v_fma_f32; each role is timed alone and then both together.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define ITERS 8192

__global__ __launch_bounds__(512) void k(float *out, float seed, int do_mfma, int do_valu, int czero) {
    const int wave = threadIdx.x >> 6;
    float s = 0;
    if (wave < 4) {
        if (do_mfma) {
            f32x16 acc0 = {0}, acc1 = {0}, zero = {0};
            h8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
            if (czero) {
                for (int it = 0; it < ITERS; ++it) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
                    asm volatile("" : "+v"(acc0), "+v"(acc1));
                }
            } else {
                for (int it = 0; it < ITERS; ++it) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                }
            }
            for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
        }
    } else if (do_valu == 1) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = seed + i;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], 0.999f, 0.001f);
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    } else if (do_valu == 2) {  // integer bit shuffling instead of floating point: v_alignbit_b32
        unsigned v[16];
        for (int i = 0; i < 16; ++i) v[i] = __float_as_uint(seed) + i * 77u + threadIdx.x;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i % 16] = __builtin_amdgcn_alignbit(v[i % 16], v[(i + 1) % 16], 31);
        }
        for (int i = 0; i < 16; ++i) s += (float)v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

double run(float *out, int blocks, int m, int v, int cz) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, 1.0f, m, v, cz);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, 1.0f, m, v, cz);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0;
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    float *out;
    (void)hipMalloc(&out, (size_t)cus * 2 * 512 * sizeof(float));
    for (int per_cu = 1; per_cu <= 2; ++per_cu)
        for (int cz = 0; cz <= 1; ++cz) {
            int blocks = cus * per_cu;
            double tm = run(out, blocks, 1, 0, cz), tv = run(out, blocks, 0, 1, cz), tb = run(out, blocks, 1, 1, cz);
            printf("%d workgroup(s)/CU, C operand %s: matrix waves alone %.3f ms, vector waves alone %.3f ms, together %.3f ms\n", per_cu,
                   cz ? "zero" : "accumulator", tm, tv, tb);
            double ti = run(out, blocks, 0, 2, cz), tbi = run(out, blocks, 1, 2, cz);
            printf("   the same with v_alignbit_b32 as the vector work: alone %.3f ms, together %.3f ms\n", ti, tbi);
        }
    return 0;
}
