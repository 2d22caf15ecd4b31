// mfma_f16_probe.hip -- three facts about v_mfma_f32_32x32x16_f16 on this GPU that a matrix-core sphere filter
// depends on:
//  1. does it overlap with independent FP32 VALU work of the same and of other waves on the SIMD,
//  2. are FP16 subnormal inputs honoured or flushed,
//  3. how far is the FP32 result of a 16-term dot product from the exactly rounded sum (cancelling terms).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define ITERS 4096

template <int N, int M>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    h8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + i;
    for (int it = 0; it < ITERS; ++it) {
        // one MFMA, then a quarter of the VALU work, four times; the fences keep that order
        if constexpr (M >= 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < N / 4; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], 0.999f, 0.001f);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M >= 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = N / 4; i < N / 2; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], 0.999f, 0.001f);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M >= 3) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = N / 2; i < 3 * N / 4; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], 0.999f, 0.001f);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M >= 4) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 3 * N / 4; i < N; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], 0.999f, 0.001f);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int N, int M>
double run(float *out, int blocks) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N, M>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<N, M>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0 * 1e-3;
}

template <int N>
void row(float *out, int cus, int wps) {
    int blocks = cus * wps;
    double tm = run<N, 4>(out, blocks), tv = run<N, 0>(out, blocks);
    double cm = tm * 2.4e9 / ((double)ITERS * wps), cv = tv * 2.4e9 / ((double)ITERS * wps);
    printf("waves/SIMD %d  4 MFMA + %3d VALU per trip: %7.1f cyc/trip/wave, VALU only %7.1f\n", wps, N, cm, cv);
}

// D = A(32x16) * B(16x32): lane l supplies row l%32 of A and column l%32 of B, k = 8*(l/32) .. +7.
__global__ void one_mfma(const _Float16 *A, const _Float16 *B, float *D) {
    int l = threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = A[(l % 32) * 16 + 8 * (l / 32) + i];
        b[i] = B[(8 * (l / 32) + i) * 32 + (l % 32)];
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int kk = 0; kk < 16; ++kk) {
        int row = 8 * (kk / 4) + 4 * (l / 32) + (kk % 4);
        D[row * 32 + (l % 32)] = acc[kk];
    }
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    float *out;
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    for (int wps = 1; wps <= 4; wps *= 2) {
        row<0>(out, cus, wps);
        row<16>(out, cus, wps);
        row<32>(out, cus, wps);
        row<64>(out, cus, wps);
        row<96>(out, cus, wps);
        row<128>(out, cus, wps);
    }

    // 2 + 3: numerics
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    std::vector<float> D(32 * 32);
    _Float16 *dA, *dB;
    float *dD;
    (void)hipMalloc(&dA, A.size() * 2);
    (void)hipMalloc(&dB, B.size() * 2);
    (void)hipMalloc(&dD, D.size() * 4);
    auto go = [&]() {
        (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    };
    // subnormals: A[i][0] = 2^-24 * (i+1) (subnormal), B[0][j] = 1024 -> expect (i+1) * 2^-14
    for (auto &x : A) x = (_Float16)0.0f;
    for (auto &x : B) x = (_Float16)0.0f;
    for (int i = 0; i < 32; ++i) A[i * 16 + 0] = (_Float16)std::ldexp((float)(i + 1), -24);
    for (int j = 0; j < 32; ++j) B[0 * 32 + j] = (_Float16)1024.0f;
    go();
    int ok = 0;
    for (int i = 0; i < 32; ++i) ok += D[i * 32 + 5] == std::ldexp((float)(i + 1), -14);
    printf("subnormal A inputs honoured in %d of 32 rows (row 0 result %g, expected %g)\n", ok, D[5], std::ldexp(1.0f, -14));
    for (auto &x : A) x = (_Float16)0.0f;
    for (auto &x : B) x = (_Float16)0.0f;
    for (int i = 0; i < 32; ++i) A[i * 16 + 0] = (_Float16)1024.0f;
    for (int j = 0; j < 32; ++j) B[0 * 32 + j] = (_Float16)std::ldexp((float)(j + 1), -24);
    go();
    ok = 0;
    for (int j = 0; j < 32; ++j) ok += D[3 * 32 + j] == std::ldexp((float)(j + 1), -14);
    printf("subnormal B inputs honoured in %d of 32 columns\n", ok);

    // accuracy: random terms with heavy cancellation; error relative to the sum of |terms|
    srand(12345);
    double worst = 0, worst_rn = 0;
    for (int trial = 0; trial < 200; ++trial) {
        for (auto &x : A) x = (_Float16)((rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 12));
        for (auto &x : B) x = (_Float16)((rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 12));
        go();
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0, m = 0;
                for (int kk = 0; kk < 16; ++kk) {
                    double t = (double)A[i * 16 + kk] * (double)B[kk * 32 + j];
                    s += t;
                    m += std::fabs(t);
                }
                double e = std::fabs((double)D[i * 32 + j] - s) / m;
                if (e > worst) worst = e;
                double ern = std::fabs((double)(float)s - s) / m;
                if (ern > worst_rn) worst_rn = ern;
            }
    }
    printf("worst |mfma - exact| / sum|terms| = %.3g (= %.2f x 2^-24); exactly rounded sum would give %.3g\n", worst,
           worst / std::ldexp(1.0, -24), worst_rn);
    return 0;
}
