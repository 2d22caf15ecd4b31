// mfma_coexec.hip -- does an FP32 MFMA (v_mfma_f32_32x32x2_f32) overlap with independent FP32 VALU work issued by
// the same and by other waves of a SIMD?  Each wave loops { 1 MFMA ; N independent v_fma_f32 } and the wall time
// per trip is reported for several N and occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITERS 2048

template <int N, bool MFMA>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    f32x16 acc = {0};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + i;
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < N; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], 0.999f, 0.001f);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int N, bool MFMA>
double run(float *out, int blocks) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N, MFMA>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<N, MFMA>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0 * 1e-3;
}

template <int N>
void row(float *out, int cus, int wps) {
    int blocks = cus * wps;
    double tm = run<N, true>(out, blocks), tv = run<N, false>(out, blocks);
    // cycles per trip per SIMD at a nominal 2.4 GHz: each SIMD runs wps waves, each doing ITERS trips
    double cm = tm * 2.4e9 / ((double)ITERS * wps), cv = tv * 2.4e9 / ((double)ITERS * wps);
    printf("waves/SIMD %d  N=%2d VALU per MFMA: with MFMA %6.1f cyc/trip/wave, VALU only %6.1f cyc/trip/wave\n", wps, N, cm, cv);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    float *out;
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    for (int wps = 1; wps <= 4; wps *= 2) {
        row<0>(out, cus, wps);
        row<8>(out, cus, wps);
        row<16>(out, cus, wps);
        row<24>(out, cus, wps);
        row<32>(out, cus, wps);
        row<48>(out, cus, wps);
        row<64>(out, cus, wps);
    }
    return 0;
}
