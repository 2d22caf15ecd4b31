// valu_rate.hip -- issue-rate microbenchmark for the FP32 vector instructions the sphere loop is built from:
// v_fma_f32, v_mul_f32+v_add_f32, v_pk_fma_f32, v_pk_mul_f32+v_pk_add_f32, at 1..8 waves per SIMD.
// Prints wave-instructions per cycle per SIMD (at the measured wall time and a nominal 2.4 GHz) and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAINS 8
#define ITERS 65536

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float x = seed + threadIdx.x * 1e-6f;
    if constexpr (KIND == 0) {  // v_fma_f32
        float a[CHAINS];
        for (int i = 0; i < CHAINS; ++i) a[i] = x + i;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
        }
        float s = 0;
        for (int i = 0; i < CHAINS; ++i) s += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (KIND == 1) {  // v_mul_f32 + v_add_f32
        float a[CHAINS];
        for (int i = 0; i < CHAINS; ++i) a[i] = x + i;
        for (int it = 0; it < ITERS / 2; ++it) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) a[i] = a[i] * 0.999f;
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) a[i] = a[i] + 0.001f;
        }
        float s = 0;
        for (int i = 0; i < CHAINS; ++i) s += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (KIND == 2) {  // v_pk_fma_f32
        f2 a[CHAINS];
        for (int i = 0; i < CHAINS; ++i) a[i] = (f2){x + i, x - i};
        f2 m = {0.999f, 0.998f}, c = {0.001f, 0.002f};
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
        }
        f2 s = {0, 0};
        for (int i = 0; i < CHAINS; ++i) s += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else {  // v_pk_mul_f32 + v_pk_add_f32
        f2 a[CHAINS];
        for (int i = 0; i < CHAINS; ++i) a[i] = (f2){x + i, x - i};
        f2 m = {0.999f, 0.998f}, c = {0.001f, 0.002f};
        for (int it = 0; it < ITERS / 2; ++it) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) a[i] = a[i] * m;
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) a[i] = a[i] + c;
        }
        f2 s = {0, 0};
        for (int i = 0; i < CHAINS; ++i) s += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    }
}

template <int KIND>
double run(float *out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0 * 1e-3;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    float *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    const char *names[4] = {"v_fma_f32", "v_mul+v_add", "v_pk_fma_f32", "v_pk_mul+v_pk_add"};
    const double flops_per_lane_instr[4] = {2, 1, 4, 2};
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD = blocks per CU (256 threads = 4 waves = 1 per SIMD)
        int blocks = cus * wps;
        double t[4] = {run<0>(out, blocks), run<1>(out, blocks), run<2>(out, blocks), run<3>(out, blocks)};
        for (int kind = 0; kind < 4; ++kind) {
            double winstr = (double)blocks * 4 * (double)ITERS * CHAINS;  // wave-instructions
            double per_simd_per_s = winstr / t[kind] / (cus * 4.0);
            double tflops = winstr * 64 * flops_per_lane_instr[kind] / t[kind] * 1e-12;
            printf("waves/SIMD %d  %-18s %8.3f ms  %.3f wave-instr/clk/SIMD @2.4GHz  %7.1f TFLOP/s\n", wps, names[kind],
                   t[kind] * 1e3, per_simd_per_s / 2.4e9, tflops);
        }
    }
    return 0;
}
