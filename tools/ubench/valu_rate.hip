// Micro-benchmark: issue rate of scalar vs packed f32 VALU ops on gfx950 (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x[8]; float2v y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = float2v{x[i], x[i] + 0.5f}; }
    const float2v a2{a, a}, b2{b, b};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
            if (MODE == 2) x[i] = x[i] * a;
            if (MODE == 3) y[i] = y[i] * a2;
            if (MODE == 4) x[i] = x[i] + a;
            if (MODE == 5) y[i] = y[i] + a2;
            if (MODE == 6) x[i] = (x[i] > b) ? x[i] - a : x[i] + a;
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int flops_per_op) {
    float* d; hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 blocks of 256 per CU = 32 waves/CU
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instrs = (double)blocks * 4 * ITERS * 8;
    const double per_simd_cycles = ms * 1e-3 * 2.4e9 / (wave_instrs / 1024.0);
    printf("%-14s %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4GHz)  %.1f TFLOP/s\n", name, ms, per_simd_cycles,
           wave_instrs * 64 * flops_per_op / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main() { run<0>("v_fma_f32", 2); run<1>("v_pk_fma_f32", 4); run<2>("v_mul_f32", 1); run<3>("v_pk_mul_f32", 2); run<4>("v_add_f32", 1); run<5>("v_pk_add_f32", 2); run<6>("cmp+cndmask", 1); return 0; }
