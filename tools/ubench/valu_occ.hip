// Micro-benchmark: f32 FMA issue rate vs waves/SIMD and per-wave ILP on gfx950 (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 8192
template <int ILP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0; for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP> void run(int waves_per_simd) {
    float* d; (void)hipMalloc(&d, 256 * 4096 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = 1 wave per SIMD each
    hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd_instrs = (double)waves_per_simd * ITERS * ILP;
    printf("waves/SIMD %d ILP %d: %.3f ms  %.2f cycles per wave-instr per SIMD (2.4 GHz)\n", waves_per_simd, ILP, ms, ms * 1e-3 * 2.4e9 / per_simd_instrs);
    (void)hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
