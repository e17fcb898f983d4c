// Micro-benchmark: packed f32 VALU instructions on gfx950 (4 independent chains per wave, 8 waves/SIMD). Development aid.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_ops.hip -o tools/ubench/pk_ops && tools/ubench/pk_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
typedef float f2 __attribute__((ext_vector_type(2)));
#define BODY(ASM)                                                                                   \
    for (int it = 0; it < ITERS; ++it) {                                                            \
        asm volatile(ASM : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");        \
    }
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float af, float bf) {
    f2 x0 = {threadIdx.x * 0.001f + 1.f, 2.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    const f2 a = {af, af * 1.01f}, b = {bf, bf * 0.99f};
    if (MODE == 0) BODY("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5")
    if (MODE == 1) BODY("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4")
    if (MODE == 2) BODY("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4")
    if (MODE == 3) BODY("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3")   // accumulate form: acc += a * b
    if (MODE == 4) BODY("v_pk_fma_f32 %0, %4, %5, %0 op_sel:[0,1,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel:[0,1,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %4, %5, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel:[0,1,0] op_sel_hi:[0,1,1]")
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0.x + x1.y + x2.x + x3.y;
}
template <int MODE>
__global__ __launch_bounds__(256) void ks(float* out, float a, float b) {  // scalar f32 reference
    float x0 = threadIdx.x * 0.001f + 1.f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    for (int it = 0; it < ITERS; ++it)
        asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <class K> void run(const char* name, K kern, int per_body) {
    float* d; (void)hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instrs = (double)blocks * 4 * ITERS * per_body;
    printf("%-44s %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (wave_instrs / 1024.0));
    (void)hipFree(d);
}
int main() {
    run("v_fma_f32 (acc += a*b)", ks<0>, 4);
    run("v_pk_fma_f32 (x = x*a+b)", k<0>, 4); run("v_pk_mul_f32", k<1>, 4); run("v_pk_add_f32", k<2>, 4);
    run("v_pk_fma_f32 (acc += a*b)", k<3>, 4); run("v_pk_fma_f32 op_sel broadcast a.lo", k<4>, 4);
    return 0;
}
