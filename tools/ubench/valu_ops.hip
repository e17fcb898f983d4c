// Micro-benchmark: issue cost of individual gfx950 VALU instructions (4 independent chains per wave, 8 waves/SIMD). Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
#define BODY(ASM)                                                                                   \
    for (int it = 0; it < ITERS; ++it) {                                                            \
        asm volatile(ASM : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "s"(sa) : "vcc"); \
    }
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, float sa_in) {
    float x0 = threadIdx.x * 0.001f + 1.f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    const float sa = __builtin_amdgcn_readfirstlane(__float_as_int(sa_in)) * 1.0f;
    if (MODE == 0) BODY("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5")
    if (MODE == 1) BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3")
    if (MODE == 2) BODY("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3")
    if (MODE == 3) BODY("v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte1 %2, %2\n v_cvt_f32_ubyte1 %3, %3")
    if (MODE == 4) BODY("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3")
    if (MODE == 5) BODY("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc")
    if (MODE == 6) BODY("v_mad_u64_u32 v[10:11], vcc, %0, %4, v[10:11]\n v_mad_u64_u32 v[12:13], vcc, %1, %4, v[12:13]\n v_mad_u64_u32 v[14:15], vcc, %2, %4, v[14:15]\n v_mad_u64_u32 v[16:17], vcc, %3, %4, v[16:17]")
    if (MODE == 7) BODY("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5")
    if (MODE == 8) BODY("v_sub_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_sub_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_sub_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_sub_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0")
    if (MODE == 9) BODY("v_mul_f32 %0, %0, %6\n v_mul_f32 %1, %1, %6\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %6")
    if (MODE == 10) BODY("v_cndmask_b32 %0, %0, %5, s[10:11]\n v_cndmask_b32 %1, %1, %5, s[10:11]\n v_cndmask_b32 %2, %2, %5, s[10:11]\n v_cndmask_b32 %3, %3, %5, s[10:11]")
    if (MODE == 11) BODY("v_cmp_lt_f32 s[10:11], %0, %4\n v_cmp_lt_f32 s[12:13], %1, %4\n v_cmp_lt_f32 s[14:15], %2, %4\n v_cmp_lt_f32 s[16:17], %3, %4")
    if (MODE == 12) BODY("v_lshrrev_b32 %0, 8, %0\n v_lshrrev_b32 %1, 8, %1\n v_lshrrev_b32 %2, 8, %2\n v_lshrrev_b32 %3, 8, %3")
    if (MODE == 13) BODY("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3")
    if (MODE == 14) BODY("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8")
    if (MODE == 15) BODY("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc")
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int MODE> void run(const char* name, int per_body) {
    float* d; (void)hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 1.00001f);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 1.00001f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instrs = (double)blocks * 4 * ITERS * per_body;
    printf("%-22s %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (wave_instrs / 1024.0));
    (void)hipFree(d);
}
int main() {
    run<0>("v_fma_f32", 4); run<9>("v_mul_f32 (sgpr)", 4); run<1>("v_rcp_f32", 4); run<2>("v_floor_f32", 4); run<3>("v_cvt_f32_ubyte1", 4);
    run<4>("v_cvt_i32_f32", 4); run<13>("v_cvt_f32_i32", 4); run<5>("v_cmp+v_cndmask (vcc)", 8); run<10>("v_cndmask (sgpr mask)", 4);
    run<11>("v_cmp -> sgpr pair", 4); run<6>("v_mad_u64_u32", 4); run<7>("v_mad_u32_u24", 4); run<8>("v_sub_u32_sdwa", 4);
    run<12>("v_lshrrev_b32", 4); run<14>("v_bfe_u32", 4); run<15>("v_addc_co_u32", 4);
    return 0;
}
