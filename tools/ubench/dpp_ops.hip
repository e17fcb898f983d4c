// Micro-benchmark: issue cost of DPP adds (the row reductions of block_reduce) and of sparse-lane LDS writes on gfx950
// (4 independent chains per wave, 8 waves/SIMD). Development aid; build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_ops.hip -o /tmp/dpp_ops && /tmp/dpp_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
#define BODY(ASM)                                                                                   \
    for (int it = 0; it < ITERS; ++it) {                                                            \
        asm volatile(ASM : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");        \
    }
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    __shared__ float lds[256 * 4];
    float x0 = threadIdx.x * 0.001f + 1.f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    if (MODE == 0) BODY("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4")
    if (MODE == 1) BODY("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
    if (MODE == 2) BODY("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
    if (MODE == 3) BODY("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
    if (MODE == 4) BODY("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
    if (MODE == 5) BODY("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
    if (MODE == 6) {  // what block_reduce does per value: 4 dependent DPP steps (16 per body: 4 values)
        BODY("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
             "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
             "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
             "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
    }
    if (MODE == 7) {  // sparse-lane LDS writes: 4 stores by the lanes (lane & 15) == 0, like the partial-sum hand-over
        float* p = lds + (threadIdx.x >> 4) * 4;
        for (int it = 0; it < ITERS; ++it) {
            if ((threadIdx.x & 15) == 0) {
                asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:4\n ds_write_b32 %0, %3 offset:8\n ds_write_b32 %0, %4 offset:12" ::"v"((unsigned)(size_t)p), "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory");
            }
            x0 += a;
        }
    }
    if (MODE == 8) {  // the same 4 values as ONE 16-byte store by those lanes
        float* p = lds + (threadIdx.x >> 4) * 4;
        for (int it = 0; it < ITERS; ++it) {
            if ((threadIdx.x & 15) == 0) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 q = {x0, x1, x2, x3};
                asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)p), "v"(q) : "memory");
            }
            x0 += a;
        }
    }
    if (MODE == 9) {  // all 64 lanes store one value each (the transposed hand-over)
        for (int it = 0; it < ITERS; ++it) {
            asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(threadIdx.x * 4)), "v"(x0) : "memory");
            x0 += a; x1 += a; x2 += a; x3 += a;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + lds[threadIdx.x];
}
template <int MODE> void run(const char* name, int per_body) {
    float* d; (void)hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instrs = (double)blocks * 4 * ITERS * per_body;
    printf("%-44s %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (wave_instrs / 1024.0));
    (void)hipFree(d);
}
int main() {
    run<0>("v_add_f32", 4); run<1>("v_add_f32_dpp quad_perm", 4); run<2>("v_add_f32_dpp row_half_mirror", 4); run<3>("v_add_f32_dpp row_mirror", 4);
    run<4>("v_add_f32_dpp row_ror:4", 4); run<5>("v_mov_b32_dpp quad_perm", 4); run<6>("row_sum16 x4 values (16 dpp adds)", 16);
    run<7>("4 ds_write_b32 by 4 of 64 lanes (+1 add)", 5); run<8>("1 ds_write_b128 by 4 of 64 lanes (+1 add)", 2); run<9>("1 ds_write_b32 all lanes (+4 adds)", 5);
    return 0;
}
