#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define NACC 29
template <int CTRL>
__device__ __forceinline__ float dpp_add_row(float v) {
    const int t = __builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add_row<0xB1>(v); v = dpp_add_row<0x4E>(v); v = dpp_add_row<0x141>(v); v = dpp_add_row<0x140>(v); return v;
}
__global__ void k(float* out_old, float* out_new) {
    __shared__ alignas(16) float part[8 * 32];
    float acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = sinf(threadIdx.x * 0.37f + i * 1.3f) * (1.0f + i);
    // old
    float r[NACC];
    for (int i = 0; i < NACC; ++i) r[i] = row_sum16(acc[i]);
    if ((threadIdx.x & 15) == 0) for (int i = 0; i < NACC; ++i) part[(threadIdx.x >> 4) * 32 + i] = r[i];
    __syncthreads();
    if (threadIdx.x < NACC) { float t = 0.f; for (int w = 0; w < 8; ++w) t += part[w * 32 + threadIdx.x]; out_old[threadIdx.x] = t; }
    __syncthreads();
    // new
    float a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = i < NACC ? acc[i] : 0.f;
    float b[16], c[8];
    const bool odd = (threadIdx.x & 1) != 0, hi = (threadIdx.x & 2) != 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {  // lane ^ 1: an even lane keeps sum 2j, an odd lane sum 2j+1; each gives the other one away
        const float keep = odd ? a[2 * j + 1] : a[2 * j], give = odd ? a[2 * j] : a[2 * j + 1];
        b[j] = keep + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(give), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // lane ^ 2: lanes 0, 1 of a quad keep b[2k], lanes 2, 3 keep b[2k+1]
        const float keep = hi ? b[2 * k + 1] : b[2 * k], give = hi ? b[2 * k] : b[2 * k + 1];
        c[k] = keep + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(give), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(c[k]), 0x104, 0xf, 0xf, true));  // row_shl:4: Q0 + Q1
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(c[k]), 0x108, 0xf, 0xf, true));  // row_shl:8: + (Q2 + Q3)
    if ((threadIdx.x & 15) < 4) {
        float4* p = reinterpret_cast<float4*>(part + (threadIdx.x >> 4) * 32 + (threadIdx.x & 3) * 8);
        p[0] = make_float4(c[0], c[1], c[2], c[3]);
        p[1] = make_float4(c[4], c[5], c[6], c[7]);
    }
    __syncthreads();
    if (threadIdx.x < NACC) { const int slot = (threadIdx.x & 3) * 8 + (threadIdx.x >> 2); float t = 0.f; for (int w = 0; w < 8; ++w) t += part[w * 32 + slot]; out_new[threadIdx.x] = t; }
}
int main() {
    float *a, *b; hipMalloc(&a, 128); hipMalloc(&b, 128);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, a, b);
    float ha[32], hb[32]; hipMemcpy(ha, a, 116, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 116, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < NACC; ++i) { if (memcmp(&ha[i], &hb[i], 4)) { ++bad; printf("i %d old %.9g new %.9g\n", i, ha[i], hb[i]); } }
    printf("mismatches: %d\n", bad);
}
