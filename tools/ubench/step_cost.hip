// Micro-benchmark (development aid): shader-clock cycles of ONE lm_step_fast (6x6 Cholesky + solves + se3::exp + pose update, FUSED arithmetic)
// as the per-pair kernel runs it: a dependent chain on one wavefront. The function text is extracted from lm_kernels.hip by tools/ubench/make_step_cost.py.
#include <hip/hip_runtime.h>
#include "/root/repo/visual-odometry-rs_amd/csrc/lie.h"
namespace vors {
__device__ __forceinline__ bool lm_step_fast(const float* h, const float* g, const Iso& model, float lm_coef, Iso* out) {
    float a[6][6], inv[6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) a[r][c] = h[r * 6 + c];
    const float scale = 1.0f + lm_coef;
#pragma unroll
    for (int r = 0; r < 6; ++r) a[r][r] *= scale;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k < j; ++k) {
#pragma unroll
            for (int i = j; i < 6; ++i) a[i][j] = fmaf(-a[j][k], a[i][k], a[i][j]);
        }
        const float diag = a[j][j];
        ok = ok && (diag > 0.0f);
        inv[j] = __builtin_amdgcn_rsqf(diag);
        a[j][j] = diag * inv[j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) a[i][j] *= inv[j];
    }
    if (!ok) return false;
    float b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = g[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        b[i] *= inv[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) b[k] = fmaf(-b[i], a[k][i], b[k]);
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        float acc = b[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) acc = fmaf(-a[k][i], b[k], acc);
        b[i] = acc * inv[i];
    }
    // se3::exp (se3.rs:65-95) as in lie.h, from_quaternion by reciprocal square root
    const float vx = b[0], vy = b[1], vz = b[2], wx = b[3], wy = b[4], wz = b[5];
    const float theta_2 = fmaf(wz, wz, fmaf(wy, wy, wx * wx));
    float real_factor, imag_factor, c1, c2;
    if (theta_2 < 1e-2f * 1e-2f) {
        real_factor = fmaf(-0.125f, theta_2, 1.0f);
        imag_factor = fmaf(-(1.0f / 48.0f), theta_2, 0.5f);
        c1 = fmaf(-(1.0f / 24.0f), theta_2, 0.5f);
        c2 = fmaf(-(1.0f / 120.0f), theta_2, 1.0f / 6.0f);
    } else {
        const float theta = sqrtf(theta_2), half_theta = 0.5f * theta, rt2 = 1.0f / theta_2;
        real_factor = cosf(half_theta);
        imag_factor = sinf(half_theta) / theta;
        const float sh = sinf(half_theta);
        c1 = 2.0f * sh * sh * rt2;  // (1 - cos t) / t^2 without the cancellation
        c2 = (theta - sinf(theta)) * rt2 / theta;
    }
    const float w11 = wx * wx, w12 = wx * wy, w13 = wx * wz, w22 = wy * wy, w23 = wy * wz, w33 = wz * wz;
    Iso dw;
    dw.t.x = fmaf(fmaf(c2, -w22 - w33, 1.0f), vx, fmaf(fmaf(c2, w12, -c1 * wz), vy, fmaf(c2, w13, c1 * wy) * vz));
    dw.t.y = fmaf(fmaf(c2, w12, c1 * wz), vx, fmaf(fmaf(c2, -w11 - w33, 1.0f), vy, fmaf(c2, w23, -c1 * wx) * vz));
    dw.t.z = fmaf(fmaf(c2, w13, -c1 * wy), vx, fmaf(fmaf(c2, w23, c1 * wx), vy, fmaf(c2, -w11 - w22, 1.0f) * vz));
    const Quat q{imag_factor * wx, imag_factor * wy, imag_factor * wz, real_factor};
    const float rn = __builtin_amdgcn_rsqf(quat_norm_squared(q));
    dw.q = Quat{q.i * rn, q.j * rn, q.k * rn, q.w * rn};
    *out = renormalize(iso_mul(model, iso_inverse(dw)));
    return true;
}


__global__ void bench_fast(const float* in, float* out, int iters, int one_lane) {
    float h[36], g[6];
    for (int q = 0; q < 6; ++q) g[q] = in[2 + q];
    int k = 8;
    for (int q = 0; q < 6; ++q) for (int r = q; r < 6; ++r) { h[q * 6 + r] = in[k]; h[r * 6 + q] = in[k]; ++k; }
    Iso model = iso_load(in + 32);
    const float lam = in[40];
    long long t0 = clock64();
    if (!one_lane || threadIdx.x % 64 == 0) {
        for (int it = 0; it < iters; ++it) {
            Iso cand;
            if (lm_step_fast(h, g, model, lam, &cand)) { model = cand; g[0] += cand.t.x * 1e-20f; }
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { iso_store(model, out + blockIdx.x * 16); out[blockIdx.x * 16 + 8] = (float)(t1 - t0) / iters; }
}
}
#include <cstdio>
#include <vector>
int main() {
    std::vector<float> in(64, 0.f);
    // a well-conditioned SPD H, small g
    float H[6][6] = {{0}}; for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H[i][j] = (i == j ? 1000.f + 100.f * i : 10.f / (1 + i + j));
    for (int q = 0; q < 6; ++q) in[2 + q] = 0.5f + q * 0.1f;
    int k = 8; for (int q = 0; q < 6; ++q) for (int r = q; r < 6; ++r) in[k++] = H[q][r];
    in[32 + 6] = 1.f; in[40] = 0.1f;
    float *di, *dout; hipMalloc(&di, 64 * 4); hipMalloc(&dout, 4096 * 16 * 4); hipMemcpy(di, in.data(), 64 * 4, hipMemcpyHostToDevice);
    for (int one = 0; one < 2; ++one) for (int waves : {1, 2, 4}) {
        hipLaunchKernelGGL(vors::bench_fast, dim3(256 * waves), dim3(256), 0, 0, di, dout, 2000, one);
        hipDeviceSynchronize();
        float o[16]; hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost);
        printf("lm_step_fast: %s, %d waves/SIMD: %.0f shader-clock cycles per step (t = %g %g %g)\n", one ? "one lane" : "all lanes", waves, o[8], o[0], o[1], o[2]);
    }
    return 0;
}
