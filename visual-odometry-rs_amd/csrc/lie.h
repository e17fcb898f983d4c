// Rigid-motion arithmetic of the tracker, host + device (f32, compile with -ffp-contract=off).
//
// Implements what the reference gets from src/math/se3.rs, src/math/so3.rs and from nalgebra 0.17
// (Isometry3 / UnitQuaternion / 6x6 Cholesky). Operation ORDER matters: the LM loop branches on f32
// comparisons, so every expression below is written in the order the reference evaluates it.
// nalgebra itself is not vendored in the reference; its evaluation orders are restated from its source
// (see DESIGN.md "nalgebra assumptions").
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VORS_HD __host__ __device__ inline
#else
#define VORS_HD inline
#endif

namespace vors {

struct V3 {
    float x, y, z;
};
struct Quat {  // nalgebra coords order [i, j, k, w]
    float i, j, k, w;
};
struct Iso {  // Isometry3<f32>
    V3 t;
    Quat q;
};

VORS_HD Iso iso_identity() { return Iso{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 1.f}}; }
VORS_HD Iso iso_load(const float* p) { return Iso{{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}}; }
VORS_HD void iso_store(const Iso& m, float* p) {
    p[0] = m.t.x; p[1] = m.t.y; p[2] = m.t.z;
    p[3] = m.q.i; p[4] = m.q.j; p[5] = m.q.k; p[6] = m.q.w;
}

VORS_HD V3 cross(const V3& a, const V3& b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
VORS_HD float dot3(const V3& a, const V3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
VORS_HD float quat_norm_squared(const Quat& q) {
    float a = q.i * q.i, b = q.j * q.j;
    const float c = q.k * q.k, d = q.w * q.w;
    a += c;
    b += d;
    return a + b;
}
// UnitQuaternion * Vector3:  t = 2 (qv x p);  p' = (t w + qv x t) + p
VORS_HD V3 quat_rotate(const Quat& q, const V3& p) {
    const V3 qv{q.i, q.j, q.k};
    V3 t = cross(qv, p);
    t = V3{t.x * 2.0f, t.y * 2.0f, t.z * 2.0f};
    const V3 c = cross(qv, t);
    return V3{(t.x * q.w + c.x) + p.x, (t.y * q.w + c.y) + p.y, (t.z * q.w + c.z) + p.z};
}
VORS_HD Quat quat_mul(const Quat& a, const Quat& b) {
    Quat r;
    r.w = a.w * b.w - dot3(V3{a.i, a.j, a.k}, V3{b.i, b.j, b.k});
    r.i = a.w * b.i + a.i * b.w + a.j * b.k - a.k * b.j;
    r.j = a.w * b.j - a.i * b.k + a.j * b.w + a.k * b.i;
    r.k = a.w * b.k + a.i * b.j - a.j * b.i + a.k * b.w;
    return r;
}
VORS_HD Quat unit_from_quaternion(const Quat& q) {
    const float n = sqrtf(quat_norm_squared(q));
    return Quat{q.i / n, q.j / n, q.k / n, q.w / n};
}
VORS_HD V3 iso_transform_point(const Iso& m, const V3& p) {
    const V3 r = quat_rotate(m.q, p);
    return V3{r.x + m.t.x, r.y + m.t.y, r.z + m.t.z};
}
VORS_HD Iso iso_mul(const Iso& a, const Iso& b) {
    const V3 s = quat_rotate(a.q, b.t);
    return Iso{V3{a.t.x + s.x, a.t.y + s.y, a.t.z + s.z}, quat_mul(a.q, b.q)};
}
VORS_HD Iso iso_inverse(const Iso& a) {
    const Quat qi{-a.q.i, -a.q.j, -a.q.k, a.q.w};
    return Iso{quat_rotate(qi, V3{-a.t.x, -a.t.y, -a.t.z}), qi};
}
// reference: src/core/track/lm_optimizer.rs:198-209 (first-order re-normalisation)
VORS_HD Iso renormalize(Iso m) {
    const float f = 0.5f * (3.0f - quat_norm_squared(m.q));
    m.q = Quat{f * m.q.i, f * m.q.j, f * m.q.k, f * m.q.w};
    return m;
}

// sinf / cosf as the reference computes them: Rust's f32::sin / cos call the platform libm, i.e. glibc's sinf / cosf (>= 2.28: the
// ARM optimized-routines algorithm — argument widened to f64, |x| < pi/4: odd / even polynomial in f64; otherwise x - n pi/2 with
// n = round(x 2/pi) and the polynomial selected by the quadrant — rounded ONCE to f32). That function is NOT the correctly rounded
// sine (1 % of the arguments in [1e-3, 4] differ from RN(sin x) by one ulp), and the device's ocml sinf is a third function, so the
// algorithm is restated here, host + device: bit-identical to glibc 2.35's sinf / cosf for EVERY f32 in [2^-12, 4) — with or
// without FMA contraction of the f64 polynomial (checked exhaustively against the platform libm through vors_ref_sincos: tests/test_capi_host.py). se3::exp
// only calls them with theta / 2 and theta, theta >= 0.01 (se3.rs:82-87); arguments >= 4 rad go to the platform function.
VORS_HD float ref_sincos_poly(double x, int n) {
    const double x2 = x * x;
    if ((n & 1) == 0) {
        const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
        const double x3 = x * x2, s1 = S2 + x2 * S3, x7 = x3 * x2, s = x + x3 * S1;
        return (float)(s + x7 * s1);
    }
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2, c2 = C3 + x2 * C4, c1 = C0 + x2 * C1, x6 = x4 * x2, c = c1 + x4 * C2;
    return (float)(c + x6 * c2);
}
VORS_HD float ref_sincos(float y, int quadrant_shift) {  // 0: sin, 1: cos; 0 <= y < 4
    double x = (double)y;
    int n = 0;
    if (!(y < 0x1.921fb6p-1f)) {  // pi/4
        const double r = x * 0x1.45F306DC9C883p+23;  // x * 2/pi * 2^24
        n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * 0x1.921FB54442D18p0;
    } else if (y < 0x1p-12f) {
        return quadrant_shift ? 1.0f : y;
    }
    n += quadrant_shift;
    const float v = ref_sincos_poly((n & 2) && !(n & 1) ? -x : x, n);
    return ((n & 2) && (n & 1)) ? -v : v;
}
VORS_HD float ref_sinf(float y) { return (y >= 0.f && y < 4.0f) ? ref_sincos(y, 0) : sinf(y); }
VORS_HD float ref_cosf(float y) { return (y >= 0.f && y < 4.0f) ? ref_sincos(y, 1) : cosf(y); }

// reference: src/math/se3.rs:65-95 with so3::hat / hat_2 (src/math/so3.rs:27-51) expanded in place.
VORS_HD Iso se3_exp(const float xi[6]) {
    const float vx = xi[0], vy = xi[1], vz = xi[2];
    const float wx = xi[3], wy = xi[4], wz = xi[5];
    const float theta_2 = (wx * wx + wy * wy) + wz * wz;
    float real_factor, imag_factor, c1, c2;
    if (theta_2 < 1e-2f * 1e-2f) {
        real_factor = 1.0f - 0.125f * theta_2;
        imag_factor = 0.5f - (1.0f / 48.0f) * theta_2;
        c1 = 0.5f - (1.0f / 24.0f) * theta_2;
        c2 = (1.0f / 6.0f) - (1.0f / 120.0f) * theta_2;
    } else {
        const float theta = sqrtf(theta_2);
        const float half_theta = 0.5f * theta;
        real_factor = ref_cosf(half_theta);
        imag_factor = ref_sinf(half_theta) / theta;
        c1 = (1.0f - ref_cosf(theta)) / theta_2;
        c2 = (theta - ref_sinf(theta)) / (theta * theta_2);
    }
    const float w11 = wx * wx, w12 = wx * wy, w13 = wx * wz, w22 = wy * wy, w23 = wy * wz, w33 = wz * wz;
    const float O[3][3] = {{0.0f, -wz, wy}, {wz, 0.0f, -wx}, {-wy, wx, 0.0f}};
    const float O2[3][3] = {{-w22 - w33, w12, w13}, {w12, -w11 - w33, w23}, {w13, w23, -w11 - w22}};
    float V[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) V[r][c] = ((r == c ? 1.0f : 0.0f) + c1 * O[r][c]) + c2 * O2[r][c];
    Iso out;
    out.t.x = (V[0][0] * vx + V[0][1] * vy) + V[0][2] * vz;
    out.t.y = (V[1][0] * vx + V[1][1] * vy) + V[1][2] * vz;
    out.t.z = (V[2][0] * vx + V[2][1] * vy) + V[2][2] * vz;
    out.q = unit_from_quaternion(Quat{imag_factor * wx, imag_factor * wy, imag_factor * wz, real_factor});
    return out;
}

// 6x6 Cholesky (lower, left-looking) + solve, the order nalgebra's Cholesky::new / ::solve use.
// H is the full symmetric matrix, row-major h[r*6+c]; only the lower triangle is read.
// Returns false when a pivot is not > 0 (nalgebra returns None -> Err at lm_optimizer.rs:131-133).
VORS_HD bool cholesky6_solve(const float* h, const float* g, float lm_coef, float* delta) {
    float a[6][6];
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) a[r][c] = h[r * 6 + c];
    const float scale = 1.0f + lm_coef;  // lm_optimizer.rs:125-130
    for (int r = 0; r < 6; ++r) a[r][r] *= scale;
    for (int j = 0; j < 6; ++j) {
        for (int k = 0; k < j; ++k) {
            const float factor = -a[j][k];
            for (int i = j; i < 6; ++i) a[i][j] = factor * a[i][k] + a[i][j];
        }
        const float diag = a[j][j];
        if (!(diag > 0.0f)) return false;
        const float denom = sqrtf(diag);
        a[j][j] = denom;
        for (int i = j + 1; i < 6; ++i) a[i][j] /= denom;
    }
    float b[6];
    for (int i = 0; i < 6; ++i) b[i] = g[i];
    for (int i = 0; i < 6; ++i) {
        const float coeff = b[i] / a[i][i];
        b[i] = coeff;
        for (int k = i + 1; k < 6; ++k) b[k] = (-coeff) * a[k][i] + b[k];
    }
    for (int i = 5; i >= 0; --i) {
        float dot = 0.0f;
        for (int k = i + 1; k < 6; ++k) dot += a[k][i] * b[k];
        b[i] = (b[i] - dot) / a[i][i];
    }
    for (int i = 0; i < 6; ++i) delta[i] = b[i];
    return true;
}

// One LM step: lm_optimizer.rs:123-136.
VORS_HD bool lm_step(const float* h36, const float* g6, const Iso& model, float lm_coef, Iso* out) {
    float delta[6];
    if (!cholesky6_solve(h36, g6, lm_coef, delta)) return false;
    const Iso dw = se3_exp(delta);
    *out = renormalize(iso_mul(model, iso_inverse(dw)));
    return true;
}

// Per-level pinhole intrinsics; reference: src/core/camera.rs:84-140.
struct Intr {
    float cu, cv, fu, fv, skew;
};
VORS_HD Intr intr_half_res(const Intr& k) {  // camera.rs:115-123 (skew left unscaled, as in the reference)
    return Intr{(k.cu + 0.5f) / 2.0f - 0.5f, (k.cv + 0.5f) / 2.0f - 0.5f, 0.5f * k.fu, 0.5f * k.fv, k.skew};
}
VORS_HD V3 back_project(const Intr& k, float px, float py, float depth) {  // camera.rs:135-140
    const float z = depth;
    const float y = (py - k.cv) * z / k.fv;
    const float x = ((px - k.cu) * z - k.skew * y) / k.fu;
    return V3{x, y, z};
}
// project + perspective division: camera.rs:126-132 and lm_optimizer.rs:217-218.
VORS_HD void project_uv(const Intr& k, const V3& p, float* u, float* v) {
    const float pu = (k.fu * p.x + k.skew * p.y) + k.cu * p.z;
    const float pv = k.fv * p.y + k.cv * p.z;
    *u = pu / p.z;
    *v = pv / p.z;
}
// Jacobian of the warp: src/core/track/inverse_compositional.rs:313-341.
VORS_HD void warp_jacobian_at(float gu, float gv, float u, float v, float _z, const Intr& k, float J[6]) {
    const float a = u - k.cu;
    const float b = v - k.cv;
    const float c = a * k.fv - k.skew * b;
    const float _fv = 1.0f / k.fv;
    const float _fuv = 1.0f / (k.fu * k.fv);
    J[0] = gu * _z * k.fu;
    J[1] = _z * (gu * k.skew + gv * k.fv);
    J[2] = -_z * (gu * a + gv * b);
    J[3] = gu * (-a * b * _fv - k.skew) + gv * (-b * b * _fv - k.fv);
    J[4] = gu * (a * c * _fuv + k.fu) + gv * (b * c * _fuv);
    J[5] = gu * (-k.fu * k.fu * b + k.skew * c) * _fuv + gv * (c / k.fu);
}

// Division by a value that is uniform over a level (the focal lengths): x / d == fma(fma(-q, d, x), r, q) with r = RN(1/d),
// q = RN(x r), whenever `ok` — and `ok` is only set after the identity has been checked EXHAUSTIVELY for this d over all 2^23
// significands of x on the device (kernels.hip, verify_fastdiv_kernel): both sides scale exactly with the exponent of x and
// are odd in x, so one binade proves every finite x whose quotient stays in the normal range. Otherwise: IEEE division.
struct FastDiv {
    float d, r;
    int ok;
};
template <bool FAST>
VORS_HD float div_uniform(float x, const FastDiv& f) {
    if (FAST) {
        const float q = x * f.r;
        const float e = fmaf(-q, f.d, x);
        const float q1 = fmaf(e, f.r, q);
        // q already carries the sign of the true quotient (also for x = -0, where the correction term would give +0): one v_bfi
        return __builtin_copysignf(q1, q);
    }
    return x / f.d;
}
// The same with the choice made at run time (`ok` is uniform over a level: one scalar branch), for code that has no template to spare.
VORS_HD float div_uniform_rt(float x, const FastDiv& f) {
    return f.ok ? div_uniform<true>(x, f) : x / f.d;
}
struct IntrFast {
    Intr k;
    FastDiv fu, fv;
};
// back_project / warp_jacobian_at with the three divisions by the focal lengths through div_uniform_rt (bit-identical by construction).
VORS_HD V3 back_project_rt(const IntrFast& kf, float px, float py, float depth) {
    const float z = depth;
    const float y = div_uniform_rt((py - kf.k.cv) * z, kf.fv);
    const float x = div_uniform_rt((px - kf.k.cu) * z - kf.k.skew * y, kf.fu);
    return V3{x, y, z};
}
// back_project with the two divisions by fv / fu through div_uniform (bit-identical to back_project by construction).
template <bool FAST>
VORS_HD V3 back_project_fast(const IntrFast& kf, float px, float py, float depth) {
    const float z = depth;
    const float y = div_uniform<FAST>((py - kf.k.cv) * z, kf.fv);
    const float x = div_uniform<FAST>((px - kf.k.cu) * z - kf.k.skew * y, kf.fu);
    return V3{x, y, z};
}
template <bool FAST>
VORS_HD void warp_jacobian_at_fast(float gu, float gv, float u, float v, float _z, const IntrFast& kf, float J[6]) {
    const Intr& k = kf.k;
    const float a = u - k.cu;
    const float b = v - k.cv;
    const float c = a * k.fv - k.skew * b;
    const float _fv = 1.0f / k.fv;
    const float _fuv = 1.0f / (k.fu * k.fv);
    J[0] = gu * _z * k.fu;
    J[1] = _z * (gu * k.skew + gv * k.fv);
    J[2] = -_z * (gu * a + gv * b);
    J[3] = gu * (-a * b * _fv - k.skew) + gv * (-b * b * _fv - k.fv);
    J[4] = gu * (a * c * _fuv + k.fu) + gv * (b * c * _fuv);
    J[5] = gu * (-k.fu * k.fu * b + k.skew * c) * _fuv + gv * div_uniform<FAST>(c, kf.fu);
}
VORS_HD void warp_jacobian_at_rt(float gu, float gv, float u, float v, float _z, const IntrFast& kf, float J[6]) {
    const Intr& k = kf.k;
    const float a = u - k.cu;
    const float b = v - k.cv;
    const float c = a * k.fv - k.skew * b;
    const float _fv = 1.0f / k.fv;
    const float _fuv = 1.0f / (k.fu * k.fv);
    J[0] = gu * _z * k.fu;
    J[1] = _z * (gu * k.skew + gv * k.fv);
    J[2] = -_z * (gu * a + gv * b);
    J[3] = gu * (-a * b * _fv - k.skew) + gv * (-b * b * _fv - k.fv);
    J[4] = gu * (a * c * _fuv + k.fu) + gv * (b * c * _fuv);
    J[5] = gu * (-k.fu * k.fu * b + k.skew * c) * _fuv + gv * div_uniform_rt(c, kf.fu);
}

// so3 / se3 log: API parity only (src/math/so3.rs:81-99, src/math/se3.rs:99-129); host use.
inline void so3_log(const Quat& r, float w[3]) {
    const V3 imag{r.i, r.j, r.k};
    const float imag_norm_2 = dot3(imag, imag);
    const float real_factor = r.w;
    float s;
    if (imag_norm_2 < 1e-2f * 1e-2f) {
        s = 2.0f / real_factor;
    } else if (fabsf(real_factor) < 1e-2f) {
        const float imag_norm = sqrtf(imag_norm_2);
        const float alpha = fabsf(real_factor) / imag_norm;
        const float sign = signbit(real_factor) ? -1.0f : 1.0f;
        s = (sign * (3.14159265358979323846f - 2.0f * alpha)) / imag_norm;
    } else {
        const float imag_norm = sqrtf(imag_norm_2);
        s = (2.0f * atanf(imag_norm / real_factor)) / imag_norm;
    }
    w[0] = s * imag.x; w[1] = s * imag.y; w[2] = s * imag.z;
}
inline Quat so3_exp(const float w[3]) {  // so3.rs:62-77
    const float theta_2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    float real_factor, imag_factor;
    if (theta_2 < 1e-2f * 1e-2f) {
        real_factor = 1.0f - 0.125f * theta_2;
        imag_factor = 0.5f - (1.0f / 48.0f) * theta_2;
    } else {
        const float theta = sqrtf(theta_2);
        const float half_theta = 0.5f * theta;
        real_factor = cosf(half_theta);
        imag_factor = sinf(half_theta) / theta;
    }
    return unit_from_quaternion(Quat{imag_factor * w[0], imag_factor * w[1], imag_factor * w[2], real_factor});
}
inline void se3_log(const Iso& iso, float xi[6]) {
    const V3 imag{iso.q.i, iso.q.j, iso.q.k};
    const float imag_norm_2 = dot3(imag, imag);
    const float real_factor = iso.q.w;
    float w[3], c2;
    if (imag_norm_2 < 1e-2f * 1e-2f) {
        const float s = 2.0f / real_factor;
        w[0] = s * imag.x; w[1] = s * imag.y; w[2] = s * imag.z;
        const float x_2 = imag_norm_2 / (real_factor * real_factor);
        c2 = (1.0f / 12.0f) * (1.0f + (1.0f / 15.0f) * x_2);
    } else {
        const float imag_norm = sqrtf(imag_norm_2);
        float theta;
        if (fabsf(real_factor) < 1e-2f) {
            const float alpha = fabsf(real_factor) / imag_norm;
            const float sign = signbit(real_factor) ? -1.0f : 1.0f;
            theta = sign * (3.14159265358979323846f - 2.0f * alpha);
        } else {
            theta = 2.0f * atanf(imag_norm / real_factor);
        }
        const float theta_2 = theta * theta;
        const float s = theta / imag_norm;
        w[0] = s * imag.x; w[1] = s * imag.y; w[2] = s * imag.z;
        c2 = (1.0f - 0.5f * theta * real_factor / imag_norm) / theta_2;
    }
    const float wx = w[0], wy = w[1], wz = w[2];
    const float w11 = wx * wx, w12 = wx * wy, w13 = wx * wz, w22 = wy * wy, w23 = wy * wz, w33 = wz * wz;
    const float O[3][3] = {{0.0f, -wz, wy}, {wz, 0.0f, -wx}, {-wy, wx, 0.0f}};
    const float O2[3][3] = {{-w22 - w33, w12, w13}, {w12, -w11 - w33, w23}, {w13, w23, -w11 - w22}};
    float V[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) V[r][c] = ((r == c ? 1.0f : 0.0f) + (-0.5f) * O[r][c]) + c2 * O2[r][c];
    const float tx = iso.t.x, ty = iso.t.y, tz = iso.t.z;
    xi[0] = (V[0][0] * tx + V[0][1] * ty) + V[0][2] * tz;
    xi[1] = (V[1][0] * tx + V[1][1] * ty) + V[1][2] * tz;
    xi[2] = (V[2][0] * tx + V[2][1] * ty) + V[2][2] * tz;
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

}  // namespace vors
