// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Single-threaded CPU restatement (C++17, f32, no FMA contraction) of the hot path of
// mpizenberg/visual-odometry-rs ("vors"): pyramidal inverse-compositional direct RGB-D alignment.
// It exists to CHECK the HIP product path and to be timed as the CPU baseline. Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product library
// (visual-odometry-rs_amd/) never includes, links or calls anything in this directory.
//
// PARITY STATUS: "parity unpinned" for the tracker as a whole. The reference is pure Rust and there is
// no Rust toolchain in this image, so the reference binary cannot be run; and the reference's own tests
// pin only so3/se3 identities and three prune_with_thresh examples. Those ARE checked
// (tests/test_oracle_kat.py). Everything else is pinned by line-by-line fidelity to the cited files, by
// ground-truth pose recovery on analytic scenes, and by first principles in float64
// (tests/test_oracle_first_principles.py: se3 exp / log = matrix exponential / logarithm, the warp Jacobian =
// the derivative of the warp, the evaluation sums = their definition, step = the damped normal equations,
// the inverse-depth pyramid = block means) — which checks the mathematics, not the reference's rounding.
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).
// Build with:  g++ -O3 -std=c++17 -ffp-contract=off -fno-fast-math   (Rust never fuses a*b+c).
//
// Third-party arithmetic that is NOT in /root/reference: nalgebra 0.17 (Cargo.toml:20, exact patch
// unpinned, Cargo.lock git-ignored). Its semantics are restated from its published source from memory
// and each such block is marked "nalgebra assumption".
//
// SENSITIVITY VARIANTS (never "the oracle"; probes built as separate libraries by oracle/Makefile):
//   -DVORS_ORACLE_ACC64          energy / gradient / Hessian / optical-flow sums accumulate in f64 (per-point arithmetic stays f32):
//                                how much of a pose is decided by the f32 summation order the GPU cannot reproduce.
//   -DVORS_ORACLE_NALG_VARIANT=m each bit swaps ONE "nalgebra assumption" for the other plausible evaluation order (<= 1 ulp each):
//                                1 quaternion product scalar part as w1 w2 - i1 i2 - j1 j2 - k1 k2 (left to right);
//                                2 3-/4-vector dots folded the other way; 4 q * v as (p + w t) + qv x t; 8 right-looking
//                                (outer-product) Cholesky and solve by multiplication with reciprocal pivots;
//                                16 from_quaternion multiplies by 1/norm.
//                                tests/test_oracle_sensitivity.py reports the pose each choice moves.
#pragma once
#ifndef VORS_ORACLE_NALG_VARIANT
#define VORS_ORACLE_NALG_VARIANT 0
#endif
#include <cmath>
#include <cstddef>
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace vors_oracle {

// src/misc/type_aliases.rs:10
typedef float Float;
#ifdef VORS_ORACLE_ACC64
typedef double Acc;  // sensitivity probe only
#else
typedef float Acc;   // the reference: f32 sums (lm_optimizer.rs:72-100)
#endif

// ---------------------------------------------------------------------------------------------
// Containers and small algebra (nalgebra restatement)
// ---------------------------------------------------------------------------------------------

// nalgebra::DMatrix<T>: column-major, element (row=y, col=x) at x*nrows + y (SURVEY.md §8b layout).
template <class T>
struct DMatrix {
    int nrows = 0, ncols = 0;
    std::vector<T> data;
    DMatrix() {}
    DMatrix(int r, int c, T v) : nrows(r), ncols(c), data((size_t)r * c, v) {}
    // DMatrix::from_row_slice (src/bin/vors_track.rs:142, src/misc/interop.rs:53-56)
    static DMatrix from_row_slice(int r, int c, const T* src) {
        DMatrix m(r, c, T());
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < c; ++j) m(i, j) = src[(size_t)i * c + j];
        return m;
    }
    void to_row_slice(T* dst) const {
        for (int i = 0; i < nrows; ++i)
            for (int j = 0; j < ncols; ++j) dst[(size_t)i * ncols + j] = (*this)(i, j);
    }
    T& operator()(int r, int c) { return data[(size_t)c * nrows + r]; }
    const T& operator()(int r, int c) const { return data[(size_t)c * nrows + r]; }
};

struct Vec3 {
    Float x, y, z;
};
// nalgebra Quaternion coords order [i, j, k, w].
struct Quat {
    Float i, j, k, w;
};
// Isometry3<f32> = (translation, unit quaternion).
struct Iso3 {
    Vec3 t;
    Quat q;
};
struct Vec6 {
    Float v[6];
};
struct Mat6 {  // m[r][c]
    Float m[6][6];
};

inline Iso3 iso_identity() { return Iso3{{0, 0, 0}, {0, 0, 0, 1}}; }

// nalgebra assumption: Vector3::cross.
inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// nalgebra assumption: static U3 dot special case = (a + b) + c.
inline Float dot3(const Vec3& a, const Vec3& b) {
    if (VORS_ORACLE_NALG_VARIANT & 2) return a.x * b.x + (a.y * b.y + a.z * b.z);
    return (a.x * b.x + a.y * b.y) + a.z * b.z;
}
// nalgebra assumption: static U4 dot special case: a=x0*y0, b=x1*y1, c=x2*y2, d=x3*y3; (a+c)+(b+d).
inline Float quat_norm_squared(const Quat& q) {
    Float a = q.i * q.i, b = q.j * q.j, c = q.k * q.k, d = q.w * q.w;
    if (VORS_ORACLE_NALG_VARIANT & 2) return ((a + b) + c) + d;
    a += c;
    b += d;
    return a + b;
}
// nalgebra assumption: UnitQuaternion * Vector3:  t = cross(qv, p) * 2; (t * w + cross(qv, t)) + p.
inline Vec3 quat_rotate(const Quat& q, const Vec3& p) {
    const Vec3 qv{q.i, q.j, q.k};
    Vec3 t = cross(qv, p);
    t = Vec3{t.x * 2.0f, t.y * 2.0f, t.z * 2.0f};
    const Vec3 c = cross(qv, t);
    if (VORS_ORACLE_NALG_VARIANT & 4) return Vec3{(p.x + q.w * t.x) + c.x, (p.y + q.w * t.y) + c.y, (p.z + q.w * t.z) + c.z};
    return Vec3{(t.x * q.w + c.x) + p.x, (t.y * q.w + c.y) + p.y, (t.z * q.w + c.z) + p.z};
}
// nalgebra assumption: Quaternion * Quaternion (Hamilton), scalar part uses vector dot.
inline Quat quat_mul(const Quat& a, const Quat& b) {
    Quat r;
    r.w = a.w * b.w - dot3(Vec3{a.i, a.j, a.k}, Vec3{b.i, b.j, b.k});
    if (VORS_ORACLE_NALG_VARIANT & 1) r.w = a.w * b.w - a.i * b.i - a.j * b.j - a.k * b.k;
    r.i = a.w * b.i + a.i * b.w + a.j * b.k - a.k * b.j;
    r.j = a.w * b.j - a.i * b.k + a.j * b.w + a.k * b.i;
    r.k = a.w * b.k + a.i * b.j - a.j * b.i + a.k * b.w;
    return r;
}
inline Quat quat_conj(const Quat& q) { return Quat{-q.i, -q.j, -q.k, q.w}; }
// nalgebra assumption: UnitQuaternion::from_quaternion = q / sqrt(norm_squared), per-coordinate division.
inline Quat unit_from_quaternion(const Quat& q) {
    const Float n = std::sqrt(quat_norm_squared(q));
    if (VORS_ORACLE_NALG_VARIANT & 16) {
        const Float r = 1.0f / n;
        return Quat{q.i * r, q.j * r, q.k * r, q.w * r};
    }
    return Quat{q.i / n, q.j / n, q.k / n, q.w / n};
}
// nalgebra assumption: Isometry * Point = rotation*p + translation.
inline Vec3 iso_transform_point(const Iso3& m, const Vec3& p) {
    const Vec3 r = quat_rotate(m.q, p);
    return Vec3{r.x + m.t.x, r.y + m.t.y, r.z + m.t.z};
}
// nalgebra assumption: Isometry * Isometry = (t_A + R_A t_B, q_A q_B), no renormalisation.
inline Iso3 iso_mul(const Iso3& a, const Iso3& b) {
    const Vec3 shift = quat_rotate(a.q, b.t);
    return Iso3{Vec3{a.t.x + shift.x, a.t.y + shift.y, a.t.z + shift.z}, quat_mul(a.q, b.q)};
}
// nalgebra assumption: Isometry::inverse = (R^-1 (-t), q*).
inline Iso3 iso_inverse(const Iso3& a) {
    const Quat qi = quat_conj(a.q);
    const Vec3 nt{-a.t.x, -a.t.y, -a.t.z};
    return Iso3{quat_rotate(qi, nt), qi};
}

// nalgebra assumption: Cholesky::new (left-looking, lower triangle, fail when a pivot is not > 0) and
// Cholesky::solve (forward substitution by columns with axpy, backward substitution with a sequential dot).
// Returns false where nalgebra returns None (lm_optimizer.rs:131-133).
inline bool cholesky6(Mat6& a) {
    if (VORS_ORACLE_NALG_VARIANT & 8) {  // right-looking: scale the column, then the rank-1 update of the trailing block
        for (int j = 0; j < 6; ++j) {
            const Float diag = a.m[j][j];
            if (!(diag > 0.0f)) return false;
            const Float denom = std::sqrt(diag), inv = 1.0f / denom;
            a.m[j][j] = denom;
            for (int i = j + 1; i < 6; ++i) a.m[i][j] *= inv;
            for (int c = j + 1; c < 6; ++c)
                for (int i = c; i < 6; ++i) a.m[i][c] -= a.m[i][j] * a.m[c][j];
        }
        return true;
    }
    for (int j = 0; j < 6; ++j) {
        for (int k = 0; k < j; ++k) {
            const Float factor = -a.m[j][k];
            for (int i = j; i < 6; ++i) a.m[i][j] = factor * a.m[i][k] + a.m[i][j];
        }
        const Float diag = a.m[j][j];
        if (diag > 0.0f) {
            const Float denom = std::sqrt(diag);
            a.m[j][j] = denom;
            for (int i = j + 1; i < 6; ++i) a.m[i][j] /= denom;
            continue;
        }
        return false;
    }
    return true;
}
inline Vec6 cholesky6_solve(const Mat6& l, const Vec6& rhs) {
    Vec6 b = rhs;
    if (VORS_ORACLE_NALG_VARIANT & 8) {
        for (int i = 0; i < 6; ++i) {
            Float acc = b.v[i];
            for (int k = 0; k < i; ++k) acc -= l.m[i][k] * b.v[k];
            b.v[i] = acc * (1.0f / l.m[i][i]);
        }
        for (int i = 5; i >= 0; --i) {
            Float acc = b.v[i];
            for (int k = 5; k > i; --k) acc -= l.m[k][i] * b.v[k];
            b.v[i] = acc * (1.0f / l.m[i][i]);
        }
        return b;
    }
    for (int i = 0; i < 6; ++i) {
        const Float coeff = b.v[i] / l.m[i][i];
        b.v[i] = coeff;
        for (int k = i + 1; k < 6; ++k) b.v[k] = (-coeff) * l.m[k][i] + b.v[k];
    }
    for (int i = 5; i >= 0; --i) {
        Float dot = 0.0f;
        for (int k = i + 1; k < 6; ++k) dot += l.m[k][i] * b.v[k];
        b.v[i] = (b.v[i] - dot) / l.m[i][i];
    }
    return b;
}

// ---------------------------------------------------------------------------------------------
// src/math/so3.rs and src/math/se3.rs
// ---------------------------------------------------------------------------------------------
struct Mat3 {
    Float m[3][3];
};
struct Mat4 {
    Float m[4][4];
};

namespace so3 {
// so3.rs:19-22
constexpr Float EPSILON_TAYLOR_SERIES = 1e-2f;
constexpr Float EPSILON_TAYLOR_SERIES_2 = EPSILON_TAYLOR_SERIES * EPSILON_TAYLOR_SERIES;
constexpr Float _1_8 = 0.125f;
constexpr Float _1_48 = 1.0f / 48.0f;
constexpr Float PI = 3.14159265358979323846f;  // std::f32::consts::PI

// so3.rs:27-33
inline Mat3 hat(const Vec3& w) { return Mat3{{{0.0f, -w.z, w.y}, {w.z, 0.0f, -w.x}, {-w.y, w.x, 0.0f}}}; }
// so3.rs:38-51
inline Mat3 hat_2(const Vec3& w) {
    const Float w11 = w.x * w.x, w12 = w.x * w.y, w13 = w.x * w.z;
    const Float w22 = w.y * w.y, w23 = w.y * w.z, w33 = w.z * w.z;
    return Mat3{{{-w22 - w33, w12, w13}, {w12, -w11 - w33, w23}, {w13, w23, -w11 - w22}}};
}
// so3.rs:55-57  (m32, m13, m21)
inline Vec3 vee(const Mat3& a) { return Vec3{a.m[2][1], a.m[0][2], a.m[1][0]}; }
// so3.rs:62-77
inline Quat exp(const Vec3& w) {
    const Float theta_2 = dot3(w, w);
    Float real_factor, imag_factor;
    if (theta_2 < EPSILON_TAYLOR_SERIES_2) {
        real_factor = 1.0f - _1_8 * theta_2;
        imag_factor = 0.5f - _1_48 * theta_2;
    } else {
        const Float theta = std::sqrt(theta_2);
        const Float half_theta = 0.5f * theta;
        real_factor = std::cos(half_theta);
        imag_factor = std::sin(half_theta) / theta;
    }
    return unit_from_quaternion(Quat{imag_factor * w.x, imag_factor * w.y, imag_factor * w.z, real_factor});
}
// so3.rs:81-99
inline Vec3 log(const Quat& rotation) {
    const Vec3 imag{rotation.i, rotation.j, rotation.k};
    const Float imag_norm_2 = dot3(imag, imag);
    const Float real_factor = rotation.w;
    if (imag_norm_2 < EPSILON_TAYLOR_SERIES_2) {
        const Float s = 2.0f / real_factor;
        return Vec3{s * imag.x, s * imag.y, s * imag.z};
    } else if (std::fabs(real_factor) < EPSILON_TAYLOR_SERIES) {
        const Float imag_norm = std::sqrt(imag_norm_2);
        const Float alpha = std::fabs(real_factor) / imag_norm;
        const Float sign = std::signbit(real_factor) ? -1.0f : 1.0f;  // f32::signum
        const Float theta = sign * (PI - 2.0f * alpha);
        const Float s = theta / imag_norm;
        return Vec3{s * imag.x, s * imag.y, s * imag.z};
    } else {
        const Float imag_norm = std::sqrt(imag_norm_2);
        const Float theta = 2.0f * std::atan(imag_norm / real_factor);
        const Float s = theta / imag_norm;
        return Vec3{s * imag.x, s * imag.y, s * imag.z};
    }
}
}  // namespace so3

namespace se3 {
// se3.rs:19-27
constexpr Float EPSILON_TAYLOR_SERIES = 1e-2f;
constexpr Float EPSILON_TAYLOR_SERIES_2 = EPSILON_TAYLOR_SERIES * EPSILON_TAYLOR_SERIES;
constexpr Float _1_6 = 1.0f / 6.0f;
constexpr Float _1_8 = 0.125f;
constexpr Float _1_12 = 1.0f / 12.0f;
constexpr Float _1_15 = 1.0f / 15.0f;
constexpr Float _1_24 = 1.0f / 24.0f;
constexpr Float _1_48 = 1.0f / 48.0f;
constexpr Float _1_120 = 1.0f / 120.0f;

// se3.rs:45-56
inline Mat4 hat(const Vec6& xi) {
    const Float w1 = xi.v[3], w2 = xi.v[4], w3 = xi.v[5];
    return Mat4{{{0.0f, -w3, w2, xi.v[0]}, {w3, 0.0f, -w1, xi.v[1]}, {-w2, w1, 0.0f, xi.v[2]}, {0.0f, 0.0f, 0.0f, 0.0f}}};
}
// se3.rs:60-62  (m14, m24, m34, m32, m13, m21)
inline Vec6 vee(const Mat4& a) { return Vec6{{a.m[0][3], a.m[1][3], a.m[2][3], a.m[2][1], a.m[0][2], a.m[1][0]}}; }

// V = I + c1*omega + c2*omega_2 evaluated elementwise as (I + c1*O) + c2*O2; then V*v via nalgebra gemv
// (column axpy: ((V_i0 v0) + V_i1 v1) + V_i2 v2).   nalgebra assumption.
inline Vec3 v_times(Float c1, const Mat3& o, Float c2, const Mat3& o2, const Vec3& v) {
    Float V[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) V[r][c] = ((r == c ? 1.0f : 0.0f) + c1 * o.m[r][c]) + c2 * o2.m[r][c];
    Vec3 out;
    out.x = (V[0][0] * v.x + V[0][1] * v.y) + V[0][2] * v.z;
    out.y = (V[1][0] * v.x + V[1][1] * v.y) + V[1][2] * v.z;
    out.z = (V[2][0] * v.x + V[2][1] * v.y) + V[2][2] * v.z;
    return out;
}

// se3.rs:65-95
inline Iso3 exp(const Vec6& xi) {
    const Vec3 xi_v{xi.v[0], xi.v[1], xi.v[2]};
    const Vec3 xi_w{xi.v[3], xi.v[4], xi.v[5]};
    const Float theta_2 = dot3(xi_w, xi_w);
    const Mat3 omega = so3::hat(xi_w), omega_2 = so3::hat_2(xi_w);
    Float real_factor, imag_factor, coef_omega, coef_omega_2;
    if (theta_2 < EPSILON_TAYLOR_SERIES_2) {
        real_factor = 1.0f - _1_8 * theta_2;
        imag_factor = 0.5f - _1_48 * theta_2;
        coef_omega = 0.5f - _1_24 * theta_2;
        coef_omega_2 = _1_6 - _1_120 * theta_2;
    } else {
        const Float theta = std::sqrt(theta_2);
        const Float half_theta = 0.5f * theta;
        real_factor = std::cos(half_theta);
        imag_factor = std::sin(half_theta) / theta;
        coef_omega = (1.0f - std::cos(theta)) / theta_2;
        coef_omega_2 = (theta - std::sin(theta)) / (theta * theta_2);
    }
    const Vec3 t = v_times(coef_omega, omega, coef_omega_2, omega_2, xi_v);
    const Quat rotation =
        unit_from_quaternion(Quat{imag_factor * xi_w.x, imag_factor * xi_w.y, imag_factor * xi_w.z, real_factor});
    return Iso3{t, rotation};
}

// se3.rs:99-129
inline Vec6 log(const Iso3& iso) {
    const Vec3 imag{iso.q.i, iso.q.j, iso.q.k};
    const Float imag_norm_2 = dot3(imag, imag);
    const Float real_factor = iso.q.w;
    if (imag_norm_2 < EPSILON_TAYLOR_SERIES_2) {
        const Float theta_by_imag_norm = 2.0f / real_factor;
        const Vec3 w{theta_by_imag_norm * imag.x, theta_by_imag_norm * imag.y, theta_by_imag_norm * imag.z};
        const Mat3 omega = so3::hat(w), omega_2 = so3::hat_2(w);
        const Float x_2 = imag_norm_2 / (real_factor * real_factor);
        const Float coef_omega_2 = _1_12 * (1.0f + _1_15 * x_2);
        const Vec3 xi_v = v_times(-0.5f, omega, coef_omega_2, omega_2, iso.t);  // I - 0.5*omega + c*omega_2
        return Vec6{{xi_v.x, xi_v.y, xi_v.z, w.x, w.y, w.z}};
    } else {
        const Float imag_norm = std::sqrt(imag_norm_2);
        Float theta;
        if (std::fabs(real_factor) < EPSILON_TAYLOR_SERIES) {
            const Float alpha = std::fabs(real_factor) / imag_norm;
            const Float sign = std::signbit(real_factor) ? -1.0f : 1.0f;
            theta = sign * (so3::PI - 2.0f * alpha);
        } else {
            theta = 2.0f * std::atan(imag_norm / real_factor);
        }
        const Float theta_2 = theta * theta;
        const Float s = theta / imag_norm;
        const Vec3 w{s * imag.x, s * imag.y, s * imag.z};
        const Mat3 omega = so3::hat(w), omega_2 = so3::hat_2(w);
        const Float coef_omega_2 = (1.0f - 0.5f * theta * real_factor / imag_norm) / theta_2;
        const Vec3 xi_v = v_times(-0.5f, omega, coef_omega_2, omega_2, iso.t);
        return Vec6{{xi_v.x, xi_v.y, xi_v.z, w.x, w.y, w.z}};
    }
}
}  // namespace se3

// ---------------------------------------------------------------------------------------------
// src/core/camera.rs:84-140
// ---------------------------------------------------------------------------------------------
struct Intrinsics {
    Float cu, cv;  // principal_point
    Float fu, fv;  // focal
    Float skew;
    // camera.rs:115-123 (skew intentionally left unscaled, as in the reference)
    Intrinsics half_res() const {
        return Intrinsics{(cu + 0.5f) / 2.0f - 0.5f, (cv + 0.5f) / 2.0f - 0.5f, 0.5f * fu, 0.5f * fv, skew};
    }
    // camera.rs:106-108 via multires::limited_sequence: always n levels (n>=1).
    std::vector<Intrinsics> multi_res(size_t n) const {
        std::vector<Intrinsics> s{*this};
        size_t length = 1;
        while (length < n) {
            ++length;
            s.push_back(s.back().half_res());
        }
        return s;
    }
    // camera.rs:126-132
    Vec3 project(const Vec3& p) const { return Vec3{fu * p.x + skew * p.y + cu * p.z, fv * p.y + cv * p.z, p.z}; }
    // camera.rs:135-140
    Vec3 back_project(Float px, Float py, Float depth) const {
        const Float z = depth;
        const Float y = (py - cv) * z / fv;
        const Float x = ((px - cu) * z - skew * y) / fu;
        return Vec3{x, y, z};
    }
};

// ---------------------------------------------------------------------------------------------
// src/core/multires.rs
// ---------------------------------------------------------------------------------------------
// multires.rs:67-88
template <class T, class U, class F>
bool halve(const DMatrix<T>& mat, F f, DMatrix<U>& out) {
    const int half_r = mat.nrows / 2, half_c = mat.ncols / 2;
    if (half_r == 0 || half_c == 0) return false;
    out = DMatrix<U>(half_r, half_c, U());
    for (int j = 0; j < half_c; ++j)  // from_fn fills column-major
        for (int i = 0; i < half_r; ++i) {
            const T a = mat(2 * i, 2 * j), b = mat(2 * i + 1, 2 * j);
            const T c = mat(2 * i, 2 * j + 1), d = mat(2 * i + 1, 2 * j + 1);
            out(i, j) = f(a, b, c, d);
        }
    return true;
}
// multires.rs:38-60: apply f while it succeeds and the length is < max_length (0 behaves like 1).
template <class T, class F>
std::vector<T> limited_sequence(size_t max_length, T data, F f) {
    std::vector<T> s;
    s.push_back(std::move(data));
    size_t length = 1;
    while (length < max_length) {
        ++length;
        T next;
        if (!f(s.back(), next)) break;
        s.push_back(std::move(next));
    }
    return s;
}
// multires.rs:21-31
inline std::vector<DMatrix<uint8_t>> mean_pyramid(size_t max_levels, DMatrix<uint8_t> mat) {
    auto mean = [](uint8_t a, uint8_t b, uint8_t c, uint8_t d) -> uint8_t {
        const uint16_t s = (uint16_t)((uint16_t)a + (uint16_t)b + (uint16_t)c + (uint16_t)d);
        return (uint8_t)(s / 4);
    };
    return limited_sequence(max_levels, std::move(mat), [&](const DMatrix<uint8_t>& m, DMatrix<uint8_t>& o) {
        return halve<uint8_t, uint8_t>(m, mean, o);
    });
}

// ---------------------------------------------------------------------------------------------
// src/core/gradient.rs
// ---------------------------------------------------------------------------------------------
namespace gradient {
// gradient.rs:15-33 — i16 `/ 2` truncates toward zero, as C++ does.
inline void centered(const DMatrix<uint8_t>& img, DMatrix<int16_t>& gx, DMatrix<int16_t>& gy) {
    const int nr = img.nrows, nc = img.ncols;
    gx = DMatrix<int16_t>(nr, nc, 0);
    gy = DMatrix<int16_t>(nr, nc, 0);
    for (int j = 0; j < nc - 2; ++j)
        for (int i = 0; i < nr - 2; ++i) {
            // top = img[(i, j+1)], bottom = img[(i+2, j+1)], left = img[(i+1, j)], right = img[(i+1, j+2)]
            gx(i + 1, j + 1) = (int16_t)(((int16_t)img(i + 1, j + 2) - (int16_t)img(i + 1, j)) / 2);
            gy(i + 1, j + 1) = (int16_t)(((int16_t)img(i + 2, j + 1) - (int16_t)img(i, j + 1)) / 2);
        }
}
// gradient.rs:38-44 — computed in i32 then `as u16` (wraps mod 65536).
inline DMatrix<uint16_t> squared_norm(const DMatrix<int16_t>& gx, const DMatrix<int16_t>& gy) {
    DMatrix<uint16_t> out(gx.nrows, gx.ncols, 0);
    for (size_t k = 0; k < out.data.size(); ++k) {
        const int32_t x = gx.data[k], y = gy.data[k];
        out.data[k] = (uint16_t)(uint32_t)(x * x + y * y);
    }
    return out;
}
// gradient.rs:74-80 — block layout a c / b d
inline int16_t bloc_x(uint8_t a, uint8_t b, uint8_t c, uint8_t d) {
    return (int16_t)(((int16_t)c + (int16_t)d - (int16_t)a - (int16_t)b) / 2);
}
// gradient.rs:87-93
inline int16_t bloc_y(uint8_t a, uint8_t b, uint8_t c, uint8_t d) {
    return (int16_t)(((int16_t)b - (int16_t)a + (int16_t)d - (int16_t)c) / 2);
}
}  // namespace gradient

// multires.rs:112-126 — one entry per level 1..n-1, computed from the next finer image.
inline std::vector<std::pair<DMatrix<int16_t>, DMatrix<int16_t>>> gradients_xy(
    const std::vector<DMatrix<uint8_t>>& multires_mat) {
    std::vector<std::pair<DMatrix<int16_t>, DMatrix<int16_t>>> out;
    for (size_t l = 0; l + 1 < multires_mat.size(); ++l) {
        DMatrix<int16_t> gx, gy;
        halve<uint8_t, int16_t>(multires_mat[l], gradient::bloc_x, gx);
        halve<uint8_t, int16_t>(multires_mat[l], gradient::bloc_y, gy);
        out.emplace_back(std::move(gx), std::move(gy));
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// src/core/candidates/coarse_to_fine.rs
// ---------------------------------------------------------------------------------------------
namespace candidates {
// coarse_to_fine.rs:73-89.  sort_unstable_by on 4 elements = insertion sort (stable) in the 2019-era
// Rust std (assumption): ties keep original order, so the later element ranks higher.
// `y + thresh` is a u16 add that wraps in --release builds.
inline void prune_with_thresh(uint16_t thresh, uint16_t a, uint16_t b, uint16_t c, uint16_t d, bool result[4]) {
    std::pair<uint16_t, int> temp[4] = {{a, 0}, {b, 1}, {c, 2}, {d, 3}};
    for (int i = 1; i < 4; ++i) {  // insertion sort, shift while strictly less
        std::pair<uint16_t, int> x = temp[i];
        int k = i;
        while (k > 0 && x.first < temp[k - 1].first) {
            temp[k] = temp[k - 1];
            --k;
        }
        temp[k] = x;
    }
    const int first = temp[3].second;
    const uint16_t x = temp[2].first;
    const int second = temp[2].second;
    const uint16_t y = temp[1].first;
    result[0] = result[1] = result[2] = result[3] = false;
    result[first] = true;
    if (x > (uint16_t)(y + thresh)) result[second] = true;
}
// coarse_to_fine.rs:37-62
inline DMatrix<uint8_t> select_2x2_bloc(const DMatrix<uint8_t>& pre_mask, const DMatrix<uint16_t>& mat, uint16_t thresh) {
    DMatrix<uint8_t> mask(mat.nrows, mat.ncols, 0);
    for (int j = 0; j < pre_mask.ncols; ++j)
        for (int i = 0; i < pre_mask.nrows; ++i)
            if (pre_mask(i, j)) {
                bool ok[4];
                prune_with_thresh(thresh, mat(2 * i, 2 * j), mat(2 * i + 1, 2 * j), mat(2 * i, 2 * j + 1),
                                  mat(2 * i + 1, 2 * j + 1), ok);
                mask(2 * i, 2 * j) = ok[0];
                mask(2 * i + 1, 2 * j) = ok[1];
                mask(2 * i, 2 * j + 1) = ok[2];
                mask(2 * i + 1, 2 * j + 1) = ok[3];
            }
    return mask;
}
// coarse_to_fine.rs:15-32 — gradients[0] = finest; returns masks coarsest -> finest.
inline std::vector<DMatrix<uint8_t>> select(uint16_t diff_threshold, const std::vector<DMatrix<uint16_t>>& gradients) {
    std::vector<DMatrix<uint8_t>> masks;
    masks.emplace_back(gradients.back().nrows, gradients.back().ncols, (uint8_t)1);
    for (int l = (int)gradients.size() - 2; l >= 0; --l) masks.push_back(select_2x2_bloc(masks.back(), gradients[l], diff_threshold));
    return masks;
}
}  // namespace candidates

// ---------------------------------------------------------------------------------------------
// src/core/candidates/dso.rs — DSO-style candidate selection. NOT used by the reference's Tracker (which hard-wires
// coarse_to_fine, inverse_compositional.rs:16,120-125); only by examples/candidates_dso.rs:40-59, whose parameters are
// used when it serves as an alternative level-0 mask source here (candidates_mode = 2, BASELINE config 3).
// Deviation: the reference's random sub-sampling branch draws from an UNSEEDED rand::thread_rng() (dso.rs:140-143), which
// nothing can reproduce; it is replaced by a counter-based hash of (seed, row, col) so that results are repeatable.
// ---------------------------------------------------------------------------------------------
namespace gradient {
// gradient.rs:49-65: ((gx^2 + gy^2) / 4) as u16 with un-halved centred differences; 1-px border = 0.
inline DMatrix<uint16_t> squared_norm_direct(const DMatrix<uint8_t>& im) {
    const int nr = im.nrows, nc = im.ncols;
    DMatrix<uint16_t> out(nr, nc, 0);
    for (int j = 0; j < nc - 2; ++j)
        for (int i = 0; i < nr - 2; ++i) {
            const int32_t gx = (int32_t)im(i + 1, j + 2) - (int32_t)im(i + 1, j);
            const int32_t gy = (int32_t)im(i + 2, j + 1) - (int32_t)im(i, j + 1);
            out(i + 1, j + 1) = (uint16_t)((gx * gx + gy * gy) / 4);
        }
    return out;
}
}  // namespace gradient

namespace dso {
struct RegionConfig {  // dso.rs:38-43
    size_t size;
    Float coef_a;
    uint16_t coef_b;
};
struct BlockConfig {  // dso.rs:46-54
    size_t base_size, nb_levels;
    Float threshold_factor;
};
struct RecursiveConfig {  // dso.rs:59-70
    size_t nb_iterations_left;
    Float low_thresh, high_thresh, random_thresh;
};
struct MaxGrad {
    uint16_t g;
    size_t i, j;
};
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// dso.rs:307-325
inline DMatrix<uint16_t> region_median_gradients(const DMatrix<uint16_t>& g, size_t size) {
    const size_t nr = g.nrows, nc = g.ncols;
    const size_t rr = nr / size + (nr % size ? 1 : 0), rc = nc / size + (nc % size ? 1 : 0);
    DMatrix<uint16_t> out((int)rr, (int)rc, 0);
    std::vector<uint16_t> tmp;
    for (size_t j = 0; j < rc; ++j)
        for (size_t i = 0; i < rr; ++i) {
            const size_t h = std::min(size, nr - i * size), w = std::min(size, nc - j * size);
            tmp.clear();
            for (size_t c = 0; c < w; ++c)
                for (size_t r = 0; r < h; ++r) tmp.push_back(g((int)(i * size + r), (int)(j * size + c)));
            std::sort(tmp.begin(), tmp.end());
            out((int)i, (int)j) = tmp[tmp.size() / 2];
        }
    return out;
}
// dso.rs:284-303: threshold = a * (mean3x3(median) + b)^2, cast to u16 (truncation). The reference's num_traits::cast(..)
// .expect("woops") panics when the value does not fit u16; with the input of examples/candidates_dso.rs:42 (gradient magnitudes
// sqrt(g2 / 4) <= 180, a = 1, b = 3) the value is at most (180 + 3)^2 = 33,489, so the panic is unreachable on this path and the
// plain cast below never wraps (asserted).
inline DMatrix<uint16_t> region_thresholds(const DMatrix<uint16_t>& med, Float a, uint16_t b) {
    const int nr = med.nrows, nc = med.ncols;
    DMatrix<uint16_t> out(nr, nc, 0);
    for (int j = 0; j < nc; ++j)
        for (int i = 0; i < nr; ++i) {
            const int si = std::max(0, i - 1), sj = std::max(0, j - 1), ei = std::min(nr, i + 2), ej = std::min(nc, j + 2);
            uint16_t sum = 0;
            int n = 0;
            for (int jj = sj; jj < ej; ++jj)
                for (int ii = si; ii < ei; ++ii) {
                    sum = (uint16_t)(sum + med(ii, jj));
                    ++n;
                }
            const Float t = (Float)sum / (Float)n + (Float)b;
            if (!(a * t * t < 65536.0f)) throw std::runtime_error("woops");  // the reference panics here (dso.rs:300)
            out(i, j) = (uint16_t)(a * t * t);
        }
    return out;
}
// dso.rs:192-222: first maximum in column-major order inside each block
inline DMatrix<MaxGrad> init_max_gradients(const DMatrix<uint16_t>& g, size_t bs) {
    const size_t nr = g.nrows, nc = g.ncols;
    const size_t br = nr / bs + (nr % bs ? 1 : 0), bc = nc / bs + (nc % bs ? 1 : 0);
    DMatrix<MaxGrad> out((int)br, (int)bc, MaxGrad{0, 0, 0});
    for (size_t bj = 0; bj < bc; ++bj)
        for (size_t bi = 0; bi < br; ++bi) {
            const size_t si = bi * bs, sj = bj * bs, ei = std::min(si + bs, nr), ej = std::min(sj + bs, nc);
            MaxGrad m{g((int)si, (int)sj), si, sj};
            for (size_t j = sj; j < ej; ++j)
                for (size_t i = si; i < ei; ++i)
                    if (g((int)i, (int)j) > m.g) m = MaxGrad{g((int)i, (int)j), i, j};
            out((int)bi, (int)bj) = m;
        }
    return out;
}
// dso.rs:225-241: g_max(g1, g_max(g2, g_max(g3, g4))), ties keep the first argument
inline MaxGrad max_of_four(const MaxGrad& g1, const MaxGrad& g2, const MaxGrad& g3, const MaxGrad& g4) {
    auto gmax = [](const MaxGrad& a, const MaxGrad& b) { return a.g < b.g ? b : a; };
    return gmax(g1, gmax(g2, gmax(g3, g4)));
}
// dso.rs:156-189 + 248-276. Returns the per-level pick counts; `picked` = 0 or the level (1-based) that picked the pixel.
inline std::vector<size_t> pick_all_block_candidates(const BlockConfig& bc, size_t regions_size, const DMatrix<uint16_t>& thr,
                                                     const DMatrix<uint16_t>& g, DMatrix<uint8_t>& picked) {
    auto multires = limited_sequence(bc.nb_levels, init_max_gradients(g, bc.base_size), [](const DMatrix<MaxGrad>& m, DMatrix<MaxGrad>& o) {
        return halve<MaxGrad, MaxGrad>(m, max_of_four, o);
    });
    Float coef = 1.0f;
    std::vector<size_t> nb_picked;
    DMatrix<uint8_t> mask(multires[0].nrows, multires[0].ncols, (uint8_t)1);
    picked = DMatrix<uint8_t>(g.nrows, g.ncols, (uint8_t)0);
    for (size_t level = 0; level < multires.size(); ++level) {
        const DMatrix<MaxGrad>& mg = multires[level];
        const int mh = mask.nrows, mw = mask.ncols;
        DMatrix<uint8_t> next(mh / 2, mw / 2, (uint8_t)1);
        size_t n = 0;
        for (int j = 0; j < mw / 2 * 2; ++j)
            for (int i = 0; i < mh / 2 * 2; ++i) {
                if (mask(i, j)) {
                    const MaxGrad& m = mg(i, j);
                    const uint16_t threshold = thr((int)(m.i / regions_size), (int)(m.j / regions_size));
                    if ((Float)m.g >= coef * (Float)threshold) {
                        next(i / 2, j / 2) = 0;
                        picked((int)m.i, (int)m.j) = (uint8_t)(level + 1);
                        ++n;
                    }
                } else {
                    next(i / 2, j / 2) = 0;
                }
            }
        nb_picked.push_back(n);
        mask = std::move(next);
        coef *= bc.threshold_factor;
    }
    return nb_picked;
}
// dso.rs:98-147
inline DMatrix<uint8_t> select(const DMatrix<uint16_t>& g, const RegionConfig& rc, BlockConfig bc, RecursiveConfig rec, size_t nb_target,
                               uint64_t seed, std::vector<size_t>* trace_base_sizes = nullptr) {
    const DMatrix<uint16_t> med = region_median_gradients(g, rc.size);
    const DMatrix<uint16_t> thr = region_thresholds(med, rc.coef_a, rc.coef_b);
    for (;;) {
        if (trace_base_sizes) trace_base_sizes->push_back(bc.base_size);
        DMatrix<uint8_t> picked;
        const std::vector<size_t> counts = pick_all_block_candidates(bc, rc.size, thr, g, picked);
        size_t nb_candidates = 0;
        for (size_t c : counts) nb_candidates += c;
        const Float ratio = (Float)nb_candidates / (Float)nb_target;
        const Float ts = std::sqrt(ratio) * ((Float)bc.base_size + 1.0f) - 1.0f;
        const size_t target_size = (size_t)std::max(1, (int)std::round(ts));
        DMatrix<uint8_t> mask(picked.nrows, picked.ncols, (uint8_t)0);
        if (ratio < rec.low_thresh || ratio > rec.high_thresh) {
            if (target_size != bc.base_size && rec.nb_iterations_left > 0) {
                bc.base_size = target_size;
                rec.nb_iterations_left -= 1;
                continue;  // the medians / thresholds of the recursive call are identical
            }
            for (size_t k = 0; k < mask.data.size(); ++k) mask.data[k] = picked.data[k] > 0;
        } else if (ratio > rec.random_thresh) {
            const uint8_t keep = (uint8_t)(255.0f / ratio);
            for (int j = 0; j < mask.ncols; ++j)
                for (int i = 0; i < mask.nrows; ++i) {
                    const uint8_t r = (uint8_t)(splitmix64(seed ^ splitmix64(((uint64_t)(uint32_t)i << 32) | (uint32_t)j)) & 0xff);
                    mask(i, j) = picked(i, j) > 0 && r <= keep;  // deviation: counter-based stand-in for rng.gen::<u8>()
                }
        } else {
            for (size_t k = 0; k < mask.data.size(); ++k) mask.data[k] = picked.data[k] > 0;
        }
        return mask;
    }
}
// examples/candidates_dso.rs:40-59: gradient magnitude = sqrt(squared_norm_direct) as u16; defaults with 2 recursion rounds.
inline DMatrix<uint8_t> select_like_example(const DMatrix<uint8_t>& img, size_t nb_target, uint64_t seed, std::vector<size_t>* trace = nullptr) {
    DMatrix<uint16_t> g = gradient::squared_norm_direct(img);
    for (auto& v : g.data) v = (uint16_t)std::sqrt((Float)v);
    return select(g, RegionConfig{32, 1.0f, 3}, BlockConfig{4, 3, 0.5f}, RecursiveConfig{2, 0.8f, 4.0f, 1.1f}, nb_target, seed, trace);
}
}  // namespace dso

// ---------------------------------------------------------------------------------------------
// src/core/inverse_depth.rs
// ---------------------------------------------------------------------------------------------
struct InverseDepth {  // inverse_depth.rs:12-19
    enum Kind : uint32_t { Unknown = 0, Discarded = 1, WithVariance = 2 } kind;
    Float d, v;
};
namespace inverse_depth {
// inverse_depth.rs:24-29
inline InverseDepth from_depth(Float scale, uint16_t depth, Float variance) {
    if (depth == 0) return InverseDepth{InverseDepth::Unknown, 0, 0};
    return InverseDepth{InverseDepth::WithVariance, scale / (Float)depth, variance};
}
// inverse_depth.rs:49-66 + 81-98: keep known values in order [a,b,c,d], weighted mean, weights add.
inline InverseDepth fuse_dso_mean(const InverseDepth& a, const InverseDepth& b, const InverseDepth& c, const InverseDepth& d) {
    Float dv[4], vv[4];
    int n = 0;
    const InverseDepth* in[4] = {&a, &b, &c, &d};
    for (int k = 0; k < 4; ++k)
        if (in[k]->kind == InverseDepth::WithVariance) {
            dv[n] = in[k]->d;
            vv[n] = in[k]->v;
            ++n;
        }
    switch (n) {
        case 1: return InverseDepth{InverseDepth::WithVariance, dv[0], vv[0]};
        case 2: {
            const Float sum = vv[0] + vv[1];
            return InverseDepth{InverseDepth::WithVariance, (dv[0] * vv[0] + dv[1] * vv[1]) / sum, sum};
        }
        case 3: {
            const Float sum = vv[0] + vv[1] + vv[2];
            return InverseDepth{InverseDepth::WithVariance, (dv[0] * vv[0] + dv[1] * vv[1] + dv[2] * vv[2]) / sum, sum};
        }
        case 4: {
            const Float sum = vv[0] + vv[1] + vv[2] + vv[3];
            return InverseDepth{InverseDepth::WithVariance,
                                (dv[0] * vv[0] + dv[1] * vv[1] + dv[2] * vv[2] + dv[3] * vv[3]) / sum, sum};
        }
        default: return InverseDepth{InverseDepth::Unknown, 0, 0};
    }
}
}  // namespace inverse_depth

// ---------------------------------------------------------------------------------------------
// src/math/optimizer.rs — the generic "optimizer trait": static-polymorphic restatement.
// S must provide: static S init(const Obs&, Model); bool step(Model&, Error&) const;
// EvalState eval(const Obs&, Model) const; static Continue stop_criterion(S&, size_t, EvalState&&).
// ---------------------------------------------------------------------------------------------
namespace optimizer {
enum class Continue { Stop, Forward };  // optimizer.rs:9-14
// optimizer.rs:57-70
template <class S, class Obs, class Model, class Error>
bool iterative_solve(const Obs& obs, Model initial_model, S& state_out, size_t& nb_iter_out, Error& err) {
    S state = S::init(obs, initial_model);
    size_t nb_iter = 0;
    for (;;) {
        nb_iter += 1;
        Model new_model;
        if (!state.step(new_model, err)) return false;
        auto eval_state = state.eval(obs, new_model);
        const Continue c = S::stop_criterion(state, nb_iter, std::move(eval_state));
        if (c == Continue::Stop) {
            state_out = std::move(state);
            nb_iter_out = nb_iter;
            return true;
        }
    }
}
}  // namespace optimizer

// ---------------------------------------------------------------------------------------------
// src/core/track/lm_optimizer.rs
// ---------------------------------------------------------------------------------------------
namespace lm_optimizer {

struct EvalData {  // lm_optimizer.rs:31-40
    Mat6 hessian;
    Vec6 gradient;
    Float energy;
    Iso3 model;
};
struct EvalState {  // Result<EvalData, Float>  (lm_optimizer.rs:28)
    bool ok;
    EvalData data;
    Float err_energy;
};
struct Obs {  // lm_optimizer.rs:43-58
    const Intrinsics* intrinsics;
    const DMatrix<uint8_t>* template_;
    const DMatrix<uint8_t>* image;
    const std::vector<std::pair<size_t, size_t>>* coordinates;
    const std::vector<Float>* _z_candidates;
    const std::vector<Vec6>* jacobians;
    const std::vector<Mat6>* hessians;
    // Extension (NOT in the reference): Huber threshold on |r|; <= 0 disables.
    Float huber_delta = 0.0f;
};
struct Precomputed {  // lm_optimizer.rs:61
    Float energy;
    std::vector<size_t> inside_indices;
    std::vector<Float> residuals;
};

// lm_optimizer.rs:213-219 (duplicate at inverse_compositional.rs:352-358)
inline void warp(const Iso3& model, Float x, Float y, Float _z, const Intrinsics& intr, Float& u, Float& v) {
    const Vec3 x1 = intr.back_project(x, y, 1.0f / _z);
    const Vec3 x2 = iso_transform_point(model, x1);
    const Vec3 uvz2 = intr.project(x2);
    u = uvz2.x / uvz2.z;
    v = uvz2.y / uvz2.z;
}
// lm_optimizer.rs:227-251
inline bool interpolate(Float x, Float y, const DMatrix<uint8_t>& image, Float& out) {
    const int height = image.nrows, width = image.ncols;
    const Float u = std::floor(x), v = std::floor(y);
    if (u >= 0.0f && u < (Float)(width - 2) && v >= 0.0f && v < (Float)(height - 2)) {
        const int u_0 = (int)u, v_0 = (int)v;
        const int u_1 = u_0 + 1, v_1 = v_0 + 1;
        const Float vu_00 = (Float)image(v_0, u_0);
        const Float vu_10 = (Float)image(v_1, u_0);
        const Float vu_01 = (Float)image(v_0, u_1);
        const Float vu_11 = (Float)image(v_1, u_1);
        const Float a = x - u, b = y - v;
        out = (1.0f - b) * (1.0f - a) * vu_00 + b * (1.0f - a) * vu_10 + (1.0f - b) * a * vu_01 + b * a * vu_11;
        return true;
    }
    return false;
}

struct LMOptimizerState {  // lm_optimizer.rs:16-21
    Float lm_coef;
    EvalData eval_data;

    // lm_optimizer.rs:68-87
    static Precomputed eval_energy(const Obs& obs, const Iso3& model) {
        Precomputed pre;
        Acc energy_sum = 0;
        const auto& coords = *obs.coordinates;
        for (size_t idx = 0; idx < coords.size(); ++idx) {
            const size_t x = coords[idx].first, y = coords[idx].second;
            const Float _z = (*obs._z_candidates)[idx];
            Float u, v, im;
            warp(model, (Float)x, (Float)y, _z, *obs.intrinsics, u, v);
            if (interpolate(u, v, *obs.image, im)) {
                const uint8_t tmp = (*obs.template_)((int)y, (int)x);
                const Float r = im - (Float)tmp;
                if (obs.huber_delta > 0.0f) {  // extension, parity unpinned
                    const Float ar = std::fabs(r);
                    energy_sum += (Acc)((ar <= obs.huber_delta) ? r * r : obs.huber_delta * (2.0f * ar - obs.huber_delta));
                } else {
                    energy_sum += (Acc)(r * r);
                }
                pre.residuals.push_back(r);
                pre.inside_indices.push_back(idx);
            }
        }
        pre.energy = (Float)(energy_sum / (Acc)pre.residuals.size());
        return pre;
    }
    // lm_optimizer.rs:90-107
    static EvalData compute_eval_data(const Obs& obs, const Iso3& model, const Precomputed& pre) {
        EvalData e;
        Acc grad[6], hess[6][6];
        for (int a = 0; a < 6; ++a) {
            grad[a] = 0;
            for (int b = 0; b < 6; ++b) hess[a][b] = 0;
        }
        for (size_t i = 0; i < pre.inside_indices.size(); ++i) {
            const size_t idx = pre.inside_indices[i];
            const Vec6& jac = (*obs.jacobians)[idx];
            const Mat6& hes = (*obs.hessians)[idx];
            const Float r = pre.residuals[i];
            if (obs.huber_delta > 0.0f) {  // extension
                const Float ar = std::fabs(r);
                const Float w = (ar <= obs.huber_delta) ? 1.0f : obs.huber_delta / ar;
                for (int a = 0; a < 6; ++a) grad[a] += (Acc)(jac.v[a] * (w * r));
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) hess[a][b] += (Acc)(w * hes.m[a][b]);
            } else {
                for (int a = 0; a < 6; ++a) grad[a] += (Acc)(jac.v[a] * r);
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) hess[a][b] += (Acc)hes.m[a][b];
            }
        }
        for (int a = 0; a < 6; ++a) {
            e.gradient.v[a] = (Float)grad[a];
            for (int b = 0; b < 6; ++b) e.hessian.m[a][b] = (Float)hess[a][b];
        }
        e.energy = pre.energy;
        e.model = model;
        return e;
    }
    // lm_optimizer.rs:113-118
    static LMOptimizerState init(const Obs& obs, const Iso3& model) {
        return LMOptimizerState{0.1f, compute_eval_data(obs, model, eval_energy(obs, model))};
    }
    // lm_optimizer.rs:123-136
    bool step(Iso3& out, std::string& err) const {
        Mat6 hessian = eval_data.hessian;
        for (int a = 0; a < 6; ++a) hessian.m[a][a] *= 1.0f + lm_coef;
        if (!cholesky6(hessian)) {
            err = "Error at Cholesky decomposition of hessian";
            return false;
        }
        const Iso3 delta_warp = se3::exp(cholesky6_solve(hessian, eval_data.gradient));
        out = renormalize(iso_mul(eval_data.model, iso_inverse(delta_warp)));
        return true;
    }
    // lm_optimizer.rs:140-149
    EvalState eval(const Obs& obs, const Iso3& model) const {
        const Precomputed pre = eval_energy(obs, model);
        EvalState s;
        if (pre.energy > eval_data.energy) {
            s.ok = false;
            s.err_energy = pre.energy;
        } else {
            s.ok = true;
            s.data = compute_eval_data(obs, model, pre);
        }
        return s;
    }
    // lm_optimizer.rs:156-192
    static optimizer::Continue stop_criterion(LMOptimizerState& self, size_t nb_iter, EvalState&& eval_state) {
        const bool too_many_iterations = nb_iter > 20;
        if (!eval_state.ok && too_many_iterations) return optimizer::Continue::Stop;
        if (eval_state.ok && too_many_iterations) {
            self.eval_data = eval_state.data;
            return optimizer::Continue::Stop;
        }
        if (!eval_state.ok) {
            self.lm_coef *= 10.0f;
            return optimizer::Continue::Forward;
        }
        const Float d_energy = self.eval_data.energy - eval_state.data.energy;
        const optimizer::Continue c = d_energy > 1.0f ? optimizer::Continue::Forward : optimizer::Continue::Stop;
        self.lm_coef = 0.1f * self.lm_coef;
        self.eval_data = eval_state.data;
        return c;
    }
    // lm_optimizer.rs:198-209
    static Iso3 renormalize(Iso3 motion) {
        const Float sq_norm = quat_norm_squared(motion.q);
        const Float f = 0.5f * (3.0f - sq_norm);
        motion.q = Quat{f * motion.q.i, f * motion.q.j, f * motion.q.k, f * motion.q.w};
        return motion;
    }
    static bool iterative_solve(const Obs& obs, const Iso3& model, LMOptimizerState& out, size_t& nb_iter, std::string& err) {
        return optimizer::iterative_solve<LMOptimizerState, Obs, Iso3, std::string>(obs, model, out, nb_iter, err);
    }
};
}  // namespace lm_optimizer

// ---------------------------------------------------------------------------------------------
// src/core/track/inverse_compositional.rs
// ---------------------------------------------------------------------------------------------
namespace track {

// Candidate-mask source. 0 = candidates::coarse_to_fine (the reference, inverse_compositional.rs:120-125).
// 1 = dense all-true level-0 mask (build extension for BASELINE config 2; SURVEY.md top table).
enum CandidatesMode : int { COARSE_TO_FINE = 0, DENSE = 1, DSO = 2 };
constexpr size_t DSO_NB_TARGET = 2000;      // examples/candidates_dso.rs:58
constexpr uint64_t DSO_SEED = 0x5EEDD50ull;  // counter-RNG seed of the sub-sampling branch (deviation, see namespace dso)

struct Config {  // inverse_compositional.rs:37-49
    size_t nb_levels;
    uint16_t candidates_diff_threshold;
    Float depth_scale;
    Intrinsics intrinsics;
    Float idepth_variance;
    // extensions (defaults reproduce the reference)
    int candidates_mode = COARSE_TO_FINE;
    Float huber_delta = 0.0f;
};

struct MultiresData {  // inverse_compositional.rs:64-70
    std::vector<Intrinsics> intrinsics_multires;
    std::vector<DMatrix<uint8_t>> img_multires;
    std::vector<std::pair<std::vector<std::pair<size_t, size_t>>, std::vector<Float>>> usable_candidates_multires;
    std::vector<std::vector<Vec6>> jacobians_multires;
    std::vector<std::vector<Mat6>> hessians_multires;
    // Kept for inspection by tests (the reference drops these at the end of precompute_multires_data).
    std::vector<std::pair<DMatrix<int16_t>, DMatrix<int16_t>>> gradients_multires;
    std::vector<DMatrix<uint16_t>> gradients_squared_norm_multires;
    DMatrix<uint8_t> candidates_points;
    bool keep_debug = false;
};

// inverse_compositional.rs:260-279 — column-major walk (x outer, y inner).
inline std::pair<std::vector<std::pair<size_t, size_t>>, std::vector<Float>> extract_z(const DMatrix<InverseDepth>& m) {
    std::pair<std::vector<std::pair<size_t, size_t>>, std::vector<Float>> out;
    size_t u = 0, v = 0;
    const size_t nb_rows = (size_t)m.nrows;
    for (const InverseDepth& idepth : m.data) {
        if (idepth.kind == InverseDepth::WithVariance) {
            out.first.emplace_back(u, v);
            out.second.push_back(idepth.d);
        }
        v += 1;
        if (v >= nb_rows) {
            u += 1;
            v = 0;
        }
    }
    return out;
}
// inverse_compositional.rs:313-341
inline Vec6 warp_jacobian_at(Float gu, Float gv, Float u, Float v, Float _z, Float cu, Float cv, Float fu, Float fv, Float s) {
    const Float a = u - cu;
    const Float b = v - cv;
    const Float c = a * fv - s * b;
    const Float _fv = 1.0f / fv;
    const Float _fuv = 1.0f / (fu * fv);
    return Vec6{{
        gu * _z * fu,
        _z * (gu * s + gv * fv),
        -_z * (gu * a + gv * b),
        gu * (-a * b * _fv - s) + gv * (-b * b * _fv - fv),
        gu * (a * c * _fuv + fu) + gv * (b * c * _fuv),
        gu * (-fu * fu * b + s * c) * _fuv + gv * (c / fu),
    }};
}
// inverse_compositional.rs:284-306
inline std::vector<Vec6> warp_jacobians(const Intrinsics& intr, const std::vector<std::pair<size_t, size_t>>& coordinates,
                                        const std::vector<Float>& _z_candidates, const DMatrix<int16_t>& grad_x,
                                        const DMatrix<int16_t>& grad_y) {
    std::vector<Vec6> out;
    out.reserve(coordinates.size());
    for (size_t k = 0; k < coordinates.size(); ++k) {
        const size_t u = coordinates[k].first, v = coordinates[k].second;
        const Float gu = (Float)grad_x((int)v, (int)u);
        const Float gv = (Float)grad_y((int)v, (int)u);
        out.push_back(warp_jacobian_at(gu, gv, (Float)u, (Float)v, _z_candidates[k], intr.cu, intr.cv, intr.fu, intr.fv, intr.skew));
    }
    return out;
}
// inverse_compositional.rs:345-348
inline std::vector<Mat6> hessians_vec(const std::vector<Vec6>& jacobians) {
    std::vector<Mat6> out(jacobians.size());
    for (size_t k = 0; k < jacobians.size(); ++k)
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) out[k].m[a][b] = jacobians[k].v[a] * jacobians[k].v[b];
    return out;
}

// inverse_compositional.rs:105-161
inline MultiresData precompute_multires_data(const Config& config, const DMatrix<uint16_t>& depth_map,
                                             std::vector<Intrinsics> intrinsics_multires,
                                             std::vector<DMatrix<uint8_t>> img_multires, bool keep_debug) {
    auto gradients_multires = gradients_xy(img_multires);
    {
        DMatrix<int16_t> gx, gy;
        gradient::centered(img_multires[0], gx, gy);
        gradients_multires.insert(gradients_multires.begin(), std::make_pair(std::move(gx), std::move(gy)));
    }
    std::vector<DMatrix<uint16_t>> g2;
    for (auto& g : gradients_multires) g2.push_back(gradient::squared_norm(g.first, g.second));

    DMatrix<uint8_t> candidates_points;
    if (config.candidates_mode == DENSE) {
        candidates_points = DMatrix<uint8_t>(img_multires[0].nrows, img_multires[0].ncols, (uint8_t)1);
    } else if (config.candidates_mode == DSO) {
        candidates_points = dso::select_like_example(img_multires[0], DSO_NB_TARGET, DSO_SEED);
    } else {
        auto masks = candidates::select(config.candidates_diff_threshold, g2);
        candidates_points = std::move(masks.back());  // .pop().unwrap()
    }

    // helper::zip_mask_map (src/misc/helper.rs:40-47) with inverse_depth::from_depth
    DMatrix<InverseDepth> idepth_candidates(depth_map.nrows, depth_map.ncols, InverseDepth{InverseDepth::Unknown, 0, 0});
    for (size_t k = 0; k < idepth_candidates.data.size(); ++k)
        if (candidates_points.data[k])
            idepth_candidates.data[k] = inverse_depth::from_depth(config.depth_scale, depth_map.data[k], config.idepth_variance);

    auto idepth_multires = limited_sequence(
        config.nb_levels, std::move(idepth_candidates), [](const DMatrix<InverseDepth>& m, DMatrix<InverseDepth>& o) {
            return halve<InverseDepth, InverseDepth>(m, inverse_depth::fuse_dso_mean, o);
        });

    MultiresData out;
    for (auto& m : idepth_multires) out.usable_candidates_multires.push_back(extract_z(m));
    const size_t n = std::min(intrinsics_multires.size(), std::min(out.usable_candidates_multires.size(), gradients_multires.size()));
    for (size_t l = 0; l < n; ++l)
        out.jacobians_multires.push_back(warp_jacobians(intrinsics_multires[l], out.usable_candidates_multires[l].first,
                                                        out.usable_candidates_multires[l].second, gradients_multires[l].first,
                                                        gradients_multires[l].second));
    for (auto& j : out.jacobians_multires) out.hessians_multires.push_back(hessians_vec(j));
    out.intrinsics_multires = std::move(intrinsics_multires);
    out.img_multires = std::move(img_multires);
    out.keep_debug = keep_debug;
    if (keep_debug) {
        out.gradients_multires = std::move(gradients_multires);
        out.gradients_squared_norm_multires = std::move(g2);
        out.candidates_points = std::move(candidates_points);
    }
    return out;
}

struct LevelStats {
    int nb_iter = 0;      // value returned by iterative_solve
    int n_points = 0;     // candidates at this level
    Float energy = 0.0f;  // energy of the kept state
};

struct Tracker {  // inverse_compositional.rs:31-34 + 52-60
    Config config;
    MultiresData keyframe_multires_data;
    double keyframe_depth_timestamp, keyframe_img_timestamp;
    Iso3 keyframe_pose;
    double current_frame_depth_timestamp, current_frame_img_timestamp;
    Iso3 current_frame_pose;
    // inspection only
    bool keep_debug = false;
    bool last_optimization_went_well = true;
    bool last_changed_keyframe = false;
    Float last_optical_flow = 0.0f;
    Iso3 last_lm_model = iso_identity();
    std::vector<LevelStats> last_level_stats;
    std::string last_error;

    // inverse_compositional.rs:74-100. Returns false where the reference would panic (pyramid shorter than nb_levels).
    static bool init(const Config& config, double depth_t, const DMatrix<uint16_t>& depth_map, double img_t,
                     DMatrix<uint8_t> img, bool keep_debug, Tracker& out) {
        auto intrinsics_multires = config.intrinsics.multi_res(config.nb_levels);
        auto img_multires = mean_pyramid(config.nb_levels, std::move(img));
        if (img_multires.size() < config.nb_levels || config.nb_levels == 0) return false;
        out.config = config;
        out.keep_debug = keep_debug;
        out.keyframe_multires_data =
            precompute_multires_data(config, depth_map, std::move(intrinsics_multires), std::move(img_multires), keep_debug);
        out.keyframe_depth_timestamp = depth_t;
        out.keyframe_img_timestamp = img_t;
        out.keyframe_pose = iso_identity();
        out.current_frame_depth_timestamp = depth_t;
        out.current_frame_img_timestamp = img_t;
        out.current_frame_pose = iso_identity();
        return true;
    }

    // inverse_compositional.rs:170-240. Returns 0 = ok, 1 = optimizer failed and the pose was kept.
    int track(double depth_time, const DMatrix<uint16_t>& depth_map, double img_time, DMatrix<uint8_t> img) {
        Iso3 lm_model = iso_mul(iso_inverse(current_frame_pose), keyframe_pose);
        auto img_multires = mean_pyramid(config.nb_levels, std::move(img));
        const MultiresData& keyframe_data = keyframe_multires_data;
        bool optimization_went_well = true;
        last_level_stats.assign(config.nb_levels, LevelStats());
        for (size_t l = 0; l < config.nb_levels; ++l)
            last_level_stats[l].n_points = (int)keyframe_data.usable_candidates_multires[l].second.size();
        last_error.clear();
        for (int lvl = (int)config.nb_levels - 1; lvl >= 0; --lvl) {
            lm_optimizer::Obs obs;
            obs.intrinsics = &keyframe_data.intrinsics_multires[lvl];
            obs.template_ = &keyframe_data.img_multires[lvl];
            obs.image = &img_multires[lvl];
            obs.coordinates = &keyframe_data.usable_candidates_multires[lvl].first;
            obs._z_candidates = &keyframe_data.usable_candidates_multires[lvl].second;
            obs.jacobians = &keyframe_data.jacobians_multires[lvl];
            obs.hessians = &keyframe_data.hessians_multires[lvl];
            obs.huber_delta = config.huber_delta;
            lm_optimizer::LMOptimizerState lm_state;
            size_t nb_iter = 0;
            std::string err;
            if (lm_optimizer::LMOptimizerState::iterative_solve(obs, lm_model, lm_state, nb_iter, err)) {
                lm_model = lm_state.eval_data.model;
                last_level_stats[lvl].nb_iter = (int)nb_iter;
                last_level_stats[lvl].energy = lm_state.eval_data.energy;
            } else {
                last_error = err;  // eprintln!("{}", err)
                optimization_went_well = false;
                break;
            }
        }
        current_frame_depth_timestamp = depth_time;
        current_frame_img_timestamp = img_time;
        if (optimization_went_well) current_frame_pose = iso_mul(keyframe_pose, iso_inverse(lm_model));

        // keyframe test on the coarsest level (inverse_compositional.rs:211-224)
        const auto& last = keyframe_data.usable_candidates_multires.back();
        const Intrinsics& intr = keyframe_data.intrinsics_multires.back();
        Acc optical_flow_sum = 0;
        for (size_t k = 0; k < last.second.size(); ++k) {
            const Float x = (Float)last.first[k].first, y = (Float)last.first[k].second;
            Float u, v;
            lm_optimizer::warp(lm_model, x, y, last.second[k], intr, u, v);
            optical_flow_sum += (Acc)(std::fabs(x - u) + std::fabs(y - v));
        }
        const Float optical_flow = (Float)(optical_flow_sum / (Acc)last.second.size());
        const bool change_keyframe = optical_flow >= 1.0f;
        last_optical_flow = optical_flow;
        last_changed_keyframe = change_keyframe;
        last_optimization_went_well = optimization_went_well;
        last_lm_model = lm_model;
        if (change_keyframe) {
            auto intr_copy = keyframe_data.intrinsics_multires;
            keyframe_multires_data =
                precompute_multires_data(config, depth_map, std::move(intr_copy), std::move(img_multires), keep_debug);
            keyframe_depth_timestamp = depth_time;
            keyframe_img_timestamp = img_time;
            keyframe_pose = current_frame_pose;
        }
        return optimization_went_well ? 0 : 1;
    }
};
}  // namespace track

}  // namespace vors_oracle
