// lie.h's per-point arithmetic, generic over the scalar type: F = float, or F = F2 — TWO points per lane, every operation written for
// both before the next one. gfx950 does not overlap dependent VALU instructions across wavefronts (a dependent instruction issues every
// 4.6 cycles whatever the occupancy, an independent one every 2.6: DESIGN.md §3, tools/ubench/valu_occ.hip), so the instruction-level
// parallelism has to be in the instruction stream itself. Every function below is the expression-for-expression twin of the lie.h function
// named in its comment (each component runs the same f32 operations in the same order; compile with -ffp-contract=off): the results are
// bit-identical per point, which the REFERENCE-arithmetic tests assert against the oracle. Device code of lm_reference.hip only.
#pragma once
#include "lie.h"

namespace vors {

struct F2 {
    float a, b;
};
#define VORS_G __device__ __forceinline__
VORS_G F2 operator+(F2 x, F2 y) { return F2{x.a + y.a, x.b + y.b}; }
VORS_G F2 operator-(F2 x, F2 y) { return F2{x.a - y.a, x.b - y.b}; }
VORS_G F2 operator*(F2 x, F2 y) { return F2{x.a * y.a, x.b * y.b}; }
VORS_G F2 operator/(F2 x, F2 y) { return F2{x.a / y.a, x.b / y.b}; }
VORS_G F2 operator+(F2 x, float y) { return F2{x.a + y, x.b + y}; }
VORS_G F2 operator-(F2 x, float y) { return F2{x.a - y, x.b - y}; }
VORS_G F2 operator*(F2 x, float y) { return F2{x.a * y, x.b * y}; }
VORS_G F2 operator/(F2 x, float y) { return F2{x.a / y, x.b / y}; }
VORS_G F2 operator+(float x, F2 y) { return F2{x + y.a, x + y.b}; }
VORS_G F2 operator-(float x, F2 y) { return F2{x - y.a, x - y.b}; }
VORS_G F2 operator*(float x, F2 y) { return F2{x * y.a, x * y.b}; }
VORS_G F2 operator/(float x, F2 y) { return F2{x / y.a, x / y.b}; }
VORS_G F2 operator-(F2 x) { return F2{-x.a, -x.b}; }
VORS_G float g_fma(float x, float y, float z) { return fmaf(x, y, z); }
VORS_G F2 g_fma(F2 x, float y, F2 z) { return F2{fmaf(x.a, y, z.a), fmaf(x.b, y, z.b)}; }
VORS_G float g_copysign(float m, float s) { return __builtin_copysignf(m, s); }
VORS_G F2 g_copysign(F2 m, F2 s) { return F2{__builtin_copysignf(m.a, s.a), __builtin_copysignf(m.b, s.b)}; }
VORS_G float g_floor(float x) { return floorf(x); }
VORS_G F2 g_floor(F2 x) { return F2{floorf(x.a), floorf(x.b)}; }

template <class F>
struct V3G {
    F x, y, z;
};

// lie.h div_uniform<FAST>
template <bool FAST, class F>
VORS_G F g_div_uniform(F x, const FastDiv& f) {
    if (FAST) {
        const F q = x * f.r;
        const F e = g_fma(-q, f.d, x);
        const F q1 = g_fma(e, f.r, q);
        return g_copysign(q1, q);
    }
    return x / f.d;
}
// lie.h cross, with a uniform left operand
template <class F>
VORS_G V3G<F> g_cross(const V3& a, const V3G<F>& b) {
    return V3G<F>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// lie.h quat_rotate (UnitQuaternion * Vector3:  t = 2 (qv x p);  p' = (t w + qv x t) + p)
template <class F>
VORS_G V3G<F> g_quat_rotate(const Quat& q, const V3G<F>& p) {
    const V3 qv{q.i, q.j, q.k};
    V3G<F> t = g_cross(qv, p);
    t = V3G<F>{t.x * 2.0f, t.y * 2.0f, t.z * 2.0f};
    const V3G<F> c = g_cross(qv, t);
    return V3G<F>{(t.x * q.w + c.x) + p.x, (t.y * q.w + c.y) + p.y, (t.z * q.w + c.z) + p.z};
}
// lie.h iso_transform_point
template <class F>
VORS_G V3G<F> g_iso_transform_point(const Iso& m, const V3G<F>& p) {
    const V3G<F> r = g_quat_rotate(m.q, p);
    return V3G<F>{r.x + m.t.x, r.y + m.t.y, r.z + m.t.z};
}
// lie.h back_project_fast<FAST> (camera.rs:135-140)
template <bool FAST, class F>
VORS_G V3G<F> g_back_project(const IntrFast& kf, F px, F py, F depth) {
    const F z = depth;
    const F y = g_div_uniform<FAST>((py - kf.k.cv) * z, kf.fv);
    const F x = g_div_uniform<FAST>((px - kf.k.cu) * z - kf.k.skew * y, kf.fu);
    return V3G<F>{x, y, z};
}
// lie.h project_uv (camera.rs:126-132 and lm_optimizer.rs:217-218)
template <class F>
VORS_G void g_project_uv(const Intr& k, const V3G<F>& p, F* u, F* v) {
    const F pu = (k.fu * p.x + k.skew * p.y) + k.cu * p.z;
    const F pv = k.fv * p.y + k.cv * p.z;
    *u = pu / p.z;
    *v = pv / p.z;
}
// The two reciprocals of warp_jacobian_at that depend on the level only (`1.0 / fv`, `1.0 / (fu * fv)`, inverse_compositional.rs:330-331):
// the same two f32 divisions, done once per level instead of once per trip of the evaluation loop (the compiler re-materialises uniform
// values inside the loop rather than keep them in registers: 22 instructions per trip).
struct JacRecip {
    float _fv, _fuv;
};
VORS_G JacRecip g_jac_recip(const Intr& k) { return JacRecip{1.0f / k.fv, 1.0f / (k.fu * k.fv)}; }
// lie.h warp_jacobian_at_fast<FAST> (inverse_compositional.rs:313-341)
template <bool FAST, class F>
VORS_G void g_warp_jacobian_at(F gu, F gv, F u, F v, F _z, const IntrFast& kf, const JacRecip& rc, F J[6]) {
    const Intr& k = kf.k;
    const F a = u - k.cu;
    const F b = v - k.cv;
    const F c = a * k.fv - k.skew * b;
    const float _fv = rc._fv;
    const float _fuv = rc._fuv;
    J[0] = gu * _z * k.fu;
    J[1] = _z * (gu * k.skew + gv * k.fv);
    J[2] = -_z * (gu * a + gv * b);
    J[3] = gu * (-a * b * _fv - k.skew) + gv * (-b * b * _fv - k.fv);
    J[4] = gu * (a * c * _fuv + k.fu) + gv * (b * c * _fuv);
    J[5] = gu * (-k.fu * k.fu * b + k.skew * c) * _fuv + gv * g_div_uniform<FAST>(c, kf.fu);
}

}  // namespace vors
