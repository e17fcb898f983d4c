// REFERENCE arithmetic (vors_config.arithmetic = VORS_ARITH_REFERENCE): the LM evaluation with the reference's OWN SUMMATION ORDER.
//
// The reference sums sequentially in f32, candidate after candidate in extract_z's column-major order (x outer, y inner:
// src/core/track/inverse_compositional.rs:260-279):
//     energy_sum += r * r                      src/core/track/lm_optimizer.rs:72-84
//     gradient   += jac * r ; hessian += hes   lm_optimizer.rs:94-100   (hes = jac * jac^T precomputed, inverse_compositional.rs:345-348)
// i.e. one rounding for the product, one for the addition, no FMA, outside points skipped. The accept / stop comparisons of the LM loop
// (lm_optimizer.rs:144,179) are decided at ties, so ANY other order of these additions forks the loop in ~60 % of the pairs and leaves
// a 0.05-0.4 % tail beyond 1e-4. The EXACT arithmetic already gives bit-identical per-point residuals and Jacobians;
// this file adds the order (round 5: as a throughput mode; it is the boundary's DEFAULT arithmetic, include/vors_hip.h):
//
//   * candidate lists in column-major order: coarse-to-fine — rank_regions_kernel packs AND orders the keyframe kernel's staged regions in
//     one pass (an LDS bitmap of the level, rank = set bits below the key x * rows + y: keys are unique); DSO — sort_colmajor_kernel
//     re-orders the Morton-sorted lists the same way; dense — 8-byte column-major records written once per keyframe
//     (ref_dense_records_*_kernel), so that point i of the reference's enumeration is record i;
//   * ONE WAVEFRONT per frame pair (lm_ref_track_kernel): every lane evaluates one or two points in the reference's per-point arithmetic
//     (lie.h, lie_g.h) and stores the 28 PRODUCTS the reference adds — r * r, jac * r, jac * jac^T, each rounded once — product-major into
//     the wavefront's own 7.6 KB of LDS; lane k < 28 then walks row k with 64 dependent v_add_f32: the chain the reference performs for
//     sum k. An outside point stores +0: adding +-0 is exact, so it equals skipping the point. No barrier; the records of the trip after
//     next and the taps of the next trip are requested before a trip's chains run (refw_eval / refw_eval2);
//   * small batches, single trackers and the stragglers of a large batch: a WORKGROUP per pair (lm_ref_track_coop_kernel) — one wavefront
//     owns the chains and the LM state machine (ref_lm_advance), the others produce product blocks; the same chains, bit for bit;
//   * step() with the IEEE Cholesky of lie.h and sinf / cosf as glibc computes them (lie.h ref_sinf / ref_cosf), the optical-flow sum
//     of the keyframe test in list order as well (inverse_compositional.rs:213-221).
//
// Result: the device follows the oracle's LM path decision for decision — iteration counts equal at every level, poses bit-identical
// (tests/test_gpu_reference.py asserts equality, not a tolerance). Cost against FUSED at 4096 pairs: 1.3x (coarse-to-fine, DSO), 3.8x
// (dense); what binds it is the number of VALU instructions of the reference's own per-point expressions.
//
// Compile with -ffp-contract=off (no FMA may be formed from a product and the addition that follows it).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "device_common.h"
#include "engine.h"
#include "lie_g.h"

namespace vors {

// ---- the wavefront-private transposition buffer -------------------------------------------------------------------------------------
#define RW_WPB 4                         // independent wavefronts (= frame pairs) per workgroup, at most
#define RW_NSUM 28                       // sums: 0 = sum r^2 (Huber: loss), 1..6 = g, 7..27 = H upper triangle row-wise
#define RW_STRIDE 68                     // words per product row: 64 points + 4 (see refw_eval: conflict-free ds_read_b128 by lanes 0..27)
#define RW_WORDS (RW_NSUM * RW_STRIDE)   // 7,616 bytes per wavefront

struct RefImg {  // the current image of a level + its intrinsics
    const uint8_t* img;
    int rows, cols;
    Intr k;
    float huber;
};

// ---- point providers: point i of the level, in the reference's order. Three steps, so that refw_eval can keep the memory round trips of
// the NEXT 64 points in flight while it works on the current ones:
//   load(i)   issues the loads of everything the point is made of -> Raw (registers)
//   point()   the back-projected keyframe point (camera.rs:135-140) + whether the slot holds a candidate
//   jac()     the warp Jacobian (inverse_compositional.rs:313-341) + the template grey level
struct RefSlimSrc {  // sparse modes: the 12-byte lists in extract_z's order
    const SlimRec* S;
    Intr k;
    FastDiv fu, fv;  // the focal lengths as verified fast divisors (lie.h div_uniform: bit-identical to the IEEE quotient, or `ok` = 0)
    JacRecip rc;     // 1 / fv, 1 / (fu fv): once per level, in scalar registers (lie_g.h)
    static constexpr bool kTransposed = false;  // the current image is row-major
    typedef SlimRec Raw;
    __device__ __forceinline__ Raw load(int i) const { return S[(unsigned)i]; }
    __device__ __forceinline__ void point(const Raw& r, V3* P, bool* valid) const {
        *P = back_project_rt(IntrFast{k, fu, fv}, (float)(r.xy & 0xffffu), (float)(r.xy >> 16), 1.0f / r.iz);
        *valid = true;
    }
    __device__ __forceinline__ void jac(const Raw& r, float J[6], float* tmpl) const {
        *tmpl = (float)(r.tg & 0xffu);
        const float gx = (float)slim_gx(r.tg), gy = (float)slim_gy(r.tg), x = (float)(r.xy & 0xffffu), y = (float)(r.xy >> 16);
        if (fu.ok) g_warp_jacobian_at<true>(gx, gy, x, y, r.iz, IntrFast{k, fu, fv}, rc, J);  // (uniform branch; = lie.h warp_jacobian_at_rt)
        else g_warp_jacobian_at<false>(gx, gy, x, y, r.iz, IntrFast{k, fu, fv}, rc, J);
    }
    // two points per lane (lie_g.h); FAST: both focal lengths are verified fast divisors
    template <bool FAST>
    __device__ __forceinline__ void point2(const Raw& ra, const Raw& rb, V3G<F2>* P, bool* va, bool* vb) const {
        const F2 x{(float)(ra.xy & 0xffffu), (float)(rb.xy & 0xffffu)}, y{(float)(ra.xy >> 16), (float)(rb.xy >> 16)};
        *P = g_back_project<FAST>(IntrFast{k, fu, fv}, x, y, 1.0f / F2{ra.iz, rb.iz});
        *va = true;
        *vb = true;
    }
    template <bool FAST>
    __device__ __forceinline__ void jac2(const Raw& ra, const Raw& rb, F2 J[6], F2* tmpl) const {
        *tmpl = F2{(float)(ra.tg & 0xffu), (float)(rb.tg & 0xffu)};
        const F2 x{(float)(ra.xy & 0xffffu), (float)(rb.xy & 0xffffu)}, y{(float)(ra.xy >> 16), (float)(rb.xy >> 16)};
        g_warp_jacobian_at<FAST>(F2{(float)slim_gx(ra.tg), (float)slim_gx(rb.tg)}, F2{(float)slim_gy(ra.tg), (float)slim_gy(rb.tg)}, x, y,
                                 F2{ra.iz, rb.iz}, IntrFast{k, fu, fv}, rc, J);
    }
    __device__ __forceinline__ void xy_iz(int i, float* x, float* y, float* iz, bool* valid) const {
        const SlimRec r = S[(unsigned)i];
        *x = (float)(r.xy & 0xffffu);
        *y = (float)(r.xy >> 16);
        *iz = r.iz;
        *valid = true;
    }
};
struct RefDenseSrc {  // dense mode on the ROW-MAJOR planes (vors_batch_eval_level with the REFERENCE arithmetic on a handle of another
                      // arithmetic: no column-major planes there): pixel i of the column-major enumeration, gathered
    static constexpr bool kTransposed = false;
    const Geom* g;
    const uint8_t* kf0;
    const uint8_t* kfu;
    const uint16_t* depth;  // level 0, this pair
    const float* iz;        // levels >= 1, this pair and level (row-major), NaN = Unknown
    int pair, lvl, rows, cols;
    Intr k;
    struct Raw {
        uint32_t xy;   // x | y << 16
        float iz;      // inverse depth (anything when !valid)
        uint32_t tgv;  // slim_pack_tg(template, gx, gy) | valid << 31
    };
    __device__ __forceinline__ void raw(int i, int* x, int* y, float* izv, bool* valid) const {
        const int xx = i / rows, yy = i - xx * rows;
        *x = xx;
        *y = yy;
        if (lvl == 0) {
            const int dz = depth[(unsigned)(yy * cols + xx)];
            *valid = dz != 0;
            *izv = g->depth_scale / (float)dz;  // inverse_depth.rs:24-29
        } else {
            const float z = iz[(unsigned)(yy * cols + xx)];
            *valid = !(z != z);
            *izv = z;
        }
    }
    __device__ __forceinline__ Raw load(int i) const {
        int x, y, gx, gy, tm;
        float izv;
        bool valid;
        raw(i, &x, &y, &izv, &valid);
        grad_tmpl_at(*g, kf0, kfu, pair, lvl, x, y, &gx, &gy, &tm);
        return Raw{(uint32_t)x | ((uint32_t)y << 16), izv, slim_pack_tg(tm, gx, gy) | (valid ? 0x80000000u : 0u)};
    }
    __device__ __forceinline__ void point(const Raw& r, V3* P, bool* valid) const {
        *P = back_project(k, (float)(r.xy & 0xffffu), (float)(r.xy >> 16), 1.0f / r.iz);
        *valid = (r.tgv >> 31) != 0u;
    }
    __device__ __forceinline__ void jac(const Raw& r, float J[6], float* tmpl) const {
        *tmpl = (float)(r.tgv & 0xffu);
        warp_jacobian_at((float)slim_gx(r.tgv), (float)slim_gy(r.tgv), (float)(r.xy & 0xffffu), (float)(r.xy >> 16), r.iz, k, J);
    }
    __device__ __forceinline__ void xy_iz(int i, float* x, float* y, float* izv, bool* valid) const {
        int xi, yi;
        raw(i, &xi, &yi, izv, valid);
        *x = (float)xi;
        *y = (float)yi;
    }
};
// Dense mode on the COLUMN-MAJOR records (engine.h RefDensePlanes): pixel i of the enumeration is record i of the level.
struct RefDenseTSrc {
    static constexpr bool kTransposed = true;  // the current image is column-major as well
    const uint2* recs;   // this level of this pair: (inverse depth bits, template | gx | gy | valid << 31)
    int rows;
    uint32_t magic;      // floor(2^32 / rows) + 1: i / rows == __umulhi(i, magic) for i * rows < 2^32 (i < 2^21, rows <= 2^11)
    Intr k;
    FastDiv fu, fv;      // (point2 / jac2; the one-point forms divide)
    JacRecip rc;
    struct Raw {
        uint32_t xy;   // x | y << 16
        float iz;      // inverse depth (anything when !valid)
        uint32_t tgv;  // slim_pack_tg(template, gx, gy) | valid << 31
    };
    __device__ __forceinline__ Raw load(int i) const {
        const uint2 r = recs[(unsigned)i];
        const unsigned x = __umulhi((unsigned)i, magic), y = (unsigned)i - x * (unsigned)rows;
        return Raw{x | (y << 16), __uint_as_float(r.x), r.y};
    }
    __device__ __forceinline__ void raw(int i, int* x, int* y, float* izv, bool* valid) const {
        const Raw r = load(i);
        *x = (int)(r.xy & 0xffffu);
        *y = (int)(r.xy >> 16);
        *izv = r.iz;
        *valid = (r.tgv >> 31) != 0u;
    }
    __device__ __forceinline__ void point(const Raw& r, V3* P, bool* valid) const {
        *P = back_project(k, (float)(r.xy & 0xffffu), (float)(r.xy >> 16), 1.0f / r.iz);
        *valid = (r.tgv >> 31) != 0u;
    }
    __device__ __forceinline__ void jac(const Raw& r, float J[6], float* tmpl) const {
        *tmpl = (float)(r.tgv & 0xffu);
        g_warp_jacobian_at<false>((float)slim_gx(r.tgv), (float)slim_gy(r.tgv), (float)(r.xy & 0xffffu), (float)(r.xy >> 16), r.iz, IntrFast{k, fu, fv}, rc, J);
    }
    template <bool FAST>
    __device__ __forceinline__ void point2(const Raw& ra, const Raw& rb, V3G<F2>* P, bool* va, bool* vb) const {
        const F2 x{(float)(ra.xy & 0xffffu), (float)(rb.xy & 0xffffu)}, y{(float)(ra.xy >> 16), (float)(rb.xy >> 16)};
        *P = g_back_project<FAST>(IntrFast{k, fu, fv}, x, y, 1.0f / F2{ra.iz, rb.iz});
        *va = (ra.tgv >> 31) != 0u;
        *vb = (rb.tgv >> 31) != 0u;
    }
    template <bool FAST>
    __device__ __forceinline__ void jac2(const Raw& ra, const Raw& rb, F2 J[6], F2* tmpl) const {
        *tmpl = F2{(float)(ra.tgv & 0xffu), (float)(rb.tgv & 0xffu)};
        const F2 x{(float)(ra.xy & 0xffffu), (float)(rb.xy & 0xffffu)}, y{(float)(ra.xy >> 16), (float)(rb.xy >> 16)};
        g_warp_jacobian_at<FAST>(F2{(float)slim_gx(ra.tgv), (float)slim_gx(rb.tgv)}, F2{(float)slim_gy(ra.tgv), (float)slim_gy(rb.tgv)}, x, y,
                                 F2{ra.iz, rb.iz}, IntrFast{k, fu, fv}, rc, J);
    }
    __device__ __forceinline__ void xy_iz(int i, float* x, float* y, float* izv, bool* valid) const {
        int xi, yi;
        raw(i, &xi, &yi, izv, valid);
        *x = (float)xi;
        *y = (float)yi;
    }
};
struct RefObsSrc {  // operator level: explicit observations in the caller's order (Obs, lm_optimizer.rs:43-58)
    static constexpr bool kTransposed = false;
    const float4* A;
    const float4* B;
    const float2* C;
    struct Raw {
        float4 a, b;
        float2 c;
    };
    __device__ __forceinline__ Raw load(int i) const { return Raw{A[(unsigned)i], B[(unsigned)i], C[(unsigned)i]}; }
    __device__ __forceinline__ void point(const Raw& r, V3* P, bool* valid) const {
        *P = V3{r.a.x, r.a.y, r.a.z};
        *valid = r.a.w >= 0.f;
    }
    __device__ __forceinline__ void jac(const Raw& r, float J[6], float* tmpl) const {
        *tmpl = r.a.w;
        J[0] = r.b.x; J[1] = r.b.y; J[2] = r.b.z; J[3] = r.b.w; J[4] = r.c.x; J[5] = r.c.y;
    }
};

// warp (lm_optimizer.rs:213-219) of one point and the REQUESTS for the four grey levels interpolate (lm_optimizer.rs:227-251) will read:
// two 16-bit loads (left | right << 8 of the upper and of the lower row).
struct RefTap {
    float fa, fb;     // u - floor(u), v - floor(v)
    uint32_t t0, t1;  // row-major image: taps of row floor(v) and of the row below; column-major: of column floor(u) and of the next one
    bool inside;
};
template <bool TR>
__device__ __forceinline__ RefTap refw_warp(const V3& P, bool valid, const RefImg& c, const Iso& model) {
    float u, v;
    project_uv(c.k, iso_transform_point(model, P), &u, &v);
    const float uf = floorf(u), vf = floorf(v);
    RefTap t;
    t.inside = valid && (uf >= 0.f) && (uf < (float)(c.cols - 2)) && (vf >= 0.f) && (vf < (float)(c.rows - 2));
    const unsigned pitch = (unsigned)(TR ? c.rows : c.cols);
    const unsigned off = t.inside ? (TR ? (unsigned)((int)uf * c.rows + (int)vf) : (unsigned)((int)vf * c.cols + (int)uf)) : 0u;
    uint16_t a, b;
    __builtin_memcpy(&a, c.img + off, 2);
    __builtin_memcpy(&b, c.img + (off + pitch), 2);
    t.t0 = a;
    t.t1 = b;
    t.fa = u - uf;
    t.fb = v - vf;
    return t;
}

// The same for the two points of a lane (lie_g.h).
struct RefTap2 {
    F2 fa, fb;
    uint32_t t0a, t1a, t0b, t1b;
    bool ina, inb;
};
template <bool TR>
__device__ __forceinline__ RefTap2 refw_warp2(const V3G<F2>& P, bool va, bool vb, const RefImg& c, const Iso& model) {
    F2 u, v;
    g_project_uv(c.k, g_iso_transform_point(model, P), &u, &v);
    const F2 uf = g_floor(u), vf = g_floor(v);
    const float cmax = (float)(c.cols - 2), rmax = (float)(c.rows - 2);
    RefTap2 t;
    t.ina = va && (uf.a >= 0.f) && (uf.a < cmax) && (vf.a >= 0.f) && (vf.a < rmax);
    t.inb = vb && (uf.b >= 0.f) && (uf.b < cmax) && (vf.b >= 0.f) && (vf.b < rmax);
    const unsigned pitch = (unsigned)(TR ? c.rows : c.cols);
    const unsigned offa = t.ina ? (TR ? (unsigned)((int)uf.a * c.rows + (int)vf.a) : (unsigned)((int)vf.a * c.cols + (int)uf.a)) : 0u;
    const unsigned offb = t.inb ? (TR ? (unsigned)((int)uf.b * c.rows + (int)vf.b) : (unsigned)((int)vf.b * c.cols + (int)uf.b)) : 0u;
#ifdef VORS_REFW_FAKE_TAPS  // (experiment: every tap from the first 4 KB of the image = the cache-resident bound of the stage)
    const unsigned offa_ = offa & 0xfffu, offb_ = offb & 0xfffu;
#define offa offa_
#define offb offb_
#endif
    uint16_t a0, a1, b0, b1;
    __builtin_memcpy(&a0, c.img + offa, 2);
    __builtin_memcpy(&b0, c.img + offb, 2);
    __builtin_memcpy(&a1, c.img + (offa + pitch), 2);
    __builtin_memcpy(&b1, c.img + (offb + pitch), 2);
#ifdef VORS_REFW_FAKE_TAPS
#undef offa
#undef offb
#endif
#if defined(VORS_REFW_DOUBLE) && (VORS_REFW_DOUBLE & 16)  // ablation: every tap gather issued twice (the second one 64 bytes on: other lines)
    {
        uint16_t a2, a3, b2, b3;
        const unsigned far = t.ina ? 64u : 0u, fbr = t.inb ? 64u : 0u;
        __builtin_memcpy(&a2, c.img + (offa + far), 2);
        __builtin_memcpy(&b2, c.img + (offb + fbr), 2);
        __builtin_memcpy(&a3, c.img + (offa + pitch + far), 2);
        __builtin_memcpy(&b3, c.img + (offb + pitch + fbr), 2);
        if ((unsigned)(a2 + a3 + b2 + b3) == 123457u) a0 += 1;
    }
#endif
    t.t0a = a0; t.t1a = a1; t.t0b = b0; t.t1b = b1;
    t.fa = u - uf;
    t.fb = v - vf;
    return t;
}
// interpolate + residual of both points -> (r, J) with outside points zeroed, (w, loss term) for the Huber extension
template <bool HUBER, bool TR>
__device__ __forceinline__ void refw_residual2(const RefTap2& t, F2 tmpl, float huber, F2 J[6], F2* e, F2* wr, F2* w) {
    const F2 vu_00{(float)(t.t0a & 0xffu), (float)(t.t0b & 0xffu)}, vu_01{(float)(TR ? t.t1a & 0xffu : t.t0a >> 8), (float)(TR ? t.t1b & 0xffu : t.t0b >> 8)};
    const F2 vu_10{(float)(TR ? t.t0a >> 8 : t.t1a & 0xffu), (float)(TR ? t.t0b >> 8 : t.t1b & 0xffu)}, vu_11{(float)(t.t1a >> 8), (float)(t.t1b >> 8)};
    const F2 fa = t.fa, fb = t.fb;
    const F2 im = (1.0f - fb) * (1.0f - fa) * vu_00 + fb * (1.0f - fa) * vu_10 + (1.0f - fb) * fa * vu_01 + fb * fa * vu_11;
    const F2 r_in = im - tmpl;
    const F2 r{t.ina ? r_in.a : 0.f, t.inb ? r_in.b : 0.f};
#pragma unroll
    for (int q = 0; q < 6; ++q) J[q] = F2{t.ina ? J[q].a : 0.f, t.inb ? J[q].b : 0.f};
    if (HUBER) {
        const F2 ar{fabsf(r.a), fabsf(r.b)};
        const bool qa = ar.a <= huber, qb = ar.b <= huber;
        const F2 lin = huber * (2.0f * ar - huber), rr = r * r, wdiv = huber / ar;
        *e = F2{qa ? rr.a : lin.a, qb ? rr.b : lin.b};
        *w = F2{qa ? 1.0f : wdiv.a, qb ? 1.0f : wdiv.b};
        *wr = *w * r;
    } else {
        *e = r * r;
        *wr = r;
        *w = F2{1.0f, 1.0f};
    }
}
// the 28 products of one of the two points (refw_products)
template <bool HUBER>
__device__ __forceinline__ void refw_products_of(const float J[6], float e, float wr, float w, float pr[RW_NSUM]) {
    pr[0] = e;
#pragma unroll
    for (int q = 0; q < 6; ++q) pr[1 + q] = J[q] * wr;
    int k = 7;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int s2 = q; s2 < 6; ++s2) {
            const float jj = J[q] * J[s2];
            pr[k++] = HUBER ? w * jj : jj;
        }
}

// interpolate + residual, then the 28 PRODUCTS the reference adds to its sums for this point — each rounded once, exactly the values of
// `r * r` (lm_optimizer.rs:80), `jac * r` (lm_optimizer.rs:98) and `jac * jac^T` (inverse_compositional.rs:347); Huber extension: loss term,
// jac * (w r), w * (jac * jac^T) like the oracle. An outside point contributes +0 to every sum (its residual and Jacobian are replaced by
// +0, so every product is +0): adding +0 is exact (a running sum that starts at +0 never becomes -0), so it equals skipping the point.
template <bool HUBER, bool TR>
__device__ __forceinline__ void refw_products(const RefTap& t, float tmpl, const float Jin[6], float huber, float pr[RW_NSUM], float* res_out) {
    // vu_RC: row floor(v) + R, column floor(u) + C
    const float vu_00 = (float)(t.t0 & 0xffu), vu_01 = (float)(TR ? t.t1 & 0xffu : t.t0 >> 8);
    const float vu_10 = (float)(TR ? t.t0 >> 8 : t.t1 & 0xffu), vu_11 = (float)(t.t1 >> 8);
    const float fa = t.fa, fb = t.fb;
    const float im = (1.0f - fb) * (1.0f - fa) * vu_00 + fb * (1.0f - fa) * vu_10 + (1.0f - fb) * fa * vu_01 + fb * fa * vu_11;
    const float r_in = im - tmpl;
    if (res_out) *res_out = t.inside ? r_in : __builtin_nanf("");
    const float r = t.inside ? r_in : 0.f;
    float J[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) J[q] = t.inside ? Jin[q] : 0.f;
    float e, wr, w = 1.0f;
    if (HUBER) {  // extension (oracle: lm_optimizer restatement, eval_energy / compute_eval_data with huber_delta)
        const float ar = fabsf(r);
        const bool quad = ar <= huber;
        e = quad ? r * r : huber * (2.0f * ar - huber);
        w = quad ? 1.0f : huber / ar;
        wr = w * r;
    } else {
        e = r * r;
        wr = r;
    }
    pr[0] = e;
#pragma unroll
    for (int q = 0; q < 6; ++q) pr[1 + q] = J[q] * wr;
    int k = 7;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int s2 = q; s2 < 6; ++s2) {
            const float jj = J[q] * J[s2];
            pr[k++] = HUBER ? w * jj : jj;
        }
}

// Orders this wavefront's LDS stores before its later LDS loads and the other way round. The hardware executes one wavefront's DS
// instructions in order; this only stops the compiler from moving them across (no instruction is emitted).
__device__ __forceinline__ void refw_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float refw_lane(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

// A copy the compiler cannot fold away. refw_eval keeps ONE register set for the record request in flight: it is read (here) at the top of a
// trip, which ends the life of the old value, and requested again a few instructions later. Without the real move the old value lives on
// under the same name (as the record of the group being summed), the new request gets other registers and the copy back into the loop's
// registers — placed at the end of the trip — waits for the request: the pipeline would be gone.
template <class T>
__device__ __forceinline__ T refw_moved(const T& v) {
    static_assert(sizeof(T) % 4 == 0, "dwords");
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "v"(w[i]));
    T out;
    __builtin_memcpy(&out, w, sizeof(T));
    return out;
}

// The 28 chains over the 64 points of `nb` consecutive product blocks (`stride` words apart): lane k < 28 adds row k to its sum, point
// after point — the reference's `energy_sum += ...`, `gradient += ...`, `hessian += ...` (lm_optimizer.rs:80,98-99) for these candidates.
// The additions are one dependent chain (4.6 cycles each); the 16-byte row reads that feed it take ~130 cycles each, so WIN of them are
// kept in flight (a rolling window of 4 WIN registers, running on into the next block; 8 where registers are free, 4 next to a producer): the chain never waits for LDS after the first read.
template <int WIN>
__device__ __forceinline__ void refw_consume(const float* block, int nb, int stride, float& acc) {
    const int lane = threadIdx.x & 63;
    if (lane < RW_NSUM) {
        const float4* row = reinterpret_cast<const float4*>(block + lane * RW_STRIDE);
        if constexpr (WIN == 8) {
            // Two register sets of eight reads (half a block each): the eight reads of the NEXT half are requested in one burst, then the 32
            // additions of the current half run (147 cycles: longer than the LDS latency), so the chain waits for LDS once per call. The
            // scheduler must not regroup this: left alone it requests a half only after the additions before it (522 cycles per block),
            // and one read between every four additions is worse still (608) — the chain pays every instruction issued in between.
            float4 wa[8], wb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) wa[q] = row[q];
            for (int j = 0; j < nb; ++j) {
                const float4* nxt = j + 1 < nb ? row + (stride >> 2) : row;  // (after the last block: eight reads nobody uses)
#pragma unroll
                for (int q = 0; q < 8; ++q) wb[q] = row[8 + q];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    acc = acc + wa[q].x;
                    acc = acc + wa[q].y;
                    acc = acc + wa[q].z;
                    acc = acc + wa[q].w;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) wa[q] = nxt[q];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    acc = acc + wb[q].x;
                    acc = acc + wb[q].y;
                    acc = acc + wb[q].z;
                    acc = acc + wb[q].w;
                }
                __builtin_amdgcn_sched_barrier(0);
                row = nxt;
            }
        } else {
            float4 w[WIN];
#pragma unroll
            for (int q = 0; q < WIN; ++q) w[q] = row[q];
            for (int j = 0; j < nb; ++j) {
                const float4* nxt = j + 1 < nb ? row + (stride >> 2) : row;  // (after the last block: WIN reads nobody uses)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 v4 = w[q % WIN];
                    w[q % WIN] = q + WIN < 16 ? row[q + WIN] : nxt[q + WIN - 16];
                    acc = acc + v4.x;
                    acc = acc + v4.y;
                    acc = acc + v4.z;
                    acc = acc + v4.w;
                }
                row = nxt;
            }
        }
    }
}

#ifdef VORS_REFW_TIMING  // development build (tools/build_ref_variant.sh): shader cycles per phase, summed over all wavefronts
__device__ unsigned long long refc_prof[10];  // workgroup kernel: producer 0 wait+warp, 1 products, 2 at barriers, 3 groups; wavefront 0: 4 chains, 5 verdict + step, 6 at barriers, 7 chunks; 8 kernel cycles, 9 workgroups
__device__ unsigned long long refw_prof[8];  // 0 eval cycles (coop: wavefront 0 summing), 1 step cycles, 2 groups of 64 points (coop: barriers), 3 evaluations,
                                             // 4 kernel cycles, 5 wavefronts, 6 coop: verdict + step + publish, 7 coop: wavefront 0 at barriers
#define REFW_T0(v) const unsigned long long v = __builtin_readcyclecounter()
__device__ unsigned long long refw_lvl_prof[8][3];  // per level (by the power of two of its width): evaluation cycles, groups of 64, evaluations
// (accumulated per wavefront in LDS and flushed once: per-call atomics of 4096 wavefronts in lockstep distort what they measure)
__device__ __forceinline__ unsigned long long* refw_lacc() {
    __shared__ unsigned long long a[8][32];
    return a[threadIdx.x >> 6];
}
#define REFW_ADD(slot, v) do { if ((threadIdx.x & 63) == 0) refw_lacc()[slot] += (unsigned long long)(v); } while (0)
#define REFW_LVL(cols, cyc, groups) do { if ((threadIdx.x & 63) == 0) { unsigned long long* a_ = refw_lacc() + 8 + 3 * min(7, max(0, 27 - __clz(cols))); \
    a_[0] += (unsigned long long)(cyc); a_[1] += (unsigned long long)(groups); a_[2] += 1ull; } } while (0)
#define REFW_INIT() do { if ((threadIdx.x & 63) < 32) refw_lacc()[threadIdx.x & 63] = 0ull; } while (0)
#define REFW_FLUSH() do { const int l_ = threadIdx.x & 63; if (l_ < 32) { const unsigned long long v_ = refw_lacc()[l_]; \
    if (v_) atomicAdd(l_ < 8 ? &refw_prof[l_] : &refw_lvl_prof[(l_ - 8) / 3][(l_ - 8) % 3], v_); } } while (0)
#else
#define REFW_T0(v)
#define REFW_ADD(slot, v)
#define REFW_LVL(cols, cyc, groups)
#define REFW_INIT()
#define REFW_FLUSH()
#endif

// One evaluation — eval_energy + compute_eval_data (lm_optimizer.rs:68-107) — of the n points of `src` at `model` by ONE wavefront,
// every sum in list order. 64 points at a time: each lane evaluates one point and stores its 28 products product-major into the
// wavefront's own LDS rows (28 stores to consecutive words: conflict-free); then lane k < 28 walks row k — 16 ds_read_b128 + 64 dependent
// v_add_f32, nothing else: the chain of additions the reference's loop performs for sum k. Row stride 68 words: lane k starts at bank
// 4k mod 64, so the 16 lanes of each ds_read_b128 group hit 16 disjoint bank quads. No barrier, no second wavefront to wait for.
// Software pipeline over the groups g of 64 points (the loop is a chain of memory round trips otherwise — records, then the taps whose
// addresses depend on them — with nothing else in this wavefront to cover them):
//     warp(g + 1) + its tap requests | record requests of g + 2 | products(g) -> LDS | the 28 chains over group g
// Returns the sums (lane k = sum k); *n_inside = the number of inside points (wavefront-uniform).
template <bool HUBER, class Src>
__device__ __forceinline__ float refw_eval(const Src& src, int n, const RefImg& c, const Iso& model, float* lds, int* n_inside,
                                           float* residuals = nullptr) {
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    int cnt = 0;
    *n_inside = 0;
    if (n <= 0) return acc;
    REFW_T0(t_begin);
    const int ngroups = (n + 63) >> 6;
    typename Src::Raw raw_cur = src.load(min(lane, n - 1)), raw_nxt = raw_cur, raw_ahead = raw_cur;
    RefTap tap_a, tap_b;
    {
        V3 P;
        bool valid;
        src.point(raw_cur, &P, &valid);
        tap_a = refw_warp<Src::kTransposed>(P, valid && lane < n, c, model);
    }
    tap_b = tap_a;
    raw_ahead = src.load(min(64 + lane, n - 1));
    // one group; the taps alternate between two register sets (tap_cur is read, tap_nxt is requested) so that no register move stands between
    // a request and its use a whole group later — a move would wait for the data at the end of every trip
    auto group = [&](int g, const RefTap& tap_cur, RefTap& tap_nxt) {
        // warp(g + 1) and the record request of g + 2, UNCONDITIONALLY (indices clamped to the list, points beyond it marked invalid): with
        // the same requests on every path the compiler's s_waitcnt counting stays exact — one wait per trip, here, for what the PREVIOUS trip
        // requested; a conditional request costs a vmcnt(0) in front of the products of every group. The price is one idle warp at the end.
        raw_nxt = refw_moved(raw_ahead);
        {
            V3 P;
            bool valid;
            src.point(raw_nxt, &P, &valid);
            tap_nxt = refw_warp<Src::kTransposed>(P, valid && (g + 1) * 64 + lane < n, c, model);
        }
        raw_ahead = src.load(min((g + 2) * 64 + lane, n - 1));
        {  // products(g)
            float J[6], tmpl, pr[RW_NSUM], res;
            src.jac(raw_cur, J, &tmpl);
            refw_products<HUBER, Src::kTransposed>(tap_cur, tmpl, J, c.huber, pr, residuals ? &res : nullptr);
            if (residuals && g * 64 + lane < n) residuals[g * 64 + lane] = res;
            cnt += __popcll(__ballot(tap_cur.inside));
#pragma unroll
            for (int k = 0; k < RW_NSUM; ++k) lds[k * RW_STRIDE + lane] = pr[k];
        }
        refw_lds_fence();
        refw_consume<4>(lds, 1, 0, acc);
        refw_lds_fence();
        raw_cur = raw_nxt;  // (already waited for by warp(g + 1))
    };
    for (int g = 0; g < ngroups; g += 2) {
        group(g, tap_a, tap_b);
        if (g + 1 < ngroups) group(g + 1, tap_b, tap_a);
    }
    *n_inside = cnt;
    REFW_ADD(0, __builtin_readcyclecounter() - t_begin);
    REFW_ADD(2, ngroups);
    REFW_ADD(3, 1);
    return acc;
}

// refw_eval with TWO points per lane: 128 points per trip (the lane's points are 64 apart, so each half is one 64-point product block),
// the warp, the interpolation and the Jacobian written for both points at once (lie_g.h: instruction-level parallelism 2 instead of 1 —
// the per-point arithmetic is chains of dependent f32 operations, which on gfx950 issue every 4.6 cycles instead of every 2.6 whatever
// the occupancy). The two product blocks go through the wavefront's one LDS buffer one after the other, each followed by its 64 additions
// per chain — the order of the list. Same pipeline as refw_eval: the tap requests of the next 128 points and the record requests of the
// 128 after those travel while this trip's chains run. Sources with point2 / jac2 only (candidate lists, column-major dense planes).
template <bool HUBER, bool FAST, class Src>
__device__ __forceinline__ float refw_eval2(const Src& src, int n, const RefImg& c, const Iso& model, float* lds, int* n_inside) {
    constexpr bool TR = Src::kTransposed;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    int cnt = 0;
    *n_inside = 0;
    if (n <= 0) return acc;
    REFW_T0(t_begin);
    const int ntrips = (n + 127) >> 7;
    struct Raw2 {
        typename Src::Raw a, b;
    };
    auto load2 = [&](int t) { return Raw2{src.load(min(t * 128 + lane, n - 1)), src.load(min(t * 128 + 64 + lane, n - 1))}; };
    auto warp2 = [&](const Raw2& r, int t) {
        V3G<F2> P;
        bool va, vb;
        src.template point2<FAST>(r.a, r.b, &P, &va, &vb);
        return refw_warp2<TR>(P, va && t * 128 + lane < n, vb && t * 128 + 64 + lane < n, c, model);
    };
    Raw2 raw_cur = load2(0), raw_nxt = raw_cur, raw_ahead = raw_cur;
    RefTap2 tap_a = warp2(raw_cur, 0), tap_b = tap_a;
    raw_ahead = load2(1);
    auto trip = [&](int t, const RefTap2& tap_cur, RefTap2& tap_nxt) {
        raw_nxt = refw_moved(raw_ahead);  // (unconditional requests, clamped indices: see refw_eval)
        tap_nxt = warp2(raw_nxt, t + 1);
#if defined(VORS_REFW_DOUBLE) && (VORS_REFW_DOUBLE & 4)
        { Raw2 r2 = refw_moved(raw_nxt); RefTap2 t2 = warp2(r2, t + 1); if (t2.fa.a == 12345.678f && t2.t0a == 77u && t2.t1b == 78u && t2.t0b == 3u && t2.t1a == 5u) cnt += 1; }
#endif
        raw_ahead = load2(t + 2);
        F2 J[6], tmpl, e, wr, w;
        src.template jac2<FAST>(raw_cur.a, raw_cur.b, J, &tmpl);
        refw_residual2<HUBER, TR>(tap_cur, tmpl, c.huber, J, &e, &wr, &w);
        cnt += __popcll(__ballot(tap_cur.ina)) + __popcll(__ballot(tap_cur.inb));
        {
            float Ja[6], pr[RW_NSUM];
#pragma unroll
            for (int q = 0; q < 6; ++q) Ja[q] = J[q].a;
            refw_products_of<HUBER>(Ja, e.a, wr.a, w.a, pr);
#pragma unroll
            for (int k = 0; k < RW_NSUM; ++k) lds[k * RW_STRIDE + lane] = pr[k];
#if defined(VORS_REFW_DOUBLE) && (VORS_REFW_DOUBLE & 8)
            refw_lds_fence();
            asm volatile("" : "+v"(Ja[0]), "+v"(Ja[1]), "+v"(Ja[2]), "+v"(Ja[3]), "+v"(Ja[4]), "+v"(Ja[5]));
            refw_products_of<HUBER>(Ja, e.a, wr.a, w.a, pr);
#pragma unroll
            for (int k = 0; k < RW_NSUM; ++k) lds[k * RW_STRIDE + lane] = pr[k];
#endif
        }
        refw_lds_fence();
        refw_consume<4>(lds, 1, 0, acc);
#if defined(VORS_REFW_DOUBLE) && (VORS_REFW_DOUBLE & 1)
        { float acc2 = acc; refw_consume<4>(lds, 1, 0, acc2); if (acc2 == 12345.678f) acc += 1.0f; }
#endif
        refw_lds_fence();
        if (t * 128 + 64 < n) {  // (uniform; the second half of the last trip may be empty)
            float Jb[6], pr[RW_NSUM];
#pragma unroll
            for (int q = 0; q < 6; ++q) Jb[q] = J[q].b;
            refw_products_of<HUBER>(Jb, e.b, wr.b, w.b, pr);
#pragma unroll
            for (int k = 0; k < RW_NSUM; ++k) lds[k * RW_STRIDE + lane] = pr[k];
            refw_lds_fence();
            refw_consume<4>(lds, 1, 0, acc);
            refw_lds_fence();
        }
        raw_cur = raw_nxt;
    };
    for (int t = 0; t < ntrips; t += 2) {
        trip(t, tap_a, tap_b);
        if (t + 1 < ntrips) trip(t + 1, tap_b, tap_a);
    }
    *n_inside = cnt;
    REFW_ADD(0, __builtin_readcyclecounter() - t_begin);
    REFW_ADD(2, (n + 63) >> 6);
    REFW_ADD(3, 1);
    REFW_LVL(c.cols, __builtin_readcyclecounter() - t_begin, (n + 63) >> 6);
    return acc;
}

__device__ __forceinline__ float ref_uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ Iso ref_iso_uniform(const Iso& m) {
    return Iso{V3{ref_uniform_f(m.t.x), ref_uniform_f(m.t.y), ref_uniform_f(m.t.z)},
               Quat{ref_uniform_f(m.q.i), ref_uniform_f(m.q.j), ref_uniform_f(m.q.k), ref_uniform_f(m.q.w)}};
}

// step() (lm_optimizer.rs:123-136) on the kept state's sums (lane k of `kept` = sum k), by every lane alike: the result is uniform.
__device__ __forceinline__ bool refw_step(float kept, const Iso& model, float lm_coef, Iso* cand) {
    float h[36], g6[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) g6[q] = refw_lane(kept, 1 + q);
    int k = 7;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int r = q; r < 6; ++r) {
            const float v = refw_lane(kept, k);
            h[q * 6 + r] = v;
            h[r * 6 + q] = v;
            ++k;
        }
    Iso out;
    const bool ok = lm_step(h, g6, model, lm_coef, &out);
    *cand = ref_iso_uniform(out);
    return __builtin_amdgcn_readfirstlane((int)ok) != 0;
}

// optimizer::State::iterative_solve (optimizer.rs:57-70) with init / step / eval / stop_criterion of lm_optimizer.rs:113-192 as a state
// machine driven by evaluations: whoever evaluates `cand` (one wavefront, or a workgroup of them) feeds the sums to ref_lm_advance and is
// told what comes next. Everything is wavefront-uniform except `kept` (lane k = sum k of the kept state).
enum { REF_LM_EVAL = 0, REF_LM_DONE = 1, REF_LM_FAIL = 2 };
struct RefLm {
    Iso cur_model;     // the kept state's model
    Iso cand;          // model to evaluate next (REF_LM_EVAL)
    float kept;        // sums of the kept state
    float cur_energy, lm_coef;
    int nb_iter, n_full;
    bool started;      // the initial evaluation (init: lm_optimizer.rs:113-118) has been fed
};
__device__ __forceinline__ void ref_lm_begin(RefLm& s, const Iso& model) {
    s.cur_model = model;
    s.cand = model;
    s.kept = 0.f;
    s.cur_energy = 0.f;
    s.lm_coef = 0.1f;
    s.nb_iter = 0;
    s.n_full = 0;
    s.started = false;
}
// `acc` / `cnt`: sums and inside count of the evaluation of s.cand. Returns REF_LM_EVAL (evaluate s.cand next), REF_LM_DONE (the level's
// result is s.cur_model) or REF_LM_FAIL (step() failed: "Error at Cholesky decomposition of hessian", lm_optimizer.rs:131-133).
// The candidate a REJECTED evaluation leads to does not depend on that evaluation: step() of the kept state with lm_coef * 10
// (lm_optimizer.rs:164-169). Half of all evaluations are rejected (bench.py: 34 evaluations per pair, 17 of them accepted or initial), so
// whoever has idle time while an evaluation runs (wavefront 0 of the workgroup kernel, during the producers' first chunk) computes it
// ahead — the same step() call on the same inputs, taken instead of computed when the verdict is "rejected".
struct RefLmAhead {
    Iso cand;
    bool ok, valid;
};
__device__ __forceinline__ bool ref_lm_ahead_valid(const RefLm& s) {
    return s.started && !(s.nb_iter > 20);  // (a rejected evaluation with too many iterations stops the level instead)
}
__device__ __forceinline__ RefLmAhead ref_lm_step_ahead(const RefLm& s) {
    RefLmAhead a;
    a.valid = ref_lm_ahead_valid(s);
    a.ok = false;
    a.cand = s.cand;
    if (a.valid) a.ok = refw_step(s.kept, s.cur_model, s.lm_coef * 10.0f, &a.cand);
    return a;
}
__device__ __forceinline__ int ref_lm_advance(RefLm& s, float acc, int cnt, const RefLmAhead* ahead = nullptr) {
    const float energy = refw_lane(acc, 0) / (float)cnt;  // energy_sum / residuals.len(): 0 / 0 = NaN like the reference
    if (!s.started) {  // init: lm_optimizer.rs:113-118
        s.started = true;
        s.kept = acc;
        s.cur_energy = energy;
        s.n_full = 1;
    } else {  // eval() + stop_criterion: lm_optimizer.rs:140-192
        const bool too_many_iterations = s.nb_iter > 20;
        if (energy > s.cur_energy) {  // Err(energy)
            if (too_many_iterations) return REF_LM_DONE;
            s.lm_coef *= 10.0f;
            if (ahead && ahead->valid) {  // step() of exactly this state was computed while the evaluation ran
                s.nb_iter += 1;
                s.cand = ahead->cand;
                return ahead->ok ? REF_LM_EVAL : REF_LM_FAIL;
            }
        } else {
            const float d_energy = s.cur_energy - energy;
            s.n_full += 1;
            s.kept = acc;
            s.cur_energy = energy;
            s.cur_model = s.cand;
            if (too_many_iterations) return REF_LM_DONE;
            s.lm_coef = 0.1f * s.lm_coef;
            if (!(d_energy > 1.0f)) return REF_LM_DONE;
        }
    }
    s.nb_iter += 1;
    REFW_T0(t_step);
    const bool ok = refw_step(s.kept, s.cur_model, s.lm_coef, &s.cand);
#if defined(VORS_REFW_DOUBLE) && (VORS_REFW_DOUBLE & 2)
    { Iso c2; float k2 = refw_moved(s.kept); const bool ok2 = refw_step(k2, s.cur_model, s.lm_coef, &c2); if (ok2 != ok || c2.t.x != s.cand.t.x) s.n_full += 1000; }
#endif
    REFW_ADD(1, __builtin_readcyclecounter() - t_step);
    return ok ? REF_LM_EVAL : REF_LM_FAIL;
}

// One level by one wavefront. G2: two points per lane (refw_eval2; FAST: the level's focal lengths are verified fast divisors).
// The LM state between two evaluations <-> memory (engine.h RefResume): the hand-over of a straggler pair to a workgroup.
__device__ __forceinline__ void ref_lm_save(const RefLm& s, int lvl, RefResume* r) {
    const int lane = threadIdx.x & 63;
    if (lane < RW_NSUM) r->kept[lane] = s.kept;
    if (lane == 0) {
        iso_store(s.cur_model, r->cur_model);
        iso_store(s.cand, r->cand);
        r->cur_energy = s.cur_energy;
        r->lm_coef = s.lm_coef;
        r->nb_iter = s.nb_iter;
        r->n_full = s.n_full;
        r->lvl = lvl;
    }
}
__device__ __forceinline__ void ref_lm_restore(RefLm& s, const RefResume* r) {
    const int lane = threadIdx.x & 63;
    s.cur_model = ref_iso_uniform(iso_load(r->cur_model));
    s.cand = ref_iso_uniform(iso_load(r->cand));
    s.kept = r->kept[min(lane, RW_NSUM - 1)];
    s.cur_energy = ref_uniform_f(r->cur_energy);
    s.lm_coef = ref_uniform_f(r->lm_coef);
    s.nb_iter = __builtin_amdgcn_readfirstlane(r->nb_iter);
    s.n_full = __builtin_amdgcn_readfirstlane(r->n_full);
    s.started = true;
}

// One level by one wavefront. G2: two points per lane (refw_eval2; FAST: the level's focal lengths are verified fast divisors).
// Returns 1 (done), 0 (step() failed) or 2: handed over (ho_after > 0: once that many pairs of the batch have finished, the pair is queued
// for the workgroup kernel at its next evaluation — `done` is read with a returning atomic requested BEFORE the evaluation and looked at
// after it, so the round trip costs nothing).
template <bool HUBER, int G2, class Src>
__device__ int refw_solve_level(const Src& src, int n, const RefImg& c, Iso* model, int* nb_iter_out, float* energy_out, float* lm_coef_out,
                                int* n_full_out, float* lds, const RefHandoff& ho, int ho_after, int pair, int lvl) {
    const int lane = threadIdx.x & 63;
    RefLm s;
    ref_lm_begin(s, *model);
    int cmd;
    do {
        int finished = 0;
        if (ho_after > 0 && lane == 0) finished = atomicAdd(&ho.counters[1], 0);
        int cnt;
        float acc;
        if constexpr (G2 == 2) acc = refw_eval2<HUBER, true>(src, n, c, s.cand, lds, &cnt);
        else if constexpr (G2 == 1) acc = refw_eval2<HUBER, false>(src, n, c, s.cand, lds, &cnt);
        else acc = refw_eval<HUBER>(src, n, c, s.cand, lds, &cnt);
        cmd = ref_lm_advance(s, acc, cnt);
        if (ho_after > 0 && cmd == REF_LM_EVAL && __builtin_amdgcn_readfirstlane(finished) >= ho_after) {
            ref_lm_save(s, lvl, ho.state + pair);
            if (lane == 0) ho.list[atomicAdd(&ho.counters[0], 1)] = pair;
            return 2;
        }
    } while (cmd == REF_LM_EVAL);
    if (cmd == REF_LM_FAIL) return 0;
    *model = s.cur_model;
    *nb_iter_out = s.nb_iter;
    *energy_out = s.cur_energy;
    *lm_coef_out = s.lm_coef;
    *n_full_out = s.n_full;
    return 1;
}

__device__ __forceinline__ JacRecip ref_jac_recip_uniform(const Intr& k) {
    const JacRecip r = g_jac_recip(k);
    return JacRecip{ref_uniform_f(r._fv), ref_uniform_f(r._fuv)};
}

// Where the points of a level come from (template argument SRC of the kernels below).
enum { REF_SRC_SLIM = 0, REF_SRC_DENSE_ROWMAJOR = 1, REF_SRC_DENSE_T = 2 };

template <int SRC>
__device__ __forceinline__ RefImg ref_level_img(const Geom& g, const uint8_t* cur0, const uint8_t* curu, const Records& rec, int pair, int lvl) {
    RefImg c;
    c.img = SRC == REF_SRC_DENSE_T ? level_ptr(g, rec.dense_t.cur0, rec.dense_t.curu, pair, lvl) : level_ptr(g, cur0, curu, pair, lvl);
    c.rows = g.lv[lvl].rows;
    c.cols = g.lv[lvl].cols;
    c.k = g.lv[lvl].k;
    c.huber = g.huber_delta;
    return c;
}
template <int SRC, class F>
__device__ __forceinline__ void ref_with_source(const Geom& g, int lvl, int pair, const uint8_t* kf0, const uint8_t* kfu, const uint16_t* kf_depth,
                                                const Records& rec, F&& f) {
    const LevelGeom lg = g.lv[lvl];
    if constexpr (SRC == REF_SRC_DENSE_T) {
        const RefDensePlanes& t = rec.dense_t;
        RefDenseTSrc src{t.recs + (size_t)pair * ((size_t)g.S0 + g.upper_stride) + (lvl == 0 ? 0 : g.S0 + lg.img_off), lg.rows,
                         0xffffffffu / (unsigned)lg.rows + 1u, lg.k, lg.fu, lg.fv, ref_jac_recip_uniform(lg.k)};
        f(src, lg.rows * lg.cols);
    } else if constexpr (SRC == REF_SRC_DENSE_ROWMAJOR) {
        RefDenseSrc src{&g, kf0, kfu, kf_depth + (size_t)pair * g.S0, lvl > 0 ? rec.IZ + (size_t)pair * g.slots_total + lg.slot_off : nullptr,
                        pair, lvl, lg.rows, lg.cols, lg.k};
        f(src, lg.rows * lg.cols);
    } else {
        RefSlimSrc src{rec.S + (size_t)pair * g.slots_total + lg.slot_off, lg.k, lg.fu, lg.fv, ref_jac_recip_uniform(lg.k)};
        f(src, __builtin_amdgcn_readfirstlane(rec.n_used[(size_t)pair * VORS_MAX_LEVELS + lvl]));
    }
}

// The end of Tracker::track for one pair, by ONE wavefront (`lds`: at least 64 floats of its own): the keyframe test on the coarsest level
// (inverse_compositional.rs:211-224), the pose (inverse_compositional.rs:206-208), the outputs.
template <int SRC>
__device__ __forceinline__ void ref_finish_pair(const Geom& g, int pair, const uint8_t* kf0, const uint8_t* kfu, const uint16_t* kf_depth,
                                                const Records& rec, const Iso& lm_model, bool went_well, const Iso& prev_pose, const Iso& kf_pose,
                                                float* lds, float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats) {
    const int lane = threadIdx.x & 63;
    // keyframe test on the coarsest level (inverse_compositional.rs:211-224): mean L1 displacement, summed in list order (every lane runs
    // the same chain over the 64 displacements its wavefront just left in LDS)
    float flow_sum = 0.f;
    int flow_n = 0;
    {
        const int lvl = g.L - 1;
        const Intr k = g.lv[lvl].k;
        ref_with_source<SRC>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n) {
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                float x, y, iz;
                bool valid;
                src.xy_iz(min(i, n - 1), &x, &y, &iz, &valid);
                float u, v;
                project_uv(k, iso_transform_point(lm_model, back_project(k, x, y, 1.0f / iz)), &u, &v);  // warp: lm_optimizer.rs:213-219
                lds[lane] = (valid && i < n) ? fabsf(x - u) + fabsf(y - v) : -1.0f;  // (a displacement is never negative: -1 = not a candidate)
                refw_lds_fence();
                const float4* row = reinterpret_cast<const float4*>(lds);
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    const float4 v4 = row[q];
                    const float f4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!(f4[j] < 0.f)) {  // (NaN counts as a candidate, like the reference)
                            flow_sum = flow_sum + f4[j];
                            flow_n += 1;
                        }
                }
                refw_lds_fence();
            }
        });
    }
    if (lane == 0) {
        const float optical_flow = flow_sum / (float)flow_n;
        const Iso pose = went_well ? iso_mul(kf_pose, iso_inverse(lm_model)) : prev_pose;  // inverse_compositional.rs:206-208
        iso_store(pose, out_poses7 + 7 * pair);
        out_status[pair] = went_well ? VORS_TRACK_OK : VORS_TRACK_OPTIMIZER_FAILED_POSE_KEPT;
        if (out_stats) {
            iso_store(lm_model, out_stats[pair].lm_model);
            out_stats[pair].optical_flow = optical_flow;
            out_stats[pair].change_keyframe = (optical_flow >= 1.0f) ? 1 : 0;
        }
    }
    if (out_stats) {  // usable candidates per level (diagnostics)
        for (int lvl = 0; lvl < g.L; ++lvl) {
            if constexpr (SRC == REF_SRC_DENSE_T) {
                if (lane == 0) out_stats[pair].n_points[lvl] = rec.dense_t.n_valid[(size_t)pair * VORS_MAX_LEVELS + lvl];
            } else if constexpr (SRC == REF_SRC_DENSE_ROWMAJOR) {
                int mine = 0;
                ref_with_source<SRC>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n) {
                    for (int i = lane; i < n; i += 64) {
                        int x, y;
                        float iz;
                        bool valid;
                        src.raw(i, &x, &y, &iz, &valid);
                        mine += valid ? 1 : 0;
                    }
                });
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o);
                if (lane == 0) out_stats[pair].n_points[lvl] = mine;
            } else {
                if (lane == 0) out_stats[pair].n_points[lvl] = rec.n_used[(size_t)pair * VORS_MAX_LEVELS + lvl];
            }
        }
        if (lane == 0)
            for (int lvl = g.L; lvl < VORS_MAX_LEVELS; ++lvl) {
                out_stats[pair].nb_iter[lvl] = 0;
                out_stats[pair].nb_grad_evals[lvl] = 0;
                out_stats[pair].n_points[lvl] = 0;
                out_stats[pair].energy[lvl] = 0.f;
            }
    }
}

// Tracker::track for a batch (inverse_compositional.rs:177-224): one WAVEFRONT per frame pair, all levels; the wavefronts of a
// workgroup share nothing but the LDS allocation.
template <bool HUBER, int SRC, bool FAST>
__global__ __launch_bounds__(64 * RW_WPB, 4) void lm_ref_track_kernel(Geom g, const uint8_t* __restrict__ cur0, const uint8_t* __restrict__ curu,
                                                                   const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                                   const uint16_t* __restrict__ kf_depth, Records rec,
                                                                   const float* __restrict__ prev_poses7, const float* __restrict__ kf_poses7,
                                                                   float* __restrict__ out_poses7, int32_t* __restrict__ out_status,
                                                                   vors_pair_stats* __restrict__ out_stats, int n_pairs, int ho_after) {
    __shared__ __attribute__((aligned(16))) float lds_all[RW_WPB * RW_WORDS];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int pair = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (pair >= n_pairs) return;
    float* lds = lds_all + wave * RW_WORDS;
    REFW_INIT();
    REFW_T0(t_kernel);
    const Iso prev_pose = prev_poses7 ? iso_load(prev_poses7 + 7 * pair) : iso_identity();
    const Iso kf_pose = kf_poses7 ? iso_load(kf_poses7 + 7 * pair) : iso_identity();
    Iso lm_model = ref_iso_uniform(iso_mul(iso_inverse(prev_pose), kf_pose));  // inverse_compositional.rs:177
    bool went_well = true;
    for (int lvl = g.L - 1; lvl >= 0; --lvl) {
        const RefImg c = ref_level_img<SRC>(g, cur0, curu, rec, pair, lvl);
        int nb_iter = 0, n_full = 0;
        float energy = 0.f, lm_coef = 0.f;
        int how = 0;
        ref_with_source<SRC>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n) {
            // two points per lane wherever the source offers it; FAST: every level's focal lengths are verified fast divisors (host-checked)
            how = refw_solve_level<HUBER, SRC == REF_SRC_DENSE_ROWMAJOR ? 0 : (FAST ? 2 : 1)>(src, n, c, &lm_model, &nb_iter, &energy, &lm_coef, &n_full, lds,
                                                                                    rec.handoff, ho_after, pair, lvl);
        });
        if (how == 2) return;  // handed over: the workgroup kernel finishes this pair (levels done so far have their statistics already)
        const bool ok = how == 1;
        if (out_stats && lane == 0) {
            out_stats[pair].nb_iter[lvl] = ok ? nb_iter : 0;
            out_stats[pair].nb_grad_evals[lvl] = ok ? n_full : 0;
            out_stats[pair].energy[lvl] = ok ? energy : 0.f;
        }
        if (!ok) {
            went_well = false;
            if (out_stats && lane == 0)
                for (int l2 = lvl - 1; l2 >= 0; --l2) {
                    out_stats[pair].nb_iter[l2] = 0;
                    out_stats[pair].nb_grad_evals[l2] = 0;
                    out_stats[pair].energy[l2] = 0.f;
                }
            break;
        }
    }
    REFW_ADD(4, __builtin_readcyclecounter() - t_kernel);
    REFW_ADD(5, 1);
#ifdef VORS_REFW_TIMING
    if (lane == 0) atomicMax(&refw_prof[7], (unsigned long long)(__builtin_readcyclecounter() - t_kernel));  // (slot 7 here: the longest wavefront)
#endif
    REFW_FLUSH();
    if (ho_after > 0 && lane == 0) atomicAdd(&rec.handoff.counters[1], 1);  // one pair fewer to wait for (before the epilogue: the stragglers may go now)
    ref_finish_pair<SRC>(g, pair, kf0, kfu, kf_depth, rec, lm_model, went_well, prev_pose, kf_pose, lds, out_poses7, out_status, out_stats);
}

// ------------------------------------------------------------------------------------------------------------
// The same tracker by a WORKGROUP per frame pair (small batches, the single tracker): wavefront 0 owns the 28 chains and the LM state,
// wavefronts 1 .. P each evaluate every P-th group of 64 points and leave its product block in LDS. Chunks of P blocks, double-buffered:
// while wavefront 0 walks the blocks of chunk ch - 1 in list order (330 cycles of dependent additions per block: the floor of this
// arithmetic), the producers fill chunk ch; one s_barrier per chunk. After the last block wavefront 0 takes the verdict, step() and
// publishes what comes next (a model to evaluate, or the level's result) BEFORE the evaluation's last barrier, so an evaluation costs
// nchunks + 1 barriers and nothing else is exchanged. The sums do not depend on P: the chains are the same chains.
// ------------------------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only. __syncthreads() is a release + acquire over every address space, and on gfx9 loads and
// stores share vmcnt: it would wait for the tap and record requests the producers keep in flight ACROSS the barrier (their pipeline)
// — every chunk would last a full memory round trip. Everything the wavefronts of a pair exchange lives in LDS.
__device__ __forceinline__ void refc_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct RefcShared {
    float model[7];  // REF_LM_EVAL: the model to evaluate next; REF_LM_DONE: the level's result
    int cmd;
    int cnt[2];      // inside points of the evaluation in progress (producers add), alternating per evaluation
};

// One evaluation at `model` by the whole workgroup. Wavefront 0 returns with the sums in `acc` after having called publish(acc, n_inside)
// before the last barrier; the producers return after that barrier.
#ifdef VORS_REFW_TIMING  // (timing build: cycles accumulated in registers, one atomic per wavefront and bucket at the end of the kernel)
#define REFC_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define REFC_ACC(slot, v) prof[slot] += (unsigned long long)(v)
#else
#define REFC_T(v)
#define REFC_ACC(slot, v)
#endif
// The records of a producer's first two groups do not change from one evaluation of a level to the next: requested once per level and kept
// in registers, every later evaluation of the level starts with its warp instead of a memory round trip.
template <class Src>
struct RefcFirst {
    typename Src::Raw r0, r1;
    bool have;
};
template <bool HUBER, class Src, class Publish, class Ahead>
__device__ __forceinline__ void refc_eval(const Src& src, int n, const RefImg& c, const Iso& model, float* slots, RefcShared& sh, int P, int parity,
                                          RefcFirst<Src>& first, Publish&& publish, Ahead&& ahead, unsigned long long* prof) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int ngroups = (n + 63) >> 6, nchunks = (ngroups + P - 1) / P;
    const int p = wave - 1;  // producer index
    float acc = 0.f;
    int cnt = 0;
    typename Src::Raw raw_cur = {}, raw_nxt = {}, raw_ahead = {};
    RefTap tap_a = {}, tap_b = {};
    if (wave != 0 && n > 0) {  // the producer's software pipeline over ITS groups p, p + P, ... (refw_eval)
        if (!first.have) {
            first.r0 = src.load(min(p * 64 + lane, n - 1));
            first.r1 = src.load(min((p + P) * 64 + lane, n - 1));
            first.have = true;
        }
        raw_cur = first.r0;
        V3 Pt;
        bool valid;
        src.point(raw_cur, &Pt, &valid);
        tap_a = refw_warp<Src::kTransposed>(Pt, valid && p * 64 + lane < n, c, model);
        tap_b = tap_a;
        raw_nxt = raw_cur;
        raw_ahead = first.r1;
    }
    auto chunk = [&](int ch, const RefTap& tap_cur, RefTap& tap_nxt) {
        if (wave != 0) {
            if (ch < nchunks) {
                REFC_T(tp0);
                const int g = ch * P + p;
                raw_nxt = refw_moved(raw_ahead);
                {
                    V3 Pt;
                    bool valid;
                    src.point(raw_nxt, &Pt, &valid);
                    tap_nxt = refw_warp<Src::kTransposed>(Pt, valid && (g + P) * 64 + lane < n, c, model);
                }
                raw_ahead = src.load(min((g + 2 * P) * 64 + lane, n - 1));
                REFC_T(tp1);
                float J[6], tmpl, pr[RW_NSUM];
                src.jac(raw_cur, J, &tmpl);
                refw_products<HUBER, Src::kTransposed>(tap_cur, tmpl, J, c.huber, pr, nullptr);
                cnt += __popcll(__ballot(tap_cur.inside));
                float* block = slots + ((ch & 1) * P + p) * RW_WORDS;
#pragma unroll
                for (int k = 0; k < RW_NSUM; ++k) block[k * RW_STRIDE + lane] = pr[k];
                raw_cur = raw_nxt;
                if (ch == nchunks - 1 && lane == 0 && cnt != 0) atomicAdd(&sh.cnt[parity], cnt);
                REFC_T(tp2);
                REFC_ACC(0, tp1 - tp0);  // wait for the previous trip's requests + warp + new requests
                REFC_ACC(1, tp2 - tp1);  // Jacobian, interpolation, products, LDS stores
                REFC_ACC(3, 1);
            }
        } else {
            REFC_T(t_c0);
            if (ch == 0) ahead();  // nothing to sum yet: the producers are on their first chunk
            if (ch > 0) refw_consume<8>(slots + ((ch - 1) & 1) * P * RW_WORDS, min(P, ngroups - (ch - 1) * P), RW_WORDS, acc);
            REFC_T(t_c1);
            if (ch == nchunks) publish(acc, sh.cnt[parity]);  // (the producers' counts arrived before the previous barrier)
            REFC_T(t_c2);
            REFC_ACC(4, t_c1 - t_c0);
            REFC_ACC(5, t_c2 - t_c1);
            REFC_ACC(7, 1);
        }
        REFC_T(t_b0);
        refc_barrier();
        REFC_T(t_b1);
        REFC_ACC(wave != 0 ? 2 : 6, t_b1 - t_b0);
    };
    for (int ch = 0; ch <= nchunks; ch += 2) {
        chunk(ch, tap_a, tap_b);
        if (ch + 1 <= nchunks) chunk(ch + 1, tap_b, tap_a);
    }
}

// AHEAD: wavefront 0 computes the rejected-case candidate while the producers are on an evaluation's first chunk (ref_lm_step_ahead). Its
// second copy of step() costs ~14 registers — one wavefront slot per SIMD, which the mid-sized batches (two workgroups of five wavefronts per
// CU) need and the small ones (one workgroup of eight per CU) do not: measured 128 pairs 0.397 -> 0.386 ms, one tracker 0.335 -> 0.312 ms
// per frame, but 512 pairs 0.477 -> 0.74 ms when it is on everywhere.
template <bool HUBER, int SRC, bool AHEAD>
__global__ __launch_bounds__(512) void lm_ref_track_coop_kernel(Geom g, const uint8_t* __restrict__ cur0, const uint8_t* __restrict__ curu,
                                                                const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                                const uint16_t* __restrict__ kf_depth, Records rec,
                                                                const float* __restrict__ prev_poses7, const float* __restrict__ kf_poses7,
                                                                float* __restrict__ out_poses7, int32_t* __restrict__ out_status,
                                                                vors_pair_stats* __restrict__ out_stats, int n_pairs, int resume) {
    extern __shared__ __attribute__((aligned(16))) float refc_lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int P = (int)(blockDim.x >> 6) - 1;
    float* slots = refc_lds;
    RefcShared& sh = *reinterpret_cast<RefcShared*>(refc_lds + 2 * P * RW_WORDS);
    // resume: the workgroups serve the pairs the one-wavefront launch queued (engine.h RefHandoff); the others have nothing to do
    if (resume && (int)blockIdx.x >= rec.handoff.counters[0]) return;
    const int pair = resume ? __builtin_amdgcn_readfirstlane(rec.handoff.list[blockIdx.x]) : (int)blockIdx.x;
    const RefResume* saved = resume ? rec.handoff.state + pair : nullptr;
    const int first_lvl = resume ? __builtin_amdgcn_readfirstlane(saved->lvl) : g.L - 1;
    REFW_T0(t_kernel);
    REFW_INIT();
    if (threadIdx.x == 0) {
        sh.cnt[0] = 0;
        sh.cnt[1] = 0;
    }
    __syncthreads();
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const Iso prev_pose = prev_poses7 ? iso_load(prev_poses7 + 7 * pair) : iso_identity();
    const Iso kf_pose = kf_poses7 ? iso_load(kf_poses7 + 7 * pair) : iso_identity();
    Iso lm_model = ref_iso_uniform(iso_mul(iso_inverse(prev_pose), kf_pose));  // inverse_compositional.rs:177 (every wavefront alike)
    bool went_well = true;
    int parity = 0;
    for (int lvl = first_lvl; lvl >= 0; --lvl) {
        const RefImg c = ref_level_img<SRC>(g, cur0, curu, rec, pair, lvl);
        RefLm s;
        ref_lm_begin(s, lm_model);
        if (resume && lvl == first_lvl) {  // pick the level up where the one-wavefront launch left it: `cand` is the next model to evaluate
            ref_lm_restore(s, saved);
            lm_model = s.cand;
        }
        int cmd = REF_LM_EVAL;
        ref_with_source<SRC>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n) {
            Iso eval_model = lm_model;
            RefcFirst<std::decay_t<decltype(src)>> first;
            first.have = false;
            do {
                RefLmAhead ahead;
                ahead.valid = false;
                refc_eval<HUBER>(src, n, c, eval_model, slots, sh, P, parity, first, [&](float acc, int cnt) {
                    const int next = ref_lm_advance(s, acc, cnt, &ahead);
                    if (lane == 0) {
                        iso_store(next == REF_LM_EVAL ? s.cand : s.cur_model, sh.model);
                        sh.cmd = next;
                        sh.cnt[parity ^ 1] = 0;
                    }
                }, [&]() { if constexpr (AHEAD) ahead = ref_lm_step_ahead(s); }, prof);
                parity ^= 1;
                cmd = __builtin_amdgcn_readfirstlane(sh.cmd);
                eval_model = ref_iso_uniform(iso_load(sh.model));
            } while (cmd == REF_LM_EVAL);
            lm_model = eval_model;  // REF_LM_DONE: the level's result; REF_LM_FAIL: the last kept model (not used any further)
        });
        const bool ok = cmd == REF_LM_DONE;
        if (wave == 0) lm_model = ok ? s.cur_model : lm_model;
        if (out_stats && threadIdx.x == 0) {
            out_stats[pair].nb_iter[lvl] = ok ? s.nb_iter : 0;
            out_stats[pair].nb_grad_evals[lvl] = ok ? s.n_full : 0;
            out_stats[pair].energy[lvl] = ok ? s.cur_energy : 0.f;
        }
        if (!ok) {
            went_well = false;
            if (out_stats && threadIdx.x == 0)
                for (int l2 = lvl - 1; l2 >= 0; --l2) {
                    out_stats[pair].nb_iter[l2] = 0;
                    out_stats[pair].nb_grad_evals[l2] = 0;
                    out_stats[pair].energy[l2] = 0.f;
                }
            break;
        }
    }
#ifdef VORS_REFW_TIMING
    if (lane == 0 && wave <= 1)  // wavefront 0 (buckets 4..7) and ONE producer (buckets 0..3)
        for (int k = (wave ? 0 : 4); k < (wave ? 4 : 8); ++k) atomicAdd(&refc_prof[k], prof[k]);
    if (threadIdx.x == 0) {
        atomicAdd(&refc_prof[8], (unsigned long long)(__builtin_readcyclecounter() - t_kernel));
        atomicAdd(&refc_prof[9], 1ull);
    }
    REFW_FLUSH();
#endif
    (void)prof;
    if (wave != 0) return;  // (no barrier below)
    ref_finish_pair<SRC>(g, pair, kf0, kfu, kf_depth, rec, lm_model, went_well, prev_pose, kf_pose, slots, out_poses7, out_status, out_stats);
}

// Wavefronts per workgroup of the one-wavefront-per-pair kernel: four pairs to a workgroup (one wavefront per SIMD: single-wavefront
// workgroups pile up on the SIMDs unevenly — measured: 2.84 ms instead of 1.68 ms per 4096 coarse-to-fine pairs). VORS_REF_WPB overrides (1, 2 or 4).
static int refw_waves_per_block(int n_pairs) {
    int wpb = RW_WPB;
    if (const char* e = getenv("VORS_REF_WPB")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4) wpb = v;
    }
    return wpb;
}
// Wavefronts per PAIR: a batch that cannot fill the chip with one wavefront per pair gets a workgroup per pair (2 .. 8 wavefronts: one owns
// the chains, the others produce; LDS 2 x P x 7.6 KB). VORS_REF_COOP overrides (0 = one wavefront per pair, 2 .. 8). Measured at 512
// pairs (3 / 4 / 5 / 6 wavefronts): coarse-to-fine 0.62 / 0.53 / 0.50 / 0.51 ms, DSO 1.08 / 0.86 / 0.81 / 0.84, dense 14.7 / 10.3 / 9.3 / 9.8.
// Wavefronts per pair of the workgroup kernel: the LARGEST workgroup of {8, 5, 4, 3} wavefronts of which the whole batch is resident at once
// (per CU: 16 wavefronts, 160 KB of LDS at 2 (W - 1) blocks of 7.6 KB per workgroup — MI355X: 256 pairs with 8, 512 with 5, 768 with 4,
// 1280 with 3); a batch that needs a second round of workgroups is slower than a thinner workgroup for everyone (tools/coop_sweep.py: 320
// pairs 0.83 ms per step with 8 wavefronts, 0.65 with 5; 1024 pairs 1.52 with 5, 1.22 with 3). Beyond that: one wavefront per pair.
// What the CURRENT device offers a workgroup kernel: compute units and LDS per CU (queried per call — cheap, cached by the runtime — because a
// process may hold handles on devices of different sizes; MI355X: 256 CUs, 160 KB).
struct RefcDevice {
    int cus;
    size_t lds_per_cu;
};
static RefcDevice refc_device() {
    int dev = 0, cus = 0, lds = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds <= 0) lds = 64 * 1024;
    int lds_cu = 0;  // (some runtimes report the 64 KB default per block and the real size per multiprocessor: a workgroup may take all of it)
    if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && lds_cu > lds) lds = lds_cu;
    (void)hipGetLastError();
    return RefcDevice{cus, (size_t)lds};
}
static size_t refc_lds_bytes(int waves) { return (size_t)2 * (waves - 1) * RW_WORDS * sizeof(float) + sizeof(RefcShared); }
static int refc_waves_per_pair(int n_pairs, bool dense, int n_launch = -1) {  // n_pairs: pairs resident together; n_launch: pairs of this launch
    if (n_launch < 0) n_launch = n_pairs;
    const RefcDevice d = refc_device();
    int w = 0;
    for (int cand : {8, 5, 4, 3}) {
        const size_t lds = refc_lds_bytes(cand);
        if (lds > d.lds_per_cu) continue;  // (a device with less LDS than gfx950: thinner workgroups)
        const long long per_cu = std::min<long long>(16 / cand, (long long)(d.lds_per_cu / lds));
        if (per_cu * d.cus >= n_pairs) {
            w = cand;
            break;
        }
    }
    // dense: two producers do not keep up with the chains of 409,600 points (1024 pairs: 20.5 ms per step with 3 wavefronts, 18.9 with 5 in
    // two rounds of workgroups) — five up to 1280 pairs as before
    if (dense && w < 4 && n_launch <= 5 * d.cus && refc_lds_bytes(5) <= d.lds_per_cu) w = 5;
    if (const char* e = getenv("VORS_REF_COOP")) {
        const int v = atoi(e);
        if (v == 0 || (v >= 2 && v <= 8 && refc_lds_bytes(v) <= d.lds_per_cu)) w = v;
    }
    return w;
}

void launch_lm_track_reference(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7,
                               const float* kf_poses7, float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs,
                               hipStream_t s) {
    const bool huber = g.huber_delta > 0.f;
    // dense mode: the column-major planes (capi.cpp allocates and fills them for every REFERENCE handle); without them, the gathering source
    const int src = g.mode != VORS_CANDIDATES_DENSE ? REF_SRC_SLIM : (rec.dense_t.recs ? REF_SRC_DENSE_T : REF_SRC_DENSE_ROWMAJOR);
    // (a slot of a vors_pipeline ring: the workgroup size for the pairs that are resident TOGETHER — engine.h Geom::ref_inflight_x2. Measured at
    // 512 pairs per step through a ring of 3: coarse-to-fine 0.556 -> 0.462 ms per step, DSO 0.889 -> 0.754, dense 6.18 -> 5.53 with 4
    // wavefronts per pair instead of the lone step's 5)
    const bool dense = g.mode == VORS_CANDIDATES_DENSE;
    const int resident = (int)std::min<long long>((long long)n_pairs * std::max(2, g.ref_inflight_x2) / 2, 1 << 30);
    const int coop = refc_waves_per_pair(resident, dense, n_pairs);
#define VORS_REF_DISPATCH(KERNEL)                                                                              \
    do {                                                                                                       \
        if (src == REF_SRC_DENSE_T) {                                                                          \
            if (huber) VORS_REF_LAUNCH((KERNEL<true, REF_SRC_DENSE_T>));                                       \
            else VORS_REF_LAUNCH((KERNEL<false, REF_SRC_DENSE_T>));                                            \
        } else if (src == REF_SRC_DENSE_ROWMAJOR) {                                                            \
            if (huber) VORS_REF_LAUNCH((KERNEL<true, REF_SRC_DENSE_ROWMAJOR>));                                \
            else VORS_REF_LAUNCH((KERNEL<false, REF_SRC_DENSE_ROWMAJOR>));                                     \
        } else {                                                                                               \
            if (huber) VORS_REF_LAUNCH((KERNEL<true, REF_SRC_SLIM>));                                          \
            else VORS_REF_LAUNCH((KERNEL<false, REF_SRC_SLIM>));                                               \
        }                                                                                                      \
    } while (0)
#define VORS_REF_DISPATCH_COOP(AH)                                                                              \
    do {                                                                                                       \
        if (src == REF_SRC_DENSE_T) {                                                                          \
            if (huber) VORS_REF_LAUNCH((lm_ref_track_coop_kernel<true, REF_SRC_DENSE_T, AH>));                 \
            else VORS_REF_LAUNCH((lm_ref_track_coop_kernel<false, REF_SRC_DENSE_T, AH>));                      \
        } else if (src == REF_SRC_DENSE_ROWMAJOR) {                                                            \
            if (huber) VORS_REF_LAUNCH((lm_ref_track_coop_kernel<true, REF_SRC_DENSE_ROWMAJOR, AH>));          \
            else VORS_REF_LAUNCH((lm_ref_track_coop_kernel<false, REF_SRC_DENSE_ROWMAJOR, AH>));               \
        } else {                                                                                               \
            if (huber) VORS_REF_LAUNCH((lm_ref_track_coop_kernel<true, REF_SRC_SLIM, AH>));                    \
            else VORS_REF_LAUNCH((lm_ref_track_coop_kernel<false, REF_SRC_SLIM, AH>));                         \
        }                                                                                                      \
    } while (0)
    if (coop) {
        const size_t lds = (size_t)2 * (coop - 1) * RW_WORDS * sizeof(float) + sizeof(RefcShared);
        // (a launch may ask for more than 64 KB of dynamic LDS only after the kernel has been told so; per device, hence not cached in a static)
#define VORS_REF_LAUNCH(K)                                                                                                                  \
    do {                                                                                                                                    \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            break; /* (the error stays pending: the caller's hipGetLastError reports it against this launch) */ \
        hipLaunchKernelGGL(K, dim3(n_pairs), dim3(64 * coop), lds, s, g, cur.level0, cur.upper, kf.level0, kf.upper, kf_depth, rec, prev_poses7, \
                           kf_poses7, out_poses7, out_status, out_stats, n_pairs, 0);                                                      \
    } while (0)
        if (coop >= 8) VORS_REF_DISPATCH_COOP(true);
        else VORS_REF_DISPATCH_COOP(false);
#undef VORS_REF_LAUNCH
        return;
    }
    const int wpb = refw_waves_per_block(n_pairs);
    // straggler hand-over (engine.h RefHandoff): once this share of the batch has finished, the pairs still iterating move to workgroups of
    // their own. VORS_REF_HANDOFF = percent (0 = off); batches that fill the chip only.
    int ho_after = 0;
    int ho_min_pairs = 2048;  // VORS_REF_HANDOFF_MIN_PAIRS: tests hand small batches over
    if (const char* e = getenv("VORS_REF_HANDOFF_MIN_PAIRS")) ho_min_pairs = std::max(1, atoi(e));
    if (rec.handoff.state && n_pairs >= ho_min_pairs) {
        int percent = 65;
        if (const char* e = getenv("VORS_REF_HANDOFF")) percent = std::max(0, std::min(99, atoi(e)));
        ho_after = (int)((long long)n_pairs * percent / 100);
    }
    if (ho_after > 0) (void)hipMemsetAsync(rec.handoff.counters, 0, 2 * sizeof(int), s);
#define VORS_REF_LAUNCH(K)                                                                                                                            \
    hipLaunchKernelGGL(K, dim3((n_pairs + wpb - 1) / wpb), dim3(64 * wpb), 0, s, g, cur.level0, cur.upper, kf.level0, kf.upper, kf_depth, rec, prev_poses7, \
                       kf_poses7, out_poses7, out_status, out_stats, n_pairs, ho_after)
    bool all_fast = true;
    for (int l = 0; l < g.L; ++l) all_fast = all_fast && g.lv[l].fu.ok && g.lv[l].fv.ok;
    if (src == REF_SRC_DENSE_ROWMAJOR) {
        if (huber) VORS_REF_LAUNCH((lm_ref_track_kernel<true, REF_SRC_DENSE_ROWMAJOR, false>));
        else VORS_REF_LAUNCH((lm_ref_track_kernel<false, REF_SRC_DENSE_ROWMAJOR, false>));
    } else if (src == REF_SRC_DENSE_T) {
        if (huber && all_fast) VORS_REF_LAUNCH((lm_ref_track_kernel<true, REF_SRC_DENSE_T, true>));
        else if (huber) VORS_REF_LAUNCH((lm_ref_track_kernel<true, REF_SRC_DENSE_T, false>));
        else if (all_fast) VORS_REF_LAUNCH((lm_ref_track_kernel<false, REF_SRC_DENSE_T, true>));
        else VORS_REF_LAUNCH((lm_ref_track_kernel<false, REF_SRC_DENSE_T, false>));
    } else {
        if (huber && all_fast) VORS_REF_LAUNCH((lm_ref_track_kernel<true, REF_SRC_SLIM, true>));
        else if (huber) VORS_REF_LAUNCH((lm_ref_track_kernel<true, REF_SRC_SLIM, false>));
        else if (all_fast) VORS_REF_LAUNCH((lm_ref_track_kernel<false, REF_SRC_SLIM, true>));
        else VORS_REF_LAUNCH((lm_ref_track_kernel<false, REF_SRC_SLIM, false>));
    }
#undef VORS_REF_LAUNCH
    if (ho_after > 0) {  // the queued pairs, a workgroup each (at most n_pairs - ho_after of them; a workgroup beyond the queue returns at once)
        int hw = 4;
        if (const char* e = getenv("VORS_REF_HANDOFF_WAVES")) {
            const int v = atoi(e);
            if (v >= 2 && v <= 8) hw = v;
        }
        const size_t lds = (size_t)2 * (hw - 1) * RW_WORDS * sizeof(float) + sizeof(RefcShared);
#define VORS_REF_LAUNCH(K)                                                                                                                  \
    do {                                                                                                                                    \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            break; \
        hipLaunchKernelGGL(K, dim3(n_pairs - ho_after), dim3(64 * hw), lds, s, g, cur.level0, cur.upper, kf.level0, kf.upper, kf_depth, rec, prev_poses7, \
                           kf_poses7, out_poses7, out_status, out_stats, n_pairs, 1);                                                      \
    } while (0)
        VORS_REF_DISPATCH_COOP(false);
#undef VORS_REF_LAUNCH
    }
}

#ifdef VORS_REFW_TIMING
}  // namespace vors
extern "C" int vors_debug_refc_profile(unsigned long long out[10], int reset) {  // development build only
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vors::refc_prof), 10 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(vors::refc_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int vors_debug_refw_level_profile(unsigned long long out[24], int reset) {  // development build only
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vors::refw_lvl_prof), 24 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[24] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(vors::refw_lvl_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int vors_debug_refw_profile(unsigned long long out[8], int reset) {  // development build only
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vors::refw_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(vors::refw_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
namespace vors {
#endif

// sums of one wavefront (lane k = sum k) + the inside count -> the 29-float layout of the C ABI: [0] = sum r^2, [1] = n_inside,
// [2..7] = g, [8..28] = H upper triangle row-wise
__device__ __forceinline__ void refw_store29(float acc, int cnt, float* out29) {
    const int lane = threadIdx.x & 63;
    if (lane < RW_NSUM) out29[lane == 0 ? 0 : lane + 1] = acc;
    if (lane == 0) out29[1] = (float)cnt;
}

// One evaluation of one level of one pair of a prepared batch (vors_batch_eval_level in the REFERENCE arithmetic) -> 29 sums.
template <bool HUBER, int SRC>
__global__ __launch_bounds__(64) void lm_ref_eval_level_kernel(Geom g, const uint8_t* __restrict__ cur0, const uint8_t* __restrict__ curu,
                                                               const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                               const uint16_t* __restrict__ kf_depth, Records rec, int pair, int lvl,
                                                               const float* __restrict__ model7, float* __restrict__ out29) {
    __shared__ __attribute__((aligned(16))) float lds[RW_WORDS];
    const Iso model = ref_iso_uniform(iso_load(model7));
    const RefImg c = ref_level_img<SRC>(g, cur0, curu, rec, pair, lvl);
    ref_with_source<SRC>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n) {
        int cnt;
        const float acc = refw_eval<HUBER>(src, n, c, model, lds, &cnt);
        refw_store29(acc, cnt, out29);
    });
}
void launch_lm_eval_level_reference(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, int pair, int lvl,
                                    const float* model7, float* out29, hipStream_t s) {
    const bool huber = g.huber_delta > 0.f;
    const int src = g.mode != VORS_CANDIDATES_DENSE ? REF_SRC_SLIM : (rec.dense_t.recs ? REF_SRC_DENSE_T : REF_SRC_DENSE_ROWMAJOR);
#define VORS_REF_LAUNCH(K) hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, s, g, cur.level0, cur.upper, kf.level0, kf.upper, kf_depth, rec, pair, lvl, model7, out29)
    VORS_REF_DISPATCH(lm_ref_eval_level_kernel);
#undef VORS_REF_LAUNCH
#undef VORS_REF_DISPATCH
#undef VORS_REF_DISPATCH_COOP
}

// ------------------------------------------------------------------------------------------------------------
// Dense mode: the column-major planes (engine.h RefDensePlanes). One tile kernel for bytes, 16-bit depths and f32 inverse depths:
// a TILE x TILE block goes through LDS, read along the source rows, written along the destination rows (= source columns).
// ------------------------------------------------------------------------------------------------------------
template <class T, int TILE>
__global__ __launch_bounds__(256) void ref_transpose_kernel(Geom g, const T* __restrict__ src, size_t src_stride, T* __restrict__ dst, size_t dst_stride,
                                                            int rows, int cols) {
    __shared__ T tile[TILE][TILE + 1];
    const int pair = select_pair(g, blockIdx.z);
    if (pair < 0) return;
    const T* sp = src + (size_t)pair * src_stride;
    T* dp = dst + (size_t)pair * dst_stride;
    constexpr int TY = 256 / TILE;
    const int tx = threadIdx.x % TILE, ty = threadIdx.x / TILE;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    for (int j = ty; j < TILE; j += TY)
        if (y0 + j < rows && x0 + tx < cols) tile[j][tx] = sp[(size_t)(y0 + j) * cols + x0 + tx];
    __syncthreads();
    for (int j = ty; j < TILE; j += TY)  // destination row x0 + j (a source column), TILE consecutive source rows of it
        if (x0 + j < cols && y0 + tx < rows) dp[(size_t)(x0 + j) * rows + y0 + tx] = tile[tx][j];
}
template <class T>
static void ref_transpose(const Geom& g, const T* src, size_t src_stride, T* dst, size_t dst_stride, int rows, int cols, int n_pairs, hipStream_t s) {
    constexpr int TILE = sizeof(T) == 1 ? 64 : 32;
    hipLaunchKernelGGL((ref_transpose_kernel<T, TILE>), dim3((cols + TILE - 1) / TILE, (rows + TILE - 1) / TILE, n_pairs), dim3(256), 0, s, g, src,
                       src_stride, dst, dst_stride, rows, cols);
}
// 8-byte store of data nobody re-reads soon (13 GB of records per 4096 pairs: written once per keyframe, read by the LM stage later):
// VORS_REF_NT_STORES picks the non-temporal form at compile time (A/B: tools/build_ref_variant.sh).
__device__ __forceinline__ void ref_store_stream(uint2* p, uint2 v) {
#ifdef VORS_REF_NT_STORES
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
#else
    *p = v;
#endif
}
// The keyframe side in one kernel per level: gradient + template of the pixel (device_common.h grad_tmpl_at: gradient.rs:15-33,74-93) and its
// inverse depth (level 0: scale / depth, inverse_depth.rs:24-29; above: the fused plane) as ONE 8-byte record, read along the rows of the
// row-major sources, written along the columns (32 x 64 tile through LDS).
__global__ __launch_bounds__(256) void ref_dense_records_kernel(Geom g, const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                                const uint16_t* __restrict__ depth, Records rec, int l) {
    __shared__ uint2 tile[32][65];  // 32 rows x 64 columns: a wavefront reads one row of 64 pixels, then writes 32 rows of two columns
    const int pair = select_pair(g, blockIdx.z);
    if (pair < 0) return;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 32;
    const uint16_t* dp = depth + (size_t)pair * g.S0;
    const float* izp = l > 0 ? rec.IZ + (size_t)pair * g.slots_total + g.lv[l].slot_off : nullptr;
    int n_valid = 0;
#pragma unroll 2
    for (int j = wave; j < 32; j += 4) {
        const int x = x0 + lane, y = y0 + j;
        if (x < cols && y < rows) {
            int gx, gy, tm;
            grad_tmpl_at(g, kf0, kfu, pair, l, x, y, &gx, &gy, &tm);
            float iz;
            bool valid;
            if (l == 0) {
                const int dz = dp[(unsigned)(y * cols + x)];
                valid = dz != 0;
                iz = g.depth_scale / (float)dz;
            } else {
                iz = izp[(unsigned)(y * cols + x)];
                valid = !(iz != iz);
            }
            tile[j][lane] = make_uint2(__float_as_uint(iz), slim_pack_tg(tm, gx, gy) | (valid ? 0x80000000u : 0u));
            n_valid += valid ? 1 : 0;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n_valid += __shfl_xor(n_valid, o);
    if (lane == 0 && n_valid != 0) atomicAdd(&rec.dense_t.n_valid[(size_t)pair * VORS_MAX_LEVELS + l], n_valid);
    __syncthreads();
    uint2* out = rec.dense_t.recs + (size_t)pair * ((size_t)g.S0 + g.upper_stride) + (l == 0 ? 0 : g.S0 + g.lv[l].img_off);
    const int r = lane & 31, c2 = lane >> 5;
#pragma unroll 2
    for (int j = 2 * wave + c2; j < 64; j += 8)  // column x0 + j, 32 consecutive rows of it (256 bytes per half-wavefront)
        if (x0 + j < cols && y0 + r < rows) ref_store_stream(&out[(size_t)(x0 + j) * rows + y0 + r], tile[r][j]);
}
// Level 0 (three quarters of the pixels) when the width is a multiple of 4 and the buffers are 8-byte aligned: FOUR pixels per thread —
// the rows above, at and below as dwords, the two horizontal neighbours as bytes, the four depths as one 8-byte load: 6 requests per 4
// pixels instead of 24 (the generic kernel is bound by its request count, not by the 8 bytes it writes per pixel). 128 x 32 tiles.
__global__ __launch_bounds__(256) void ref_dense_records_level0_kernel(Geom g, const uint8_t* __restrict__ kf0, const uint16_t* __restrict__ depth, Records rec) {
    __shared__ uint2 tile[32][129];
    const int pair = select_pair(g, blockIdx.z);
    if (pair < 0) return;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 128, y0 = blockIdx.y * 32;
    const uint8_t* img = kf0 + (size_t)pair * g.S0;
    const uint16_t* dp = depth + (size_t)pair * g.S0;
    int n_valid = 0;
    const int xq = lane & 31, jr = lane >> 5;  // 32 threads x 4 pixels = one 128-pixel row; a wavefront = two rows
#pragma unroll 2
    for (int j = 2 * wave + jr; j < 32; j += 8) {
        const int x = x0 + 4 * xq, y = y0 + j;
        if (x < cols && y < rows) {
            const unsigned o = (unsigned)(y * cols + x);
            const unsigned up = y > 0 ? o - (unsigned)cols : o, dn = y < rows - 1 ? o + (unsigned)cols : o;
            const uint32_t c4 = *reinterpret_cast<const uint32_t*>(img + o);
            const uint32_t u4 = *reinterpret_cast<const uint32_t*>(img + up), d4 = *reinterpret_cast<const uint32_t*>(img + dn);
            const int lft = img[x > 0 ? o - 1u : o], rgt = img[x + 4 < cols ? o + 4u : o + 3u];
            const uint2 z4 = *reinterpret_cast<const uint2*>(dp + o);
            const int p[6] = {lft, (int)(c4 & 0xffu), (int)((c4 >> 8) & 0xffu), (int)((c4 >> 16) & 0xffu), (int)(c4 >> 24), rgt};
            const bool row_border = y == 0 || y == rows - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool border = row_border || x + i == 0 || x + i == cols - 1;  // 1-px border = 0 (gradient.rs:15-33)
                const int gx = border ? 0 : (p[i + 2] - p[i]) / 2;
                const int gy = border ? 0 : ((int)((d4 >> (8 * i)) & 0xffu) - (int)((u4 >> (8 * i)) & 0xffu)) / 2;
                const int dz = (int)(((i < 2 ? z4.x : z4.y) >> (16 * (i & 1))) & 0xffffu);
                const bool valid = dz != 0;
                const float iz = g.depth_scale / (float)dz;  // inverse_depth.rs:24-29
                tile[j][4 * xq + i] = make_uint2(__float_as_uint(iz), slim_pack_tg(p[i + 1], gx, gy) | (valid ? 0x80000000u : 0u));
                n_valid += valid ? 1 : 0;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n_valid += __shfl_xor(n_valid, o);
    if (lane == 0 && n_valid != 0) atomicAdd(&rec.dense_t.n_valid[(size_t)pair * VORS_MAX_LEVELS], n_valid);
    __syncthreads();
    uint2* out = rec.dense_t.recs + (size_t)pair * ((size_t)g.S0 + g.upper_stride);
    const int r = lane & 31, c2 = lane >> 5;
#pragma unroll 4
    for (int j = 2 * wave + c2; j < 128; j += 8)  // column x0 + j, 32 consecutive rows of it
        if (x0 + j < cols && y0 + r < rows) ref_store_stream(&out[(size_t)(x0 + j) * rows + y0 + r], tile[r][j]);
}
void launch_ref_dense_planes_keyframe(const Geom& g, Pyramid kf, const uint16_t* depth, Records rec, int n_pairs, hipStream_t s) {
    if (g.mode != VORS_CANDIDATES_DENSE || !rec.dense_t.recs) return;
    launch_zero_ints(g, rec.dense_t.n_valid, VORS_MAX_LEVELS, n_pairs, s);
    int first = 0;
    if (g.lv[0].cols % 4 == 0 && g.S0 % 4 == 0 && ((uintptr_t)kf.level0) % 4 == 0 && ((uintptr_t)depth) % 8 == 0 && !getenv("VORS_REF_RECORDS_GENERIC")) {
        hipLaunchKernelGGL(ref_dense_records_level0_kernel, dim3((g.lv[0].cols + 127) / 128, (g.lv[0].rows + 31) / 32, n_pairs), dim3(256), 0, s, g, kf.level0,
                           depth, rec);
        first = 1;
    }
    for (int l = first; l < g.L; ++l)
        hipLaunchKernelGGL(ref_dense_records_kernel, dim3((g.lv[l].cols + 63) / 64, (g.lv[l].rows + 31) / 32, n_pairs), dim3(256), 0, s, g, kf.level0,
                           kf.upper, depth, rec, l);
}
// u8 planes whose sides are multiples of 4 (level 0 of any usual image): 128 x 128 tiles, 128-byte runs and dword accesses on both sides.
__global__ __launch_bounds__(256) void ref_transpose_u8_wide_kernel(Geom g, const uint8_t* __restrict__ src, size_t src_stride, uint8_t* __restrict__ dst,
                                                                    size_t dst_stride, int rows, int cols) {
    __shared__ __attribute__((aligned(16))) uint8_t t[128][132];  // t[x][y]; pitch 132: rows stay dword-aligned
    const int pair = select_pair(g, blockIdx.z);
    if (pair < 0) return;
    const uint8_t* sp = src + (size_t)pair * src_stride;
    uint8_t* dp = dst + (size_t)pair * dst_stride;
    const int x0 = blockIdx.x * 128, y0 = blockIdx.y * 128;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int d = (int)threadIdx.x + 256 * k, row = d >> 5, xd = d & 31;
        const int x = x0 + 4 * xd, y = y0 + row;
        if (y < rows && x < cols) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(sp + (size_t)y * cols + x);
            t[4 * xd + 0][row] = (uint8_t)v;
            t[4 * xd + 1][row] = (uint8_t)(v >> 8);
            t[4 * xd + 2][row] = (uint8_t)(v >> 16);
            t[4 * xd + 3][row] = (uint8_t)(v >> 24);
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int d = (int)threadIdx.x + 256 * k, col = d >> 5, yd = d & 31;
        const int x = x0 + col, y = y0 + 4 * yd;
        if (x < cols && y < rows) *reinterpret_cast<uint32_t*>(dp + (size_t)x * rows + y) = *reinterpret_cast<const uint32_t*>(&t[col][4 * yd]);
    }
}
// The current frame's pyramid, column-major: level 0 is transposed, the levels above are the mean pyramid OF THE TRANSPOSED IMAGE — halving
// commutes with transposition (the 2x2 mean is symmetric, both sides halve with the same floor: multires.rs:21-31,67-88), so the pyramid
// kernel on the geometry with rows and columns exchanged writes exactly the transposed levels, in the same slots, at its streaming rate.
void launch_ref_dense_planes_current(const Geom& g, Pyramid cur, Records rec, int n_pairs, hipStream_t s) {
    const RefDensePlanes& t = rec.dense_t;
    if (g.mode != VORS_CANDIDATES_DENSE || !t.cur0) return;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    if (rows % 4 == 0 && cols % 4 == 0 && ((uintptr_t)cur.level0) % 4 == 0 && ((uintptr_t)t.cur0) % 4 == 0)
        hipLaunchKernelGGL(ref_transpose_u8_wide_kernel, dim3((cols + 127) / 128, (rows + 127) / 128, n_pairs), dim3(256), 0, s, g, cur.level0, (size_t)g.S0,
                           t.cur0, (size_t)g.S0, rows, cols);
    else
        ref_transpose<uint8_t>(g, cur.level0, (size_t)g.S0, t.cur0, (size_t)g.S0, rows, cols, n_pairs, s);
    Geom gt = g;
    gt.sel_list = nullptr;
    gt.sel_count = nullptr;
    for (int l = 0; l < g.L; ++l) std::swap(gt.lv[l].rows, gt.lv[l].cols);
    launch_pyramid(gt, Pyramid{t.cur0, t.curu}, n_pairs, s);
}

// ---- operator level on explicit observations, sums in the order of the observations (the reference's eval on that Obs) --------------
template <bool HUBER>
__global__ __launch_bounds__(64) void lm_ref_eval_obs_kernel(RefObsSrc src, int n, RefImg c, const float* __restrict__ model7,
                                                             float* __restrict__ out, float* __restrict__ residuals) {
    __shared__ __attribute__((aligned(16))) float lds[RW_WORDS];
    const Iso model = ref_iso_uniform(iso_load(model7));
    int cnt;
    const float acc = refw_eval<HUBER>(src, n, c, model, lds, &cnt, residuals);
    float a[29];
    a[0] = refw_lane(acc, 0);
    a[1] = (float)cnt;
#pragma unroll
    for (int k = 1; k < RW_NSUM; ++k) a[k + 1] = refw_lane(acc, k);
    if (threadIdx.x == 0) {
        out[0] = a[0] / a[1];
        out[1] = a[1];
        for (int q = 0; q < 6; ++q) out[2 + q] = a[2 + q];
        int k = 8;
        for (int q = 0; q < 6; ++q)
            for (int r = q; r < 6; ++r) {
                out[8 + q * 6 + r] = a[k];
                out[8 + r * 6 + q] = a[k];
                ++k;
            }
    }
}
template <bool HUBER>
__global__ __launch_bounds__(64) void lm_ref_solve_obs_kernel(RefObsSrc src, int n, RefImg c, const float* __restrict__ model7,
                                                              float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[RW_WORDS];
    Iso model = ref_iso_uniform(iso_load(model7));
    int nb_iter = 0, n_full = 0;
    float energy = 0.f, lm_coef = 0.f;
    const bool ok = refw_solve_level<HUBER, 0>(src, n, c, &model, &nb_iter, &energy, &lm_coef, &n_full, lds, RefHandoff{nullptr, nullptr, nullptr}, 0, 0, 0) == 1;
    if (threadIdx.x == 0) {
        iso_store(model, out);
        out[7] = (float)nb_iter;
        out[8] = energy;
        out[9] = lm_coef;
        out[10] = ok ? 0.f : 1.f;
    }
}
void launch_lm_eval_obs_reference(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta, const float* model7,
                                  float* out, float* residuals, hipStream_t s) {
    const RefObsSrc src{rec.A, rec.B, rec.C};
    const RefImg c{image, rows, cols, k, huber_delta};
    if (huber_delta > 0.f) hipLaunchKernelGGL(lm_ref_eval_obs_kernel<true>, dim3(1), dim3(64), 0, s, src, n, c, model7, out, residuals);
    else hipLaunchKernelGGL(lm_ref_eval_obs_kernel<false>, dim3(1), dim3(64), 0, s, src, n, c, model7, out, residuals);
}
void launch_lm_solve_obs_reference(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta, const float* model7,
                                   float* out, hipStream_t s) {
    const RefObsSrc src{rec.A, rec.B, rec.C};
    const RefImg c{image, rows, cols, k, huber_delta};
    if (huber_delta > 0.f) hipLaunchKernelGGL(lm_ref_solve_obs_kernel<true>, dim3(1), dim3(64), 0, s, src, n, c, model7, out);
    else hipLaunchKernelGGL(lm_ref_solve_obs_kernel<false>, dim3(1), dim3(64), 0, s, src, n, c, model7, out);
}

// ------------------------------------------------------------------------------------------------------------
// Candidate lists into extract_z's order (inverse_compositional.rs:260-279): column-major, key = x * rows + y. Keys are unique (one
// candidate per pixel), so the rank of a candidate is the number of set bits below its key in a bitmap of the level: segments of 2^18
// keys in LDS, per-thread word totals + a block scan. One workgroup per (level, pair); out of place into `tmp`, then copied back.
// ------------------------------------------------------------------------------------------------------------
#define SORT_WORDS_MAX 9600  // at most 307,200 keys per segment (bits + 16-bit word prefixes: 57.6 KB of LDS); a launch per level, sized for it
#define SORT_BLOCK 512
#define SORT_U 8          // independent loads in flight per thread (a pass is a chain of global round trips otherwise)
__global__ __launch_bounds__(SORT_BLOCK) __attribute__((amdgpu_waves_per_eu(6))) void sort_colmajor_kernel(Geom g, Records rec, int reg_cap, int l_first, int seg_words) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sort_lds[];
    uint32_t* bits = sort_lds;                                            // [seg_words]
    uint16_t* wpre = reinterpret_cast<uint16_t*>(sort_lds + seg_words);   // [seg_words] set bits in the words before this one (a list has at most 65536 records; a value that wraps belongs to a word with no record behind it)
    __shared__ int wsum[SORT_BLOCK / 64];
    __shared__ int s_base;
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int l = l_first + (int)blockIdx.x;
    const unsigned SORT_WORDS = (unsigned)seg_words;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols;
    const unsigned nkeys = (unsigned)rows * (unsigned)cols;
    const int n = min(rec.n_used[(size_t)pair * VORS_MAX_LEVELS + l], g.lv[l].n_slots);
    if (n <= 1) return;
    const size_t lvl0 = (size_t)pair * g.slots_total + g.lv[l].slot_off;
    SlimRec* S = rec.S + lvl0;
    SlimRec* T = rec.sort_tmp + lvl0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    if (n <= reg_cap) {  // (reg_cap <= SORT_BLOCK * SORT_U: launch_sort_colmajor)
        // Round 4: a list of at most 4096 records (every level of the reference's shapes) stays in REGISTERS — 8 records per thread — while the
        // segments of the key space go by; each record learns its rank in its segment, and the list is written back IN PLACE at the end. One read
        // and one write of the list instead of a read per segment and pass, a scratch copy and a copy back: 3.5 GB of traffic per 4096
        // coarse-to-fine pairs became 1 GB.
        SlimRec r[SORT_U];
        int rank[SORT_U];
#pragma unroll
        for (int u = 0; u < SORT_U; ++u) {
            const int i = (int)threadIdx.x + u * SORT_BLOCK;
            r[u] = i < n ? S[i] : SlimRec{0xffffffffu, 0.f, 0u};
            rank[u] = -1;
        }
        for (unsigned seg0 = 0; seg0 < nkeys; seg0 += SORT_WORDS * 32u) {
            const unsigned seg_keys = min(nkeys - seg0, SORT_WORDS * 32u);
            const int words = (int)((seg_keys + 31u) >> 5);
            const int wpt = (words + SORT_BLOCK - 1) / SORT_BLOCK;
            for (int w = threadIdx.x; w < words; w += SORT_BLOCK) bits[w] = 0u;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < SORT_U; ++u) {
                const unsigned rel = (r[u].xy & 0xffffu) * (unsigned)rows + (r[u].xy >> 16) - seg0;
                if (r[u].xy != 0xffffffffu && rel < seg_keys) atomicOr(&bits[rel >> 5], 1u << (rel & 31u));
            }
            __syncthreads();
            const int w0 = threadIdx.x * wpt, w1 = min(words, w0 + wpt);
            int tot = 0;
            for (int w = w0; w < w1; ++w) tot += __popc(bits[w]);
            int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int run = incl - tot;
            for (int w = 0; w < wave; ++w) run += wsum[w];
            const int base = s_base;
            for (int w = w0; w < w1; ++w) {
                wpre[w] = (uint16_t)run;
                run += __popc(bits[w]);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < SORT_U; ++u) {
                const unsigned rel = (r[u].xy & 0xffffu) * (unsigned)rows + (r[u].xy >> 16) - seg0;
                if (r[u].xy != 0xffffffffu && rel < seg_keys) {
                    const unsigned w = rel >> 5;
                    rank[u] = base + wpre[w] + __popc(bits[w] & ((1u << (rel & 31u)) - 1u));
                }
            }
            __syncthreads();
            if (threadIdx.x == SORT_BLOCK - 1) s_base = base + run;
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < SORT_U; ++u)
            if (rank[u] >= 0 && rank[u] < n) S[rank[u]] = r[u];  // (every record of the list is in some thread's registers: in place)
        return;
    }
    for (unsigned seg0 = 0; seg0 < nkeys; seg0 += SORT_WORDS * 32u) {
        const unsigned seg_keys = min(nkeys - seg0, SORT_WORDS * 32u);
        const int words = (int)((seg_keys + 31u) >> 5);
        const int wpt = (words + SORT_BLOCK - 1) / SORT_BLOCK;  // consecutive words per thread
        for (int w = threadIdx.x; w < words; w += SORT_BLOCK) bits[w] = 0u;
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < n; i0 += SORT_BLOCK * SORT_U) {
            uint32_t xy[SORT_U];
#pragma unroll
            for (int u = 0; u < SORT_U; ++u) {
                const int i = i0 + u * SORT_BLOCK;
                xy[u] = i < n ? S[i].xy : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < SORT_U; ++u) {
                const unsigned rel = (xy[u] & 0xffffu) * (unsigned)rows + (xy[u] >> 16) - seg0;
                if (xy[u] != 0xffffffffu && rel < seg_keys) atomicOr(&bits[rel >> 5], 1u << (rel & 31u));
            }
        }
        __syncthreads();
        const int w0 = threadIdx.x * wpt, w1 = min(words, w0 + wpt);
        int tot = 0;
        for (int w = w0; w < w1; ++w) tot += __popc(bits[w]);
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int run = incl - tot;
        for (int w = 0; w < wave; ++w) run += wsum[w];
        const int base = s_base;
        for (int w = w0; w < w1; ++w) {
            wpre[w] = run;
            run += __popc(bits[w]);
        }
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < n; i0 += SORT_BLOCK * (SORT_U / 2)) {
            SlimRec r[SORT_U / 2];
#pragma unroll
            for (int u = 0; u < SORT_U / 2; ++u) {
                const int i = i0 + u * SORT_BLOCK;
                r[u] = i < n ? S[i] : SlimRec{0xffffffffu, 0.f, 0u};
            }
#pragma unroll
            for (int u = 0; u < SORT_U / 2; ++u) {
                const unsigned rel = (r[u].xy & 0xffffu) * (unsigned)rows + (r[u].xy >> 16) - seg0;
                if (r[u].xy != 0xffffffffu && rel < seg_keys) {
                    const unsigned w = rel >> 5;
                    const int rank = base + wpre[w] + __popc(bits[w] & ((1u << (rel & 31u)) - 1u));
                    if (rank < n) T[rank] = r[u];
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == SORT_BLOCK - 1) s_base = base + run;  // (the last thread's running count = the segment's total)
        __syncthreads();
    }
    // (workgroup scope is all this needs — T was written by this workgroup and is read back by it, through the CU's own L1 / the XCD's
    // L2; an agent-scope __threadfence() here writes the whole L2 back once per workgroup: 6 ms per 4096 pairs)
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < n; i0 += SORT_BLOCK * (SORT_U / 2)) {
        SlimRec r[SORT_U / 2];
#pragma unroll
        for (int u = 0; u < SORT_U / 2; ++u) {
            const int i = i0 + u * SORT_BLOCK;
            if (i < n) r[u] = T[i];
        }
#pragma unroll
        for (int u = 0; u < SORT_U / 2; ++u) {
            const int i = i0 + u * SORT_BLOCK;
            if (i < n) S[i] = r[u];
        }
    }
}
// ------------------------------------------------------------------------------------------------------------
// Coarse-to-fine mode: the same ranking straight from the keyframe kernel's staged regions (kernels.hip keyframe_sparse_kernel leaves the
// points of each wavefront region at the front of the region's slots + a count per region) — compaction AND sort in one pass: every
// thread keeps the coordinates of RANK_U slots in registers (a slot holds a point iff its position in the region is below the region's
// count), the whole key space of the level is ONE LDS bitmap (640x480: 9600 words + 16-bit word prefixes, 57.6 KB; larger levels take
// several segments), a point's rank = set bits below its key, and the record goes from the staging grid to its final place in the list.
// Replaces compact_regions_kernel + sort_colmajor_kernel (0.23 + 0.40 ms per 4096 pairs) whenever every level fits RANK_U slots per thread.
// ------------------------------------------------------------------------------------------------------------
#define RANK_WORDS 9600  // at most 307,200 keys per segment (57.6 KB of LDS)
#define RANK_BLOCK 512
#define RANK_U_MAX 40    // slots per thread: levels of at most 20,480 slots (640x480 6 levels: 9,600; 1280x960 7 levels: 19,200)
#define RANK_MAX_REGIONS 1024
// One launch per level: RANK_U slots per thread and `seg_words` words of bitmap (dynamic LDS) sized for THAT level, so that the small levels
// run many workgroups per CU instead of inheriting level 0's 58 KB and 40 predicated loads per thread.
template <int RANK_U>
__global__ __launch_bounds__(RANK_BLOCK) void rank_regions_kernel(Geom g, Records rec, int l, int seg_words) {
    extern __shared__ __attribute__((aligned(16))) uint32_t rank_lds[];
    uint32_t* bits = rank_lds;                                             // [seg_words]
    uint16_t* wpre = reinterpret_cast<uint16_t*>(rank_lds + seg_words);    // [seg_words] set bits in the words before this one (a level has < 65536 points)
    __shared__ int s_cnt[RANK_MAX_REGIONS];
    __shared__ int wsum[RANK_BLOCK / 64];
    __shared__ int s_base;
    const int pair = select_pair(g, blockIdx.x);
    if (pair < 0) return;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols;
    const unsigned nkeys = (unsigned)rows * (unsigned)cols;
    const int n_slots = g.lv[l].n_slots;
    const int cap_sh = (__ffs(rec.kf_r) - 1) + (g.L - 1 - l);  // log2 of the slots per region at this level
    const size_t lvl0 = (size_t)pair * g.slots_total + g.lv[l].slot_off;
    const SlimRec* src = rec.stage + lvl0;
    SlimRec* dst = rec.S + lvl0;
    const int* cnt = rec.region_cnt + ((size_t)pair * VORS_MAX_LEVELS + l) * rec.n_regions;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = threadIdx.x; r < rec.n_regions; r += RANK_BLOCK) s_cnt[r] = cnt[r];
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    SlimRec r[RANK_U];  // (whole records: one read of the staging grid, no dependent second one once the rank is known)
#pragma unroll
    for (int u = 0; u < RANK_U; ++u) {
        const int j = (int)threadIdx.x + u * RANK_BLOCK;
        const bool valid = j < n_slots && (j & ((1 << cap_sh) - 1)) < s_cnt[j >> cap_sh];
        r[u] = valid ? src[j] : SlimRec{0xffffffffu, 0.f, 0u};
    }
    for (unsigned seg0 = 0; seg0 < nkeys; seg0 += (unsigned)seg_words * 32u) {
        const unsigned seg_keys = min(nkeys - seg0, (unsigned)seg_words * 32u);
        const int words = (int)((seg_keys + 31u) >> 5);
        const int wpt = (words + RANK_BLOCK - 1) / RANK_BLOCK;  // consecutive words per thread
        for (int w = threadIdx.x; w < words; w += RANK_BLOCK) bits[w] = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RANK_U; ++u) {
            const unsigned rel = (r[u].xy & 0xffffu) * (unsigned)rows + (r[u].xy >> 16) - seg0;
            if (r[u].xy != 0xffffffffu && rel < seg_keys) atomicOr(&bits[rel >> 5], 1u << (rel & 31u));
        }
        __syncthreads();
        const int w0 = threadIdx.x * wpt, w1 = min(words, w0 + wpt);
        int tot = 0;
        for (int w = w0; w < w1; ++w) tot += __popc(bits[w]);
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int run = incl - tot;
        for (int w = 0; w < wave; ++w) run += wsum[w];
        const int base = s_base;
        for (int w = w0; w < w1; ++w) {
            wpre[w] = (uint16_t)run;
            run += __popc(bits[w]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RANK_U; ++u) {
            const unsigned rel = (r[u].xy & 0xffffu) * (unsigned)rows + (r[u].xy >> 16) - seg0;
            if (r[u].xy != 0xffffffffu && rel < seg_keys) {
                const unsigned w = rel >> 5;
                dst[base + (int)wpre[w] + __popc(bits[w] & ((1u << (rel & 31u)) - 1u))] = r[u];
            }
        }
        __syncthreads();
        if (threadIdx.x == RANK_BLOCK - 1) s_base = base + run;  // (the last thread's running count = the segment's total)
        __syncthreads();
    }
    if (threadIdx.x == 0) rec.n_used[(size_t)pair * VORS_MAX_LEVELS + l] = s_base;
}
// Whether launch_sort_colmajor takes the lists straight from the staged regions (then launch_keyframe skips compact_regions_kernel).
// VORS_REF_RANK=0 keeps the two-kernel form (A/B, tests).
bool ref_rank_from_regions(const Geom& g, const Records& rec) {
    if (g.arith != VORS_ARITH_REFERENCE || g.mode != VORS_CANDIDATES_COARSE_TO_FINE || !rec.stage || rec.n_regions > RANK_MAX_REGIONS) return false;
    if (!g.ref_rank) return false;  // (VORS_REF_RANK=0, resolved when the handle was created: capi.cpp build_geom)
    for (int l = 0; l < g.L; ++l)
        if (g.lv[l].n_slots > RANK_U_MAX * RANK_BLOCK || g.lv[l].n_slots >= 65536) return false;
    return true;
}

void launch_sort_colmajor(const Geom& g, Records rec, int n_pairs, hipStream_t s) {
    if (g.mode == VORS_CANDIDATES_DENSE || !rec.sort_tmp) return;
    if (ref_rank_from_regions(g, rec)) {
        for (int l = 0; l < g.L; ++l) {
            const int u = (g.lv[l].n_slots + RANK_BLOCK - 1) / RANK_BLOCK;
            const int words = std::min(RANK_WORDS, (g.lv[l].rows * g.lv[l].cols + 31) / 32);
            const size_t lds = (size_t)words * 6;
#define VORS_RANK_LAUNCH(U) hipLaunchKernelGGL(rank_regions_kernel<U>, dim3(n_pairs), dim3(RANK_BLOCK), lds, s, g, rec, l, words)
            if (u <= 1) VORS_RANK_LAUNCH(1);
            else if (u <= 2) VORS_RANK_LAUNCH(2);
            else if (u <= 5) VORS_RANK_LAUNCH(5);
            else if (u <= 10) VORS_RANK_LAUNCH(10);
            else if (u <= 20) VORS_RANK_LAUNCH(20);
            else VORS_RANK_LAUNCH(RANK_U_MAX);
#undef VORS_RANK_LAUNCH
        }
        return;
    }
    // VORS_REF_SORT_REGCAP (read per launch; tests): lists longer than this take the multi-pass form through the scratch copy — 0 forces it
    // for every list; the default is what 8 records per thread hold
    const char* e = getenv("VORS_REF_SORT_REGCAP");
    int reg_cap = SORT_BLOCK * SORT_U;
    if (e) reg_cap = std::max(0, std::min(reg_cap, atoi(e)));
    for (int l = 0; l < g.L; ++l) {  // (a launch per level: the bitmap of the level in one segment where it fits, and no more LDS than that)
        const int words = std::min(SORT_WORDS_MAX, (g.lv[l].rows * g.lv[l].cols + 31) / 32);
        hipLaunchKernelGGL(sort_colmajor_kernel, dim3(1, n_pairs), dim3(SORT_BLOCK), (size_t)words * 6, s, g, rec, reg_cap, l, words);
    }
}

}  // namespace vors
