// The Levenberg-Marquardt kernels of the direct-alignment hot path (gfx950 / CDNA4, wave64).
//
//   lm_track_kernel<BLOCK, Src>   Tracker::track's coarse->fine loop (reference src/core/track/inverse_compositional.rs:177-224)
//                                 with the whole optimizer::State loop (src/math/optimizer.rs:57-70,
//                                 src/core/track/lm_optimizer.rs:68-193) on the device: ONE workgroup per frame pair for all
//                                 levels and iterations; optimizer state lives in LDS; no host round trip.
//   lm_eval_obs_kernel            eval_energy + compute_eval_data for explicit observations (operator level)
//   lm_solve_obs_kernel           iterative_solve for explicit observations (operator level)
//
// Evaluation = eval_energy + compute_eval_data fused in ONE pass (lm_optimizer.rs:68-107): per point rotate+translate,
// project, strict inside test, 4 u8 taps, bilinear, residual, then 29 partial sums (sum r^2, count, g = sum J r (6),
// H = sum J J^T (21 unique)). J J^T is recomputed from J instead of streaming the reference's precomputed 6x6.
// Reduction: per-lane strided partial sums -> DPP row/bank reductions inside the wavefront -> fixed-order sum over the
// wavefronts through LDS. Deterministic for a given BLOCK. No MFMA: this is a reduction, not a contraction.
//
// Compile with -ffp-contract=off: per-point arithmetic follows the reference's evaluation order exactly (per-point
// residuals are bit-identical to the oracle); explicit fmaf only where the summation order differs anyway.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "device_common.h"
#include "engine.h"

// This file is compiled twice (csrc/Makefile): VORS_FUSED=0 -> lm_kernels.o, the EXACT arithmetic (the reference's evaluation order per
// point, bit-identical residuals / Jacobians; parity anchor; also holds the operator-level kernels), and VORS_FUSED=1 ->
// lm_kernels_fused.o, the FUSED arithmetic (vors_config.arithmetic = VORS_ARITH_FUSED): the same quantities from algebraically
// equivalent, shorter expressions — see "fused arithmetic" below. Kernels carry the mode as a template argument, so both objects link
// into one library.
#ifndef VORS_FUSED
#define VORS_FUSED 0
#endif

namespace vors {

constexpr bool kFused = VORS_FUSED != 0;

// Ablation knobs of the FUSED arithmetic (development builds only: tools/build_variant.sh; all 0 in the product). Each one puts ONE
// ingredient of FUSED back to the reference's arithmetic so that tools/parity_ablate.py can attribute the out-of-tolerance tail.
#ifndef VORS_ABL_EXACT_STEP
#define VORS_ABL_EXACT_STEP 0     // step(): lm_step (IEEE sqrt / divisions) instead of lm_step_fast
#endif
#ifndef VORS_ABL_IEEE_IDEPTH
#define VORS_ABL_IEEE_IDEPTH 0    // dense level 0: scale / depth as an IEEE division instead of scale * rcp(depth)
#endif
#ifndef VORS_ABL_WIDE_IDENTITY
#define VORS_ABL_WIDE_IDENTITY 0  // near-identity window (EXACT fallback) 1e-4 / 1e-4 instead of 5e-6 / 1e-6
#endif
#ifndef VORS_ABL_IEEE_RCP
#define VORS_ABL_IEEE_RCP 0       // warp: IEEE 1 / z' instead of v_rcp_f32
#endif
#ifndef VORS_ABL_EXACT_WARP
#define VORS_ABL_EXACT_WARP 0     // warp: (u, v) from the reference's chain (back_project, Iso3 * p, project); the rest stays FUSED
#endif


#define NACC 29
#define LM_MAX_WAVES 16

struct LmShared {
    alignas(16) float part[LM_MAX_WAVES * 4 * 32];  // partial sums per 16-lane row of each wavefront (block_reduce: slot (v & 3) * 8 + (v >> 2) = sum v)
    float sums[2][32];              // ping-pong totals: [cur] = kept state's sums, [1-cur] = candidate's
    float cand[8];                  // candidate model (7) + step-ok flag, broadcast from the solving lane
    float misc[LM_MAX_WAVES * 2];
};

// d / 2 with truncation toward zero (Rust's `/` on i16, gradient.rs:28-29,79,92), branch-free.
__device__ __forceinline__ int half_trunc(int d) { return (d + (int)((unsigned)d >> 31)) >> 1; }

// A candidate point ready for warping: back-projected keyframe point + template grey level (< 0 = empty slot).
struct Pos {
    float X, Y, Z;  // camera.rs:135-140 applied to (x, y, 1/_z)
    float tmpl;
};

// Point sources hand out GROUPS of G independent points per thread and iteration (instruction-level parallelism):
//   fetch(cursor, n_units, raw)   issue every load of the group
//   positions(raw, pos[G])        back-projected points (cheap part, needed before the taps can be addressed)
//   jacobians(raw, J[G][6])       warp Jacobians (inverse_compositional.rs:313-341) — evaluated while the taps are in flight
//   slot(raw, g)                  record index of point g (only used when residuals are written, operator level)

// ---- point source: stored record planes (sparse mode, operator level). G = 2 slots (i, i + BLOCK).
struct RecSrc {
    static constexpr bool FUSED = false;
    static constexpr int G = 2;
    static constexpr bool PREFETCH = false;
    static constexpr bool SKIP_EMPTY = true;
    const float4* A;
    const float4* B;
    const float2* C;
    struct Raw {
        float4 a[2], b[2];
        float2 c[2];
        int i[2];
    };
    struct Cursor {
        int i;
    };
    template <int BLOCK>
    __device__ __forceinline__ Cursor begin(int first = 0) const {
        return Cursor{first + (int)threadIdx.x};
    }
    template <int BLOCK>
    __device__ __forceinline__ Cursor advance(const Cursor& c) const {
        return Cursor{c.i + 2 * BLOCK};
    }
    template <int BLOCK>
    __device__ __forceinline__ void fetch(const Cursor& cur, int n, Raw& r) const {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int i = cur.i + g * BLOCK;
            r.i[g] = i;
            r.a[g] = (i < n) ? A[(unsigned)i] : make_float4(0.f, 0.f, 0.f, -1.f);  // 32-bit offsets from the uniform plane bases
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const bool v = r.a[g].w >= 0.f;
            r.b[g] = v ? B[(unsigned)r.i[g]] : make_float4(0.f, 0.f, 0.f, 0.f);
            r.c[g] = v ? C[(unsigned)r.i[g]] : make_float2(0.f, 0.f);
        }
    }
    __device__ __forceinline__ void positions(const Raw& r, Pos p[2]) const {
#pragma unroll
        for (int g = 0; g < 2; ++g) p[g] = Pos{r.a[g].x, r.a[g].y, r.a[g].z, r.a[g].w};
    }
    __device__ __forceinline__ void jacobian(const Raw& r, int g, float J[6]) const {
        J[0] = r.b[g].x; J[1] = r.b[g].y; J[2] = r.b[g].z; J[3] = r.b[g].w;
        J[4] = r.c[g].x; J[5] = r.c[g].y;
    }
    __device__ __forceinline__ int slot(const Raw& r, int g, int n) const { return r.i[g] < n ? r.i[g] : -1; }
};

// ---- point source: compact 12-byte candidate lists (coarse-to-fine and generic-mask modes of the tracker). G = 2 points (i, i + BLOCK).
// The back-projected point (camera.rs:135-140 applied to (x, y, 1/_z)) and the warp Jacobian (inverse_compositional.rs:313-341) are
// recomputed per evaluation with exactly the arithmetic of the reference's precompute: bit-identical values for 12 B of traffic per
// point instead of 40.
struct SlimSrc {
    static constexpr bool FUSED = false;
    static constexpr int G = 2;
    static constexpr bool PREFETCH = false;
    static constexpr bool SKIP_EMPTY = false;
    const SlimRec* S;
    Intr k;
    FastDiv fu, fv;  // the focal lengths as verified fast divisors (lie.h div_uniform; `ok` = 0: IEEE division)
    struct Raw {
        SlimRec r[2];
        bool valid[2];
    };
    struct Cursor {
        int i;
    };
    template <int BLOCK>
    __device__ __forceinline__ Cursor begin(int first = 0) const {
        return Cursor{first + (int)threadIdx.x};
    }
    template <int BLOCK>
    __device__ __forceinline__ Cursor advance(const Cursor& c) const {
        return Cursor{c.i + 2 * BLOCK};
    }
    template <int BLOCK>
    __device__ __forceinline__ void fetch(const Cursor& cur, int n, Raw& r) const {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int i = cur.i + g * BLOCK;
            r.valid[g] = i < n;
            r.r[g] = S[(unsigned)(r.valid[g] ? i : 0)];  // (a lane past the end re-reads record 0 and masks it)
        }
    }
    __device__ __forceinline__ void positions(const Raw& r, Pos p[2]) const {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const V3 P = back_project_rt(IntrFast{k, fu, fv}, (float)(r.r[g].xy & 0xffffu), (float)(r.r[g].xy >> 16), 1.0f / r.r[g].iz);
            p[g] = Pos{P.x, P.y, P.z, r.valid[g] ? (float)(r.r[g].tg & 0xff) : -1.0f};
        }
    }
    __device__ __forceinline__ void jacobian(const Raw& r, int g, float J[6]) const {
        warp_jacobian_at_rt((float)slim_gx(r.r[g].tg), (float)slim_gy(r.r[g].tg), (float)(r.r[g].xy & 0xffffu), (float)(r.r[g].xy >> 16),
                            r.r[g].iz, IntrFast{k, fu, fv}, J);
    }
    __device__ __forceinline__ int slot(const Raw&, int, int) const { return -1; }
};

// ---- point sources for dense mode: NOTHING is stored per point at level 0. Each evaluation recomputes the point from the
// keyframe image (template + integer gradient, gradient.rs:15-33 / 74-93), the depth map (level 0: from_depth,
// inverse_depth.rs:24-29) or the fused inverse-depth plane (levels >= 1), with exactly the arithmetic of the keyframe
// precompute (back_project camera.rs:135-140, warp_jacobian_at inverse_compositional.rs:313-341): bit-identical points
// for ~4 B of HBM traffic per point instead of 40.
//
// DenseSrc: one pixel per unit, any image width (fallback, keyframe test, diagnostics).
template <bool LEVEL0>
struct DenseSrc {
    static constexpr bool FUSED = false;
    static constexpr int G = 1;
    static constexpr bool PREFETCH = false;
    static constexpr bool SKIP_EMPTY = false;
    const uint8_t* kimg;    // keyframe image of this level
    const uint8_t* kfine;   // next finer keyframe level (levels >= 1)
    const uint16_t* depth;  // level 0
    const float* iz;        // levels >= 1: fused inverse depth, NaN = Unknown
    int rows, cols, fcols;
    Intr k;
    float depth_scale;
    struct Cursor {
        int i, x, y;
        __device__ __forceinline__ int& xq() { return x; }
        __device__ __forceinline__ const int& xq() const { return x; }
    };
    struct Raw {
        int i, x, y;
        float izv;
        int gx, gy, tm;
        bool valid;
    };
    template <int BLOCK>
    __device__ __forceinline__ Cursor begin(int first = 0) const {
        const int t = first + (int)threadIdx.x;
        const int y = t / cols;
        return Cursor{t, t - y * cols, y};
    }
    template <int BLOCK>
    __device__ __forceinline__ Cursor advance(const Cursor& c) const {
        const int dy = BLOCK / cols, dx = BLOCK - dy * cols;  // workgroup-uniform
        int x = c.x + dx, y = c.y + dy;
        if (x >= cols) {
            x -= cols;
            y += 1;
        }
        return Cursor{c.i + BLOCK, x, y};
    }
    template <int BLOCK>
    __device__ __forceinline__ void fetch(const Cursor& c, int n, Raw& r) const {
        r.i = c.i;
        r.x = c.x;
        r.y = c.y;
        r.tm = kimg[c.i];
        if (LEVEL0) {
            const bool interior = c.x > 0 && c.y > 0 && c.x < cols - 1 && c.y < rows - 1;
            const uint8_t* p = kimg + c.i;
            const int l = p[interior ? -1 : 0], rr = p[interior ? 1 : 0], u = p[interior ? -cols : 0], d = p[interior ? cols : 0];
            r.gx = (rr - l) / 2;  // borders: the taps alias the centre pixel -> 0, like gradient.rs:15-33
            r.gy = (d - u) / 2;
            const int dz = depth[c.i];
            r.valid = dz != 0;
            r.izv = depth_scale / (float)dz;
        } else {
            const uint8_t* p = kfine + (size_t)(2 * c.y) * fcols + 2 * c.x;
            const int a = p[0], cc = p[1], b = p[fcols], d = p[fcols + 1];
            r.gx = (cc + d - a - b) / 2;
            r.gy = (b - a + d - cc) / 2;
            r.izv = iz[c.i];
            r.valid = !(r.izv != r.izv);
        }
    }
    __device__ __forceinline__ void positions(const Raw& r, Pos p[1]) const {
        const V3 P = back_project(k, (float)r.x, (float)r.y, 1.0f / r.izv);
        p[0] = Pos{P.x, P.y, P.z, r.valid ? (float)r.tm : -1.0f};
    }
    __device__ __forceinline__ void jacobian(const Raw& r, int, float J[6]) const {
        warp_jacobian_at((float)r.gx, (float)r.gy, (float)r.x, (float)r.y, r.izv, k, J);
    }
    __device__ __forceinline__ int slot(const Raw& r, int g, int n) const { return r.i; }
};

// DenseQuadSrc: FOUR horizontally adjacent pixels per unit (cols % 4 == 0). One dword / dwordx2 / dwordx4 load per image
// row and plane instead of ~8 byte loads per pixel: coalesced 256 B - 1 KiB per wavefront instruction, 4-way ILP per lane.
template <bool LEVEL0, bool FAST>
struct DenseQuadSrc {
    static constexpr bool FUSED = false;
    static constexpr int G = 4;
    static constexpr bool PREFETCH = false;
    static constexpr bool SKIP_EMPTY = false;
    const uint8_t* kimg;
    const uint8_t* kfine;
    const uint16_t* depth;
    const float* iz;
    int rows, cols, fcols, qcols;  // qcols = cols / 4
    IntrFast kf;
    const float2* lut;  // level 0: depth -> (inverse depth, 1 / inverse depth), exact table
    struct Cursor {
        int i, qx, y;  // i = quad index
        __device__ __forceinline__ int& xq() { return qx; }
        __device__ __forceinline__ const int& xq() const { return qx; }
    };
    struct Loaded {  // raw words of one quad, straight from memory (kept in flight one iteration ahead)
        uint32_t cw, w1, w2, w3, w4;  // level 0: centre/up/down rows + left/right bytes; levels >= 1: fine rows (2 x uint2) + unused
        uint32_t d0, d1, d2, d3;      // level 0: depth (uint2) ; levels >= 1: inverse depths (float4 bits)
        int x0, y;
    };
    struct Raw {
        int x0, y;
        float izv[4], zv[4];
        int gx[4], gy[4], tm[4];
        bool valid[4];
    };
    template <int BLOCK>
    __device__ __forceinline__ Cursor begin(int first = 0) const {
        const int t = first + (int)threadIdx.x;
        const int y = t / qcols;
        return Cursor{t, t - y * qcols, y};
    }
    template <int BLOCK>
    __device__ __forceinline__ Cursor advance(const Cursor& c) const {
        const int dy = BLOCK / qcols, dx = BLOCK - dy * qcols;  // workgroup-uniform
        int qx = c.qx + dx, y = c.y + dy;
        if (qx >= qcols) {
            qx -= qcols;
            y += 1;
        }
        return Cursor{c.i + BLOCK, qx, y};
    }
    __device__ __forceinline__ void load(const Cursor& c, Loaded& r) const {
        const int x0 = 4 * c.qx, y = c.y;
        r.x0 = x0;
        r.y = y;
        // 32-bit unsigned offsets from workgroup-uniform bases (scalar base + vector offset addressing, no 64-bit VALU arithmetic)
        const unsigned ucols = (unsigned)cols, o = (unsigned)y * ucols + (unsigned)x0;
        r.cw = *reinterpret_cast<const uint32_t*>(kimg + o);
        if (LEVEL0) {
            const bool yin = y > 0 && y < rows - 1;
            r.w1 = *reinterpret_cast<const uint32_t*>(kimg + (o - (yin ? ucols : 0u)));
            r.w2 = *reinterpret_cast<const uint32_t*>(kimg + (o + (yin ? ucols : 0u)));
            r.w3 = kimg[o - (x0 > 0 ? 1u : 0u)];
            r.w4 = kimg[o + (x0 + 4 < cols ? 4u : 3u)];
            const uint2 dzw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(depth) + (o << 1));
            r.d0 = dzw.x;
            r.d1 = dzw.y;
            r.d2 = r.d3 = 0;
        } else {
            const unsigned ufc = (unsigned)fcols, fo = (unsigned)(2 * y) * ufc + (unsigned)(2 * x0);
            const uint2 f0 = *reinterpret_cast<const uint2*>(kfine + fo);
            const uint2 f1 = *reinterpret_cast<const uint2*>(kfine + (fo + ufc));
            r.w1 = f0.x; r.w2 = f0.y; r.w3 = f1.x; r.w4 = f1.y;
            const uint4 z4 = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(iz) + (o << 2));
            r.d0 = z4.x; r.d1 = z4.y; r.d2 = z4.z; r.d3 = z4.w;
        }
    }
    __device__ __forceinline__ void decode(const Loaded& l, Raw& r) const {
        const int x0 = l.x0, y = l.y;
        r.x0 = x0;
        r.y = y;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.tm[j] = (l.cw >> (8 * j)) & 0xff;
        if (LEVEL0) {
            // centred differences, truncating /2, zero on the 1-px border (gradient.rs:15-33)
            const int yin = (y > 0 && y < rows - 1) ? -1 : 0;
            const int b[6] = {(int)l.w3, r.tm[0], r.tm[1], r.tm[2], r.tm[3], (int)l.w4};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x0 + j;
                const int in = yin & ((x > 0 && x < cols - 1) ? -1 : 0);  // all-ones inside, 0 on the 1-px border
                const int up = (l.w1 >> (8 * j)) & 0xff, dn = (l.w2 >> (8 * j)) & 0xff;
                r.gx[j] = half_trunc(b[j + 2] - b[j]) & in;  // masks, not branches: keeps the loop one basic block
                r.gy[j] = half_trunc(dn - up) & in;
                const int dz = (j < 2 ? (l.d0 >> (16 * j)) : (l.d1 >> (16 * (j - 2)))) & 0xffff;
                r.valid[j] = dz != 0;
                // (scale / dz, 1 / (scale / dz)): inverse_depth.rs:24-29, lm_optimizer.rs:215
                const float2 zl = *reinterpret_cast<const float2*>(reinterpret_cast<const uint8_t*>(lut) + ((unsigned)dz << 3));
                r.izv[j] = zl.x;
                r.zv[j] = zl.y;
            }
        } else {
            // 2x2 block gradients of the next finer level (gradient.rs:74-93)
            const uint32_t zz[4] = {l.d0, l.d1, l.d2, l.d3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w0 = j < 2 ? l.w1 : l.w2, w1 = j < 2 ? l.w3 : l.w4;
                const int sh = (j & 1) * 16;
                const int a = (w0 >> sh) & 0xff, cc = (w0 >> (sh + 8)) & 0xff;
                const int bb = (w1 >> sh) & 0xff, d = (w1 >> (sh + 8)) & 0xff;
                r.gx[j] = half_trunc(cc + d - a - bb);
                r.gy[j] = half_trunc(bb - a + d - cc);
                r.izv[j] = __int_as_float((int)zz[j]);
                r.zv[j] = 1.0f / r.izv[j];
                r.valid[j] = !(r.izv[j] != r.izv[j]);
            }
        }
    }
    template <int BLOCK>
    __device__ __forceinline__ void fetch(const Cursor& c, int n, Raw& r) const {
        Loaded l;
        load(c, l);
        decode(l, r);
    }
    __device__ __forceinline__ void positions(const Raw& r, Pos p[4]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const V3 P = back_project_fast<FAST>(kf, (float)(r.x0 + j), (float)r.y, r.zv[j]);
            p[j] = Pos{P.x, P.y, P.z, r.valid[j] ? (float)r.tm[j] : -1.0f};
        }
    }
    __device__ __forceinline__ void jacobian(const Raw& r, int j, float J[6]) const {
        warp_jacobian_at_fast<FAST>((float)r.gx[j], (float)r.gy[j], (float)(r.x0 + j), (float)r.y, r.izv[j], kf, J);
    }
    __device__ __forceinline__ int slot(const Raw& r, int g, int n) const { return r.y * cols + r.x0 + g; }
};

struct ImgCtx {
    const uint8_t* img;  // current image of this level, row-major
    int rows, cols;
    Intr k;
    float huber;
    double inv_fu_d, inv_fv_d;    // level constants of the fused arithmetic (engine.h LevelGeom), 0 = form them here
    float inv_fu, inv_fv, s_fuv;
    bool force_exact = false;     // FUSED kernels: evaluate this level in the EXACT arithmetic (workgroup-uniform)
    bool exact_warp = false;      // FUSED kernels, candidate lists: (u, v) of this level from the reference's warp chain, the rest fused
    bool exact_step = false;      // FUSED kernels: step() with lm_step instead of lm_step_fast at this level
};

// warp (lm_optimizer.rs:213-219) + interpolate's inside test (lm_optimizer.rs:227-231): tap address or "outside".
struct Warped {
    float u, v, uf, vf;
    int off;      // offset of tap (v0, u0); 0 when outside (a safe address)
    bool inside;  // valid && inside
};
__device__ __forceinline__ Warped warp_point(const ImgCtx& c, const Iso& model, const Pos& p) {
    Warped w;
    const V3 p2 = iso_transform_point(model, V3{p.X, p.Y, p.Z});
    project_uv(c.k, p2, &w.u, &w.v);
    w.uf = floorf(w.u);
    w.vf = floorf(w.v);
    w.inside = (p.tmpl >= 0.f) && (w.uf >= 0.f) && (w.uf < (float)(c.cols - 2)) && (w.vf >= 0.f) && (w.vf < (float)(c.rows - 2));
    // masked, not branched (an outside point reads the safe address 0 and is selected away later)
    w.off = (__float2int_rz(w.vf) * c.cols + __float2int_rz(w.uf)) & (w.inside ? -1 : 0);
    return w;
}
struct Taps {
    uint32_t top, bot;  // (t00 | t01 << 8), (t10 | t11 << 8)
};
__device__ __forceinline__ Taps load_taps(const ImgCtx& c, const Warped& w) {
    const unsigned o = (unsigned)w.off;  // >= 0 by construction; 32-bit offset from the uniform image base
    uint16_t a, b;
    __builtin_memcpy(&a, c.img + o, 2);  // two adjacent bytes per row: one (possibly unaligned) 16-bit load each
    __builtin_memcpy(&b, c.img + (o + (unsigned)c.cols), 2);
    return Taps{a, b};
}
// bilinear (lm_optimizer.rs:236-247, term order as written) + residual + the 29 sums. Returns the residual (NaN if outside).
// ENERGY_ONLY: eval_energy alone (lm_optimizer.rs:68-87) — the sums 0 and 1; the Jacobian is not touched.
template <bool HUBER, bool ENERGY_ONLY = false>
__device__ __forceinline__ float accumulate_point(const ImgCtx& c, float tmpl, const float J[6], const Warped& w, const Taps& t,
                                                  float acc[NACC]) {
    const float vu_00 = (float)(t.top & 0xff), vu_01 = (float)(t.top >> 8), vu_10 = (float)(t.bot & 0xff), vu_11 = (float)(t.bot >> 8);
    const float fa = w.u - w.uf, fb = w.v - w.vf;
    const float im = (1.0f - fb) * (1.0f - fa) * vu_00 + fb * (1.0f - fa) * vu_10 + (1.0f - fb) * fa * vu_01 + fb * fa * vu_11;
    const float r_true = im - tmpl;
    // Outside / empty points contribute exactly nothing: selected to zero (never multiplied: 0 * inf would poison the sums).
    const float r = w.inside ? r_true : 0.f;
    if (ENERGY_ONLY) {
        if (HUBER) {
            const float ar = fabsf(r);
            acc[0] += ar > c.huber ? c.huber * (2.0f * ar - c.huber) : r * r;
        } else {
            acc[0] = fmaf(r, r, acc[0]);
        }
        acc[1] += w.inside ? 1.0f : 0.f;
        return w.inside ? r_true : __builtin_nanf("");
    }
    float Jm[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) Jm[q] = w.inside ? J[q] : 0.f;
    float wgt = 1.0f, wr = r;
    if (HUBER) {  // extension (not in the reference)
        const float ar = fabsf(r);
        const bool lin = ar > c.huber;
        acc[0] += lin ? c.huber * (2.0f * ar - c.huber) : r * r;
        wgt = lin ? c.huber / ar : 1.0f;
        wr = wgt * r;
    } else {
        acc[0] = fmaf(r, r, acc[0]);
    }
    acc[1] += w.inside ? 1.0f : 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[2 + q] = fmaf(Jm[q], wr, acc[2 + q]);
    int h = 8;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float jq = HUBER ? wgt * Jm[q] : Jm[q];
#pragma unroll
        for (int s = q; s < 6; ++s) {
            acc[h] = fmaf(jq, Jm[s], acc[h]);
            ++h;
        }
    }
    return w.inside ? r_true : __builtin_nanf("");
}

// Values read from LDS are uniform across the workgroup but land in vector registers; readfirstlane moves them to SGPRs.
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ Iso iso_uniform(const Iso& m) {
    return Iso{V3{uniform_f(m.t.x), uniform_f(m.t.y), uniform_f(m.t.z)}, Quat{uniform_f(m.q.i), uniform_f(m.q.j), uniform_f(m.q.k), uniform_f(m.q.w)}};
}

// ------------------------------------------------------------------------------------------------------------
// FUSED ARITHMETIC (vors_config.arithmetic = VORS_ARITH_FUSED).  Same quantities, shorter expressions, f32 throughout.
//
//  warp (lm_optimizer.rs:213-219 = back_project, Iso3 * point, project, / z): with P = z K^-1 (x, y, 1)^T,
//      (pu, pv, pz) / z = K R K^-1 (x, y, 1)^T + _z K t = H (x, y, 1)^T + _z m,      u = pu / pz, v = pv / pz
//    H (3x3) and m (3) are uniform per evaluation (formed once, in f64, from the model); H (x, y, 1)^T is affine in the pixel, so a
//    point costs 3 additions + 3 FMAs + one reciprocal + 2 multiplications instead of a quaternion rotation, two projections and
//    four divisions. The reciprocal is v_rcp_f32 (1 ulp): u, v carry ~1.5 ulp instead of the ~3-4 ulp the reference's own chain
//    accumulates.
//  interpolate (lm_optimizer.rs:236-247): two horizontal lerps + one vertical lerp (6 FMA-class ops instead of 4 triple products).
//  warp_jacobian_at (inverse_compositional.rs:313-341): the six entries are linear in (gu, gv) with coefficients that depend on the
//    pixel only through a = x - cu, b = y - cv and c' = c / (fu fv); the per-row parts are shared by the pixels of a quad.
//  level-0 inverse depth: scale * rcp(depth) instead of a division (or a table).
//  The inside test, the integer gradients / masks, the order of the 29 sums and the whole LM control flow are unchanged.
//  Gate: tests/test_gpu_parity.py runs every configuration in both modes: FUSED vs oracle <= 1e-4 (the north-star tolerance) and
//  FUSED vs EXACT reported.
// ------------------------------------------------------------------------------------------------------------
struct FusedCtx {  // uniform per evaluation, kept in scalar registers
    float h00, h01, h02, h10, h11, h12, h20, h21, h22;  // H = K R K^-1
    float m0, m1, m2;                                   // m = K t
    float h00_2, h00_3, h10_2, h10_3, h20_2, h20_3;     // 2x / 3x the first column: pixels x0+2, x0+3 of a quad
};
__device__ __forceinline__ FusedCtx make_fused_ctx(const ImgCtx& c, const Iso& m) {
    // H = K R K^-1 formed as I + K (R - I) K^-1, all in f32: D = R - I = 2 w [q]x + 2 [q]x^2 (the matrix of nalgebra's UnitQuaternion *
    // Vector3, lie.h quat_rotate, no unit-norm assumption) has entries of the size of the rotation angle and is formed without
    // cancellation, so M = K D K^-1 is good to ~1e-6 px in every entry — below the rounding of H itself as 9 floats (h00 x alone carries
    // 640 * 6e-8 = 4e-5 px). (The earlier f64 formulation of the same thing cost ~250 f64-heavy instructions per thread and evaluation:
    // a fifth of the per-pair kernel at the coarse levels / in the sparse modes; this is ~70 f32 ones.)
    const Intr& k = c.k;
    const float qi = m.q.i, qj = m.q.j, qk = m.q.k, qw = m.q.w;
    const float d00 = -2.0f * fmaf(qj, qj, qk * qk), d01 = 2.0f * fmaf(qi, qj, -(qk * qw)), d02 = 2.0f * fmaf(qi, qk, qj * qw);
    const float d10 = 2.0f * fmaf(qi, qj, qk * qw), d11 = -2.0f * fmaf(qi, qi, qk * qk), d12 = 2.0f * fmaf(qj, qk, -(qi * qw));
    const float d20 = 2.0f * fmaf(qi, qk, -(qj * qw)), d21 = 2.0f * fmaf(qj, qk, qi * qw), d22 = -2.0f * fmaf(qi, qi, qj * qj);
    const float fu = k.fu, fv = k.fv, sk = k.skew, cu = k.cu, cv = k.cv;
    // K D (camera.rs:126-132)
    const float a00 = fmaf(fu, d00, fmaf(sk, d10, cu * d20)), a01 = fmaf(fu, d01, fmaf(sk, d11, cu * d21)), a02 = fmaf(fu, d02, fmaf(sk, d12, cu * d22));
    const float a10 = fmaf(fv, d10, cv * d20), a11 = fmaf(fv, d11, cv * d21), a12 = fmaf(fv, d12, cv * d22);
    // ... K^-1 (camera.rs:135-140): col0 = A0 / fu, col1 = (A1 - s col0) / fv, col2 = A2 - cu col0 - cv col1
    const float ifu = c.inv_fu_d != 0.0 ? c.inv_fu : 1.0f / fu, ifv = c.inv_fu_d != 0.0 ? c.inv_fv : 1.0f / fv;  // (uniform branch)
    const float m00 = a00 * ifu, m10 = a10 * ifu, m20 = d20 * ifu;
    const float m01 = fmaf(-sk, m00, a01) * ifv, m11 = fmaf(-sk, m10, a11) * ifv, m21 = fmaf(-sk, m20, d21) * ifv;
    const float m02 = fmaf(-cv, m01, fmaf(-cu, m00, a02)), m12 = fmaf(-cv, m11, fmaf(-cu, m10, a12)), m22 = fmaf(-cv, m21, fmaf(-cu, m20, d22));
    const float tx = m.t.x, ty = m.t.y, tz = m.t.z;
    FusedCtx f;
    f.h00 = uniform_f(1.0f + m00); f.h01 = uniform_f(m01); f.h02 = uniform_f(m02);
    f.h10 = uniform_f(m10); f.h11 = uniform_f(1.0f + m11); f.h12 = uniform_f(m12);
    f.h20 = uniform_f(m20); f.h21 = uniform_f(m21); f.h22 = uniform_f(1.0f + m22);
    f.m0 = uniform_f(fmaf(fu, tx, fmaf(sk, ty, cu * tz)));
    f.m1 = uniform_f(fmaf(fv, ty, cv * tz));
    f.m2 = uniform_f(tz);
    f.h00_2 = uniform_f(2.0f * f.h00); f.h00_3 = uniform_f(3.0f * f.h00);
    f.h10_2 = uniform_f(2.0f * f.h10); f.h10_3 = uniform_f(3.0f * f.h10);
    f.h20_2 = uniform_f(2.0f * f.h20); f.h20_3 = uniform_f(3.0f * f.h20);
    return f;
}
// Workgroup-uniform: would this model move some pixel by less than ~1e-2 px?  (|rotation| <= 1e-5 rad, |t| <= 1e-6 m.)
__device__ __forceinline__ bool model_near_identity(const Iso& m) {
    const float r = fmaxf(fmaxf(fabsf(m.q.i), fabsf(m.q.j)), fabsf(m.q.k)), t = fmaxf(fmaxf(fabsf(m.t.x), fabsf(m.t.y)), fabsf(m.t.z));
    return __builtin_amdgcn_readfirstlane((r <= (VORS_ABL_WIDE_IDENTITY ? 1e-4f : 5e-6f) && t <= (VORS_ABL_WIDE_IDENTITY ? 1e-4f : 1e-6f)) ? 1 : 0) != 0;
}
// The context of (level intrinsics, model) written to / read from the split state (LmSplitState::fctx): formed by ONE thread per
// pair and round; the evaluation workgroups read it with scalar loads.
__device__ __forceinline__ void store_fused_ctx(const Geom& g, int lvl, const Iso& model, LmSplitState* st) {
    ImgCtx c;
    c.img = nullptr;
    c.rows = g.lv[lvl].rows;
    c.cols = g.lv[lvl].cols;
    c.k = g.lv[lvl].k;
    c.huber = g.huber_delta;
    c.inv_fu_d = g.lv[lvl].inv_fu_d; c.inv_fv_d = g.lv[lvl].inv_fv_d;
    c.inv_fu = g.lv[lvl].inv_fu; c.inv_fv = g.lv[lvl].inv_fv; c.s_fuv = g.lv[lvl].s_fuv;
    const FusedCtx f = make_fused_ctx(c, model);
    const float v[21] = {f.h00, f.h01, f.h02, f.h10, f.h11, f.h12, f.h20, f.h21, f.h22, f.m0, f.m1, f.m2,
                         f.h00_2, f.h00_3, f.h10_2, f.h10_3, f.h20_2, f.h20_3, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 18; ++q) st->fctx[q] = v[q];
    st->fctx_exact = model_near_identity(model) ? 1 : 0;
}
__device__ __forceinline__ FusedCtx load_fused_ctx(const LmSplitState* st) {
    FusedCtx f;
    float v[18];
    // written by an earlier launch, never by this one: read through the constant address space so that the (uniform) loads are scalar
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* cp = (cfloat*)(st->fctx);
#pragma unroll
    for (int q = 0; q < 18; ++q) v[q] = cp[q];
    f.h00 = v[0]; f.h01 = v[1]; f.h02 = v[2]; f.h10 = v[3]; f.h11 = v[4]; f.h12 = v[5]; f.h20 = v[6]; f.h21 = v[7]; f.h22 = v[8];
    f.m0 = v[9]; f.m1 = v[10]; f.m2 = v[11];
    f.h00_2 = v[12]; f.h00_3 = v[13]; f.h10_2 = v[14]; f.h10_3 = v[15]; f.h20_2 = v[16]; f.h20_3 = v[17];
    return f;
}
__device__ __forceinline__ bool fused_ctx_exact(const LmSplitState* st) {
    typedef const int __attribute__((address_space(4))) cint;
    return *(cint*)(&st->fctx_exact) != 0;
}
struct JacK {  // level constants of the Jacobian, uniform
    float fu, fv, s, cu, cv, inv_fu, inv_fv, s_fuv;
};
__device__ __forceinline__ JacK make_jack(const ImgCtx& c) {
    const Intr& k = c.k;
    JacK j;
    j.fu = k.fu; j.fv = k.fv; j.s = k.skew; j.cu = k.cu; j.cv = k.cv;
    if (c.inv_fu_d != 0.0) {  // (uniform) formed on the host for the tracker's levels
        j.inv_fu = c.inv_fu; j.inv_fv = c.inv_fv; j.s_fuv = c.s_fuv;
    } else {
        j.inv_fu = uniform_f((float)(1.0 / (double)k.fu));
        j.inv_fv = uniform_f((float)(1.0 / (double)k.fv));
        j.s_fuv = uniform_f((float)((double)k.skew / ((double)k.fu * (double)k.fv)));
    }
    return j;
}
__device__ __forceinline__ Taps load_taps_at(const ImgCtx& c, int off) {
    const unsigned o = (unsigned)off;  // >= 0 by construction; 32-bit offset from the uniform image base
    uint16_t a, b;
    __builtin_memcpy(&a, c.img + o, 2);  // two adjacent bytes per row: one (possibly unaligned) 16-bit load each
    __builtin_memcpy(&b, c.img + (o + (unsigned)c.cols), 2);
    return Taps{a, b};
}
// A unit (G points of one lane: 4 horizontally adjacent pixels, or one) decoded for the fused evaluation. Everything below is
// written G-WIDE — each step for all G points before the next step — because on gfx950 a wavefront's dependent VALU instructions
// issue every ~4.6 cycles however many wavefronts are resident (tools/ubench/valu_occ: ILP 1 -> 4.6, ILP 2 -> 2.7, ILP 4 -> 2.5
// cycles per instruction): the independent points of a unit are what fills the pipe.
template <int G>
struct FUnit {
    float bu[G], bv[G], bz[G];  // H (x, y, 1)^T
    float a[G], b[G];           // x - cu, y - cv
    float iz[G];                // inverse depth; finite for a pixel that is not a candidate (keeps every product finite)
    uint32_t tmw;               // template grey levels, one byte per point
    float gu[G], gv[G];         // integer gradients as floats (zero on the level-0 border); not set for energy-only evaluations
    bool valid[G];
    float px[G], py[G];         // pixel coordinates (only set / read where the reference's warp chain is evaluated: XW instantiations)
};
// What stage C (bilinear, residual, Jacobian, sums) needs of a unit whose taps are in flight.
// What bounds this loop (round 2, per-kernel probes with tools/lm_variants.sh at 4096 pairs, level-0 rounds):
//  * not latency: a software-pipelined loop (stage B of unit k+1 and the raw words of unit k+2 requested before stage C of unit k waits
//    for its taps, two units per basic block) left the energy-only round at 1760-1800 us (1759 as is) and the full round at 3303 (3339);
//  * the tap gathers, in the energy-only round: every tap load issued twice -> +66 % (full round +24 %), 16 extra FMAs per point -> +1 %
//    (full +5 %). A 16-bit (or 8-bit, or unaligned 32-bit) gather costs the CU ~19 cycles per wavefront instruction however well the
//    lanes' addresses line up, an aligned dword 12.7, 8 unaligned bytes 33;
//  * and VALU issue right behind: the current-image band of a workgroup staged in LDS (sampled bounding rows, 16-byte copies, taps as
//    ds_read2_b32 + v_alignbyte; every tap served from LDS in the bench) was SLOWER — 1927 / 3359 us: +60 VALU instructions per quad for
//    the LDS addressing, 4 instead of 8 workgroups per CU, and two barriers plus two memory latencies before a 19-iteration sweep.
// So the loop stays as simple as it is.
template <int G>
struct FusedStage {
    float fa[G], fb[G];  // fractional parts of (u, v)
    bool inside[G];      // candidate && inside the strict window of lm_optimizer.rs:227-231
    uint16_t top[G], bot[G];  // tap words (t00 | t01 << 8), (t10 | t11 << 8), as loaded: 16 bits (widened where they are consumed)
    uint32_t tmw;
    float a[G], b[G], iz[G], gu[G], gv[G];
};
template <int BYTE, class W>
__device__ __forceinline__ float cvt_ubyte(W w) {  // float(byte BYTE of w): one v_cvt_f32_ubyteN, whatever the other bytes hold
    float f;
    if (BYTE == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(w));
    if (BYTE == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(w));
    if (BYTE == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w));
    if (BYTE == 3) asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(w));
    return f;
}
// Stage B: warp (see the header of this section) + inside test + tap requests.
// XW: (u, v) from the reference's own chain — back_project (with the verified fast divisions by the focal lengths), Iso3 * point, project,
// two IEEE divisions by z' (lm_optimizer.rs:213-219) — instead of the homography form; everything after (u, v) stays fused. This is what
// the levels of FEW points of the candidate-list modes run (ImgCtx::exact_warp): ~150 instead of ~226 instructions per point
// of the full EXACT evaluation, with the same parity statistic (profiles/r03_parity_ablation.md, build `warp`; re-measured in round 4).
template <bool ENERGY_ONLY, bool XW = false, int G>
__device__ __forceinline__ void fused_stage_b(const FUnit<G>& p, const ImgCtx& c, const FusedCtx& f, FusedStage<G>& st, const Iso& model,
                                              const FastDiv& dfu = FastDiv{1.f, 1.f, 0}, const FastDiv& dfv = FastDiv{1.f, 1.f, 0}) {
    // gfx950 issues v_fma / v_mul / v_add / shifts every ~2.6 cycles but conversions, compares, selects, v_floor and v_fract every ~4.4
    // and v_rcp every ~8.9 (tools/ubench/valu_ops): floor-to-integer in ONE conversion (v_cvt_flr_i32_f32), the fractional part in one
    // v_fract, and the window test as two unsigned integer compares instead of four float ones.
    float hz[G], hu[G], hv[G], rz[G], u[G], v[G];
    int iu[G], iv[G], off[G];
#pragma unroll
    for (int g = 0; g < G; ++g) hz[g] = fmaf(f.m2, p.iz[g], p.bz[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) hu[g] = fmaf(f.m0, p.iz[g], p.bu[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) hv[g] = fmaf(f.m1, p.iz[g], p.bv[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) rz[g] = VORS_ABL_IEEE_RCP ? 1.0f / hz[g] : __builtin_amdgcn_rcpf(hz[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) u[g] = hu[g] * rz[g];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = hv[g] * rz[g];
    if (XW || VORS_ABL_EXACT_WARP) {
#ifndef VORS_XW_PLAIN_DIV
#define VORS_XW_PLAIN_DIV 1
#endif
        if constexpr (XW && VORS_XW_PLAIN_DIV) {
            // The three per-point divisions of the reference's chain (1 / _z, pu / z', pv / z') as the IEEE sequence WITHOUT its range scaling
            // (v_div_scale x 2, v_div_fmas): reciprocal, one Newton step, product, two fused corrections, v_div_fixup — the very operations
            // the compiler's expansion performs when no scaling is needed, i.e. for every operand a depth or a projected coordinate can take;
            // the two quotients by z' share the refined reciprocal. 24 instead of 33 instructions per point, the same quotients bit for bit
            // (tools/ab_bits.py: poses identical to the IEEE build).
            auto refine = [](float d) {
                const float r0 = __builtin_amdgcn_rcpf(d);
                return fmaf(fmaf(-d, r0, 1.0f), r0, r0);
            };
            auto quot = [](float n, float d, float r) {
                const float q0 = n * r;
                const float q1 = fmaf(fmaf(-d, q0, n), r, q0);
                // (v_div_fixup stays: a divisor of exactly 0, an infinity or a NaN must give the IEEE result — x / 0 = inf is OUTSIDE the image,
                // the NaN the bare sequence would produce converts to column 0)
                return __builtin_amdgcn_div_fixupf(fmaf(fmaf(-d, q1, n), r, q1), d, n);
            };
            float zc[G];
            V3 Pw[G];
#pragma unroll
            for (int g = 0; g < G; ++g) zc[g] = quot(1.0f, p.iz[g], refine(p.iz[g]));
#pragma unroll
            for (int g = 0; g < G; ++g) Pw[g] = iso_transform_point(model, back_project_rt(IntrFast{c.k, dfu, dfv}, p.px[g], p.py[g], zc[g]));
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float pu = (c.k.fu * Pw[g].x + c.k.skew * Pw[g].y) + c.k.cu * Pw[g].z;  // project: camera.rs:126-132, as lie.h project_uv
                const float pv = c.k.fv * Pw[g].y + c.k.cv * Pw[g].z;
                const float r = refine(Pw[g].z);
                u[g] = quot(pu, Pw[g].z, r);
                v[g] = quot(pv, Pw[g].z, r);
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const V3 P = XW ? back_project_rt(IntrFast{c.k, dfu, dfv}, p.px[g], p.py[g], 1.0f / p.iz[g]) : back_project(c.k, p.px[g], p.py[g], 1.0f / p.iz[g]);
                project_uv(c.k, iso_transform_point(model, P), &u[g], &v[g]);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iu[g]) : "v"(u[g]));  // (int)floor(u), saturating
#pragma unroll
    for (int g = 0; g < G; ++g) asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iv[g]) : "v"(v[g]));
    const unsigned wlim = (unsigned)(c.cols - 2), hlim = (unsigned)(c.rows - 2);
#pragma unroll
    for (int g = 0; g < G; ++g)  // the reference's strict window 0 <= floor(u) < w - 2, 0 <= floor(v) < h - 2 (lm_optimizer.rs:227-231);
                                 // +-inf saturate to an outside integer. (A NaN coordinate would need hu == hz == 0 exactly.)
        st.inside[g] = p.valid[g] && ((unsigned)iu[g] < wlim) && ((unsigned)iv[g] < hlim);
#pragma unroll
    for (int g = 0; g < G; ++g) off[g] = (iv[g] * c.cols + iu[g]) & (st.inside[g] ? -1 : 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const Taps t = load_taps_at(c, off[g]);
        st.top[g] = t.top;
        st.bot[g] = t.bot;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) st.fa[g] = __builtin_amdgcn_fractf(u[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) st.fb[g] = __builtin_amdgcn_fractf(v[g]);
    st.tmw = p.tmw;
    if (!ENERGY_ONLY) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            st.a[g] = p.a[g];
            st.b[g] = p.b[g];
            st.iz[g] = p.iz[g];
            st.gu[g] = p.gu[g];
            st.gv[g] = p.gv[g];
        }
    }
}
// Stage C: lerp-form bilinear interpolation, residual, Jacobian, the 29 sums (`cnt` = the lane's integer count of inside points).
template <bool HUBER, bool ENERGY_ONLY, int G>
__device__ __forceinline__ void fused_stage_c(const ImgCtx& c, const JacK& k, const FusedStage<G>& st, float acc[NACC], int& cnt) {
    uint16_t top[G], bot[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        top[g] = st.top[g];
        bot[g] = st.bot[g];
    }
    float t00[G], d0[G], t10[G], d1[G], tp[G], bt[G], im[G], r[G], tm[G];
    // (spelled as instructions: left to itself the compiler turns float(hi) - float(lo) into shift + SDWA subtract + convert, three
    // instructions, two of them of the slow class, and moves the zero-extension of a tap word next to its load, i.e. into stage B)
#pragma unroll
    for (int g = 0; g < G; ++g) t00[g] = cvt_ubyte<0>(top[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) t10[g] = cvt_ubyte<0>(bot[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) d0[g] = cvt_ubyte<1>(top[g]) - t00[g];
#pragma unroll
    for (int g = 0; g < G; ++g) d1[g] = cvt_ubyte<1>(bot[g]) - t10[g];
#pragma unroll
    for (int g = 0; g < G; ++g) tp[g] = fmaf(st.fa[g], d0[g], t00[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) bt[g] = fmaf(st.fa[g], d1[g], t10[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) tm[g] = (float)((st.tmw >> (8 * g)) & 0xff);
#pragma unroll
    for (int g = 0; g < G; ++g) im[g] = fmaf(st.fb[g], bt[g] - tp[g], tp[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) r[g] = st.inside[g] ? im[g] - tm[g] : 0.f;  // selected, never multiplied: outside contributes nothing
#pragma unroll
    for (int g = 0; g < G; ++g) cnt += st.inside[g] ? 1 : 0;  // (a wavefront-wide count on the scalar unit, s_bcnt1 of the mask: measured slower)
    float wgt[G], wr[G];
    if (HUBER) {  // extension (not in the reference)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float ar = fabsf(r[g]);
            const bool lin = ar > c.huber;
            acc[0] += lin ? c.huber * fmaf(2.0f, ar, -c.huber) : r[g] * r[g];
            wgt[g] = lin ? c.huber * __builtin_amdgcn_rcpf(ar) : 1.0f;
            wr[g] = wgt[g] * r[g];
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acc[0] = fmaf(r[g], r[g], acc[0]);
            wgt[g] = 1.0f;
            wr[g] = r[g];
        }
    }
    if (ENERGY_ONLY) return;
    // warp_jacobian_at (inverse_compositional.rs:313-341), linear in (gu, gv); an outside point gets gu = gv = 0 -> J = 0
    // (the per-row parts — everything that depends on b alone — are common subexpressions of the four pixels of a quad)
    // Two points at a time: ILP 2 already fills the pipe (tools/ubench/valu_occ) and 12 live Jacobian entries instead of 24 keep the
    // full evaluation inside its register budget.
    constexpr int W = G >= 2 ? 2 : 1;
#pragma unroll
    for (int h = 0; h < G; h += W) {
        float gu[W], gv[W], cp[W], b_fv[W], J[6][W];
#pragma unroll
        for (int g = 0; g < W; ++g) gu[g] = st.inside[h + g] ? st.gu[h + g] : 0.f;
#pragma unroll
        for (int g = 0; g < W; ++g) gv[g] = st.inside[h + g] ? st.gv[h + g] : 0.f;
#pragma unroll
        for (int g = 0; g < W; ++g) b_fv[g] = st.b[h + g] * k.inv_fv;
#pragma unroll
        for (int g = 0; g < W; ++g) cp[g] = fmaf(st.a[h + g], k.inv_fu, -(st.b[h + g] * k.s_fuv));  // c' = c / (fu fv), c = a fv - s b
#pragma unroll
        for (int g = 0; g < W; ++g) J[0][g] = (gu[g] * k.fu) * st.iz[h + g];
#pragma unroll
        for (int g = 0; g < W; ++g) J[1][g] = fmaf(gu[g], k.s, gv[g] * k.fv) * st.iz[h + g];
#pragma unroll
        for (int g = 0; g < W; ++g) J[2][g] = -(fmaf(gu[g], st.a[h + g], gv[g] * st.b[h + g]) * st.iz[h + g]);
#pragma unroll
        for (int g = 0; g < W; ++g) J[3][g] = fmaf(gu[g], fmaf(-st.a[h + g], b_fv[g], -k.s), gv[g] * fmaf(-st.b[h + g], b_fv[g], -k.fv));
#pragma unroll
        for (int g = 0; g < W; ++g) J[4][g] = fmaf(gu[g], fmaf(st.a[h + g], cp[g], k.fu), gv[g] * (st.b[h + g] * cp[g]));
#pragma unroll
        for (int g = 0; g < W; ++g) J[5][g] = fmaf(gu[g], fmaf(k.s, cp[g], -(k.fu * b_fv[g])), gv[g] * (cp[g] * k.fv));
#pragma unroll
        for (int g = 0; g < W; ++g) {  // 27 independent accumulators per point
#pragma unroll
            for (int q = 0; q < 6; ++q) acc[2 + q] = fmaf(J[q][g], wr[h + g], acc[2 + q]);
            int hh = 8;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const float jq = HUBER ? wgt[h + g] * J[q][g] : J[q][g];
#pragma unroll
                for (int s2 = q; s2 < 6; ++s2) {
                    acc[hh] = fmaf(jq, J[s2][g], acc[hh]);
                    ++hh;
                }
            }
        }
    }
}

// Fused point sources: the exact sources' loads, decoded into FUnit.
template <bool LEVEL0>
struct FusedPixSrc : DenseSrc<LEVEL0> {  // one pixel per unit, any width
    using Base = DenseSrc<LEVEL0>;
    static constexpr bool FUSED = true;
    template <bool ENERGY_ONLY>
    __device__ __forceinline__ void funit(const typename Base::Raw& r, const FusedCtx& f, FUnit<1>& p) const {
        const float xf = (float)r.x, yf = (float)r.y;
        p.bu[0] = fmaf(f.h00, xf, fmaf(f.h01, yf, f.h02));
        p.bv[0] = fmaf(f.h10, xf, fmaf(f.h11, yf, f.h12));
        p.bz[0] = fmaf(f.h20, xf, fmaf(f.h21, yf, f.h22));
        p.a[0] = xf - this->k.cu;
        p.b[0] = yf - this->k.cv;
        p.valid[0] = r.valid;
        p.iz[0] = r.valid ? r.izv : 0.f;
        p.tmw = (uint32_t)r.tm;
        p.gu[0] = (float)r.gx;
        p.gv[0] = (float)r.gy;
        p.px[0] = xf; p.py[0] = yf;
    }
};
template <bool LEVEL0>
struct FusedQuadSrc : DenseQuadSrc<LEVEL0, false> {  // four horizontally adjacent pixels per unit
    using Base = DenseQuadSrc<LEVEL0, false>;
    static constexpr bool FUSED = true;
    using Raw = typename Base::Loaded;  // decoded straight into FUnit
    float depth_scale;
    template <int BLOCK>
    __device__ __forceinline__ void fetch(const typename Base::Cursor& c, int, Raw& r) const {
        this->load(c, r);
    }
    template <bool ENERGY_ONLY>
    __device__ __forceinline__ void funit(const Raw& l, const FusedCtx& f, FUnit<4>& p) const {
        const float x0f = (float)l.x0, yf = (float)l.y;
        const float bu0 = fmaf(f.h00, x0f, fmaf(f.h01, yf, f.h02));
        const float bv0 = fmaf(f.h10, x0f, fmaf(f.h11, yf, f.h12));
        const float bz0 = fmaf(f.h20, x0f, fmaf(f.h21, yf, f.h22));
        const float a0 = x0f - this->kf.k.cu, b = yf - this->kf.k.cv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p.a[j] = a0 + (float)j;
            p.b[j] = b;
        }
        p.bu[0] = bu0; p.bu[1] = bu0 + f.h00; p.bu[2] = bu0 + f.h00_2; p.bu[3] = bu0 + f.h00_3;
        p.bv[0] = bv0; p.bv[1] = bv0 + f.h10; p.bv[2] = bv0 + f.h10_2; p.bv[3] = bv0 + f.h10_3;
        p.bz[0] = bz0; p.bz[1] = bz0 + f.h20; p.bz[2] = bz0 + f.h20_2; p.bz[3] = bz0 + f.h20_3;
        p.tmw = l.cw;
#pragma unroll
        for (int j = 0; j < 4; ++j) { p.px[j] = x0f + (float)j; p.py[j] = yf; }
        const int rows = this->rows, cols = this->cols;
        if (LEVEL0) {
            float dzf[4], rd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t dz = (j < 2 ? (l.d0 >> (16 * j)) : (l.d1 >> (16 * (j - 2)))) & 0xffffu;
                p.valid[j] = dz != 0;
                dzf[j] = (float)dz;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) rd[j] = __builtin_amdgcn_rcpf(dzf[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j)  // scale / depth (inverse_depth.rs:24-29) as scale * rcp(depth); an unknown depth (rcp = inf) is
                                         // clamped to a finite value (v_min is full rate, a select is not); `valid` masks the point
                p.iz[j] = (VORS_ABL_IEEE_IDEPTH || VORS_ABL_EXACT_WARP) ? fminf(depth_scale / dzf[j], 1e18f)
                                                                      : fminf(depth_scale * rd[j], 1e18f);  // (the exact table of the EXACT path as a gather here: +25-45 % on the round)
            if (!ENERGY_ONLY) {  // centred differences, truncating /2, zero on the 1-px border (gradient.rs:15-33): integer, exact
                const int yin = (l.y > 0 && l.y < rows - 1) ? -1 : 0;
                int tm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) tm[j] = (l.cw >> (8 * j)) & 0xff;
                const int bb[6] = {(int)l.w3, tm[0], tm[1], tm[2], tm[3], (int)l.w4};
                int gx[4], gy[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) gx[j] = half_trunc(bb[j + 2] - bb[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) gy[j] = half_trunc((int)((l.w2 >> (8 * j)) & 0xff) - (int)((l.w1 >> (8 * j)) & 0xff));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int x = l.x0 + j;
                    const int in = yin & ((x > 0 && x < cols - 1) ? -1 : 0);
                    p.gu[j] = (float)(gx[j] & in);
                    p.gv[j] = (float)(gy[j] & in);
                }
            }
        } else {
            const uint32_t zz[4] = {l.d0, l.d1, l.d2, l.d3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = __int_as_float((int)zz[j]);
                p.valid[j] = !(z != z);
                p.iz[j] = fminf(z, 1e18f);  // Unknown (NaN) -> finite; `valid` masks the point
            }
            if (!ENERGY_ONLY) {  // 2x2 block gradients of the next finer level (gradient.rs:74-93): integer, exact
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t w0 = j < 2 ? l.w1 : l.w2, w1 = j < 2 ? l.w3 : l.w4;
                    const int sh = (j & 1) * 16;
                    const int a = (w0 >> sh) & 0xff, cc = (w0 >> (sh + 8)) & 0xff;
                    const int b2 = (w1 >> sh) & 0xff, d = (w1 >> (sh + 8)) & 0xff;
                    p.gu[j] = (float)half_trunc(cc + d - a - b2);
                    p.gv[j] = (float)half_trunc(b2 - a + d - cc);
                }
            }
        }
    }
};

struct FusedSlimSrc : SlimSrc {  // compact 12-byte candidate lists, two points per unit
    using Base = SlimSrc;
    static constexpr bool FUSED = true;
    template <bool ENERGY_ONLY>
    __device__ __forceinline__ void funit(const Raw& r, const FusedCtx& f, FUnit<2>& p) const {
        float xf[2], yf[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            xf[g] = (float)(r.r[g].xy & 0xffffu);
            yf[g] = (float)(r.r[g].xy >> 16);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) p.bu[g] = fmaf(f.h00, xf[g], fmaf(f.h01, yf[g], f.h02));
#pragma unroll
        for (int g = 0; g < 2; ++g) p.bv[g] = fmaf(f.h10, xf[g], fmaf(f.h11, yf[g], f.h12));
#pragma unroll
        for (int g = 0; g < 2; ++g) p.bz[g] = fmaf(f.h20, xf[g], fmaf(f.h21, yf[g], f.h22));
        p.tmw = (r.r[0].tg & 0xffu) | ((r.r[1].tg & 0xffu) << 8);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            p.a[g] = xf[g] - this->k.cu;
            p.b[g] = yf[g] - this->k.cv;
            p.valid[g] = r.valid[g];
            p.iz[g] = r.r[g].iz;  // always a known inverse depth: the lists hold candidates only
            p.gu[g] = (float)slim_gx(r.r[g].tg);
            p.gv[g] = (float)slim_gy(r.r[g].tg);
            p.px[g] = xf[g]; p.py[g] = yf[g];
        }
    }
};

// The per-group body shared by both loop shapes: warp all G points, issue all taps, then Jacobians + sums two points at a
// time (keeps the live Jacobian registers at 12 while the tap loads are in flight).
template <bool HUBER, bool WRITE_RES, bool ENERGY_ONLY, class Src>
__device__ __forceinline__ void process_group(const Src& src, const typename Src::Raw& raw, int n_units, const ImgCtx& c,
                                              const Iso& model, float acc[NACC], float* residuals) {
    constexpr int G = Src::G;
    Pos pos[G];
    src.positions(raw, pos);
    if (Src::SKIP_EMPTY && !WRITE_RES) {  // a wavefront whose group holds no candidate at all (compacted empty tail) skips the arithmetic
        bool any_valid = false;
#pragma unroll
        for (int g = 0; g < G; ++g) any_valid = any_valid || (pos[g].tmpl >= 0.f);
        if (!__any(any_valid)) return;
    }
    Warped w[G];
#pragma unroll
    for (int g = 0; g < G; ++g) w[g] = warp_point(c, model, pos[g]);
    Taps t[G];
#pragma unroll
    for (int g = 0; g < G; ++g) t[g] = load_taps(c, w[g]);
#pragma unroll
    for (int g0 = 0; g0 < G; g0 += 2) {
        float J[2][6];
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (!ENERGY_ONLY && g0 + h < G) src.jacobian(raw, g0 + h, J[h]);  // independent of the taps: overlaps their latency
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (g0 + h < G) {
                const int g = g0 + h;
                const float res = accumulate_point<HUBER, ENERGY_ONLY>(c, pos[g].tmpl, J[h], w[g], t[g], acc);
                if (WRITE_RES) {
                    const int sl = src.slot(raw, g, n_units);
                    if (sl >= 0) residuals[sl] = res;
                }
            }
    }
}

// One evaluation sweep over the units of a level: each thread accumulates its strided share, Src::G points in flight.
// Sources with PREFETCH keep the raw words of the NEXT unit in flight while the current one is processed, so the wavefronts
// of a SIMD do not all stall on the same loads at the top of every iteration.
template <int BLOCK, bool HUBER, bool WRITE_RES, class Src, bool ENERGY_ONLY = false>
__device__ __forceinline__ void eval_accumulate(const Src& src, int n_units, const ImgCtx& c, const Iso& model, float acc[NACC],
                                                float* residuals, int first = 0, const LmSplitState* pre = nullptr) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    if constexpr (Src::FUSED) {
        static_assert(!WRITE_RES, "per-point residuals are an operator-level (exact arithmetic) output");
        // A (near-)identity model — every pair's first evaluation, and the first one after a keyframe switch — puts EVERY point on
        // (or within rounding of) integer coordinates, where the reference's own rounding decides on which side of the strict inside
        // test (lm_optimizer.rs:227-231) the border rows / columns fall: 1-3 % of the points at the coarsest level, and with them the
        // first energy and the path of the whole LM loop. Only the reference's evaluation order reproduces that, so such an
        // evaluation runs in the exact arithmetic (a handful per pair, mostly at the coarsest level).
        if (c.force_exact || (pre ? fused_ctx_exact(pre) : model_near_identity(model))) {
            eval_accumulate<BLOCK, HUBER, false, typename Src::Base, ENERGY_ONLY>(static_cast<const typename Src::Base&>(src), n_units, c, model, acc,
                                                                               nullptr, first);
            return;
        }
        const JacK jk = make_jack(c);
        int cnt = 0;  // inside points seen by this lane
        if constexpr (std::is_same<Src, FusedSlimSrc>::value) {
            if (c.exact_warp) {  // (workgroup-uniform) a level of few points: the reference's warp chain, the rest fused
                const FusedCtx f{};  // (the homography is not used)
                for (typename Src::Cursor cur = src.template begin<BLOCK>(first); cur.i < n_units; cur = src.template advance<BLOCK>(cur)) {
                    typename Src::Raw raw;
                    src.template fetch<BLOCK>(cur, n_units, raw);
                    FUnit<Src::G> un;
                    src.template funit<ENERGY_ONLY>(raw, f, un);
                    FusedStage<Src::G> st;
                    fused_stage_b<ENERGY_ONLY, true>(un, c, f, st, model, src.fu, src.fv);
                    fused_stage_c<HUBER, ENERGY_ONLY>(c, jk, st, acc, cnt);
                }
                acc[1] = (float)cnt;
                return;
            }
        }
        const FusedCtx f = pre ? load_fused_ctx(pre) : make_fused_ctx(c, model);  // (`pre` is a compile-time fact at every call site)
        for (typename Src::Cursor cur = src.template begin<BLOCK>(first); cur.i < n_units; cur = src.template advance<BLOCK>(cur)) {
            typename Src::Raw raw;
            src.template fetch<BLOCK>(cur, n_units, raw);
            FUnit<Src::G> un;
            src.template funit<ENERGY_ONLY>(raw, f, un);
            FusedStage<Src::G> st;
            fused_stage_b<ENERGY_ONLY>(un, c, f, st, model);
            fused_stage_c<HUBER, ENERGY_ONLY>(c, jk, st, acc, cnt);
        }
        acc[1] = (float)cnt;
    } else if constexpr (Src::PREFETCH) {
        typename Src::Cursor cur = src.template begin<BLOCK>(first);
        if (cur.i >= n_units) return;
        typename Src::Loaded ld;
        src.load(cur, ld);
        for (;;) {
            const typename Src::Cursor nxt = src.template advance<BLOCK>(cur);
            const bool more = nxt.i < n_units;
            typename Src::Loaded ld_next = ld;
            if (more) src.load(nxt, ld_next);
            typename Src::Raw raw;
            src.decode(ld, raw);
            process_group<HUBER, WRITE_RES, ENERGY_ONLY>(src, raw, n_units, c, model, acc, residuals);
            if (!more) break;
            ld = ld_next;
            cur = nxt;
        }
    } else {
        for (typename Src::Cursor cur = src.template begin<BLOCK>(first); cur.i < n_units; cur = src.template advance<BLOCK>(cur)) {
            typename Src::Raw raw;
            src.template fetch<BLOCK>(cur, n_units, raw);
            process_group<HUBER, WRITE_RES, ENERGY_ONLY>(src, raw, n_units, c, model, acc, residuals);
        }
    }
}

// ---- wavefront reduction with DPP (no LDS traffic): after the 6 steps lane 63 holds the sum of all 64 lanes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);  // row_mirror  -> every lane of a row holds the row sum
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    return v;
}

// Sum over the 16 lanes of a DPP row; every lane of the row ends up with it. (bound_ctrl + full masks: the compiler folds each
// v_mov_dpp into its v_add_f32, one instruction per step.)
template <int CTRL>
__device__ __forceinline__ float dpp_add_row(float v) {
    const int t = __builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add_row<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add_row<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add_row<0x141>(v);  // row_half_mirror
    v = dpp_add_row<0x140>(v);  // row_mirror
    return v;
}
// Workgroup reduction of the 29 partial sums into s.sums[dst][0..28]. Round 3 measured this at ~28 % of the per-pair kernel in the sparse
// modes (probes by doubling; tools/ubench/dpp_ops: a DPP add issues every 4.4 cycles, not 2.8, and each of the 29 LDS stores by 4 of 64
// lanes costs 10-16), so it is a TRANSPOSING reduction now: the two in-quad steps halve the number of values a lane carries — an even
// lane keeps sum 2j and gives 2j+1 to its neighbour, and so on: a lane of quad position q ends with the 8 sums v = 4 k + q (two selects
// + one DPP add per output; DPP bank masks select quads, not lanes, so the selects cannot be folded away) — then two row_shl steps add
// the four quads of a row into its first quad, whose 4 lanes hand the row's 32 sums over with two 16-byte LDS stores; 29 threads add
// the BLOCK / 16 rows in index order as before. 88 VALU instructions + 2 stores per wavefront instead of 116 + 29, and the SAME
// floating-point result bit for bit: every sum is still ((l0 + l1) + (l2 + l3)) within a quad and (Q0 + Q1) + (Q2 + Q3) within a row
// (tools/ubench/reduce_check.hip compares the two forms). Deterministic for a given BLOCK. Ends with a barrier.
template <int BLOCK>
__device__ __forceinline__ void block_reduce(const float acc[NACC], LmShared& s, int dst) {
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
        float4* p = reinterpret_cast<float4*>(s.part + (threadIdx.x >> 4) * 32 + (threadIdx.x & 3) * 8);
        p[0] = make_float4(c[0], c[1], c[2], c[3]);
        p[1] = make_float4(c[4], c[5], c[6], c[7]);
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        const int slot = (threadIdx.x & 3) * 8 + (threadIdx.x >> 2);
        float t = 0.f;
#pragma unroll 8
        for (int w = 0; w < BLOCK / 16; ++w) t += s.part[w * 32 + slot];
        s.sums[dst][threadIdx.x] = t;
    }
    __syncthreads();
}

template <int BLOCK>
__device__ __forceinline__ void block_sum2(float& a, float& b, LmShared& s);  // (defined after solve_level)

// step(): lm_optimizer.rs:123-136, by ONE lane on the kept state's sums; result broadcast through LDS.
// FUSED arithmetic: step() (lm_optimizer.rs:123-136) with the 6x6 Cholesky factor scaled by reciprocal square roots and the two
// triangular solves by the stored reciprocals (6 v_rsq + multiplies instead of 6 IEEE square roots and 33 IEEE divisions, FMAs
// contracted), and from_quaternion through one v_rsq. The one-lane step is ~40 % of the instructions the per-pair kernel issues in the
// sparse modes (a few hundred points per evaluation); this form is about half as long. Same pivot rule (a pivot that is not > 0 fails).
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
template <bool FAST>
__device__ __forceinline__ void solve_step_lane0(LmShared& s, int cur, const Iso& model, float lm_coef, bool exact_step = false) {
    if (threadIdx.x == 0) {
        const float* a = s.sums[cur];
        float h[36], g[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) g[q] = a[2 + q];
        int k = 8;
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int r = q; r < 6; ++r) {
                h[q * 6 + r] = a[k];
                h[r * 6 + q] = a[k];
                ++k;
            }
        Iso cand;
        bool ok;
#ifdef VORS_DEV_SOLVE_REPEAT  // (development probe: how much of an evaluation's latency is the one-lane step)
        for (int rep = 1; rep < VORS_DEV_SOLVE_REPEAT; ++rep) {
            Iso tmp;
            if (lm_step_fast(h, g, model, lm_coef + (float)rep * 1e-30f, &tmp)) h[0] += tmp.t.x * 1e-30f;
        }
#endif
        if constexpr (FAST && !VORS_ABL_EXACT_STEP) {
            if (exact_step) ok = lm_step(h, g, model, lm_coef, &cand);  // (uniform; Geom::fused_exact_step, a development knob)
            else ok = lm_step_fast(h, g, model, lm_coef, &cand);
        } else {
            ok = lm_step(h, g, model, lm_coef, &cand);
        }
        iso_store(cand, s.cand);
        s.cand[7] = ok ? 1.0f : 0.0f;
    }
    __syncthreads();
}

// optimizer::State::iterative_solve (optimizer.rs:57-70) with LMOptimizerState's init / step / eval / stop_criterion
// (lm_optimizer.rs:113-192). All control values are workgroup-uniform (read from LDS). Returns false when step() fails
// (Cholesky); *model is then left untouched: the level's progress is discarded like the reference's `Err(err) => break`
// (inverse_compositional.rs:195-199).
template <int BLOCK, bool HUBER, class Src>
__device__ bool solve_level(const Src& src, int n_slots, const ImgCtx& c, Iso* model, int* nb_iter_out, float* energy_out,
                            float* lm_coef_out, LmShared& s, const LmSplitState* resume = nullptr, int* n_full_out = nullptr, long long* ph = nullptr) {
#ifdef VORS_PROFILE_PHASES  // (development probe: shader-clock cycles of a level spent in the point loop / the reduction / the one-lane step)
#define VORS_PH(k) do { const long long t_now = clock64(); if (ph) ph[k] += t_now - t_ph; t_ph = t_now; } while (0)
    long long t_ph = clock64();
#else
#define VORS_PH(k) do { } while (0)
#endif
    float acc[NACC];
    Iso cur_model = *model;
    int cur = 0;
#ifndef VORS_TWO_PHASE_UNITS
#define VORS_TWO_PHASE_UNITS 1024  // (measured at 640x480: 2048 -> coarse levels 1.70 ms, 1024 -> 1.60 ms, 256 -> 1.68 ms)
#endif
    constexpr bool TWO_PHASE = Src::FUSED && Src::G == 4;  // dense quad levels of the fused arithmetic
    const bool two_phase = TWO_PHASE && n_slots >= VORS_TWO_PHASE_UNITS;
    float cur_energy, lm_coef = 0.1f;
    int nb_iter = 0;
    Iso cand = cur_model;
    bool have_cand = false;
    int n_full = resume ? __builtin_amdgcn_readfirstlane(resume->n_full) : 1;  // the initial evaluation + every accepted candidate (statistics)
    if (resume) {  // split path: the level was started by the chip-wide launches; a candidate is waiting for its evaluation
        if (threadIdx.x < NACC) s.sums[0][threadIdx.x] = resume->sums[threadIdx.x];
        __syncthreads();
        cur_model = iso_uniform(iso_load(resume->model));
        cand = iso_uniform(iso_load(resume->cand));
        cur_energy = uniform_f(resume->cur_energy);
        lm_coef = uniform_f(resume->lm_coef);
        nb_iter = __builtin_amdgcn_readfirstlane(resume->nb_iter);
        have_cand = true;
        if (__builtin_amdgcn_readfirstlane(resume->phase) == 4) {  // the kept model is an accepted candidate whose g, H are due
            eval_accumulate<BLOCK, HUBER, false>(src, n_slots, c, cur_model, acc, nullptr);
            block_reduce<BLOCK>(acc, s, cur);
            have_cand = false;
        }
    } else {
        VORS_PH(3);
        eval_accumulate<BLOCK, HUBER, false>(src, n_slots, c, cur_model, acc, nullptr);  // init: lm_optimizer.rs:113-118
        VORS_PH(0);
        block_reduce<BLOCK>(acc, s, cur);
        VORS_PH(1);
        cur_energy = uniform_f(s.sums[cur][0] / s.sums[cur][1]);  // energy_sum / residuals.len(): 0/0 = NaN like the reference
    }
    for (;;) {
        if (!have_cand) {
            nb_iter += 1;
            VORS_PH(3);
            solve_step_lane0<Src::FUSED>(s, cur, cur_model, lm_coef, c.exact_step);  // step(): lm_optimizer.rs:123-136
            VORS_PH(2);
            if (uniform_f(s.cand[7]) == 0.0f) return false;
            cand = iso_uniform(iso_load(s.cand));  // workgroup-uniform: keep it in scalar registers
        }
        have_cand = false;
        const bool too_many_iterations = nb_iter > 20;  // stop_criterion: lm_optimizer.rs:156-192
        if constexpr (TWO_PHASE) {
            if (two_phase) {  // (uniform) large dense level: a candidate's energy alone first, like eval_energy (lm_optimizer.rs:68-87) ...
                eval_accumulate<BLOCK, HUBER, false, Src, true>(src, n_slots, c, cand, acc, nullptr);
                float e_sum = acc[0], e_cnt = acc[1];
                block_sum2<BLOCK>(e_sum, e_cnt, s);
                const float energy = uniform_f(e_sum / e_cnt);
                if (energy > cur_energy) {
                    if (too_many_iterations) break;
                    lm_coef *= 10.0f;
                    continue;
                }
                const float d_energy = cur_energy - energy;
                n_full += 1;
                cur_energy = energy;
                cur_model = cand;
                if (too_many_iterations) break;
                lm_coef = 0.1f * lm_coef;
                if (!(d_energy > 1.0f)) break;
                // ... and its g and H (compute_eval_data, :90-107) only when the level goes on from it
                eval_accumulate<BLOCK, HUBER, false>(src, n_slots, c, cand, acc, nullptr);
                block_reduce<BLOCK>(acc, s, 1 - cur);
                cur = 1 - cur;
                continue;
            }
        }
        VORS_PH(3);
        eval_accumulate<BLOCK, HUBER, false>(src, n_slots, c, cand, acc, nullptr);  // eval(): lm_optimizer.rs:140-149
        VORS_PH(0);
        block_reduce<BLOCK>(acc, s, 1 - cur);
        VORS_PH(1);
        const float energy = uniform_f(s.sums[1 - cur][0] / s.sums[1 - cur][1]);
        if (energy > cur_energy) {                      // Err(energy)
            if (too_many_iterations) break;
            lm_coef *= 10.0f;
            continue;
        }
        const float d_energy = cur_energy - energy;
        n_full += 1;    // accepted: the reference forms its g and H (lm_optimizer.rs:147)
        cur = 1 - cur;  // the candidate's sums become the kept state (no copy)
        cur_energy = energy;
        cur_model = cand;
        if (too_many_iterations) break;
        lm_coef = 0.1f * lm_coef;
        if (!(d_energy > 1.0f)) break;
    }
    *model = cur_model;
    *nb_iter_out = nb_iter;
    *energy_out = cur_energy;
    *lm_coef_out = lm_coef;
    if (n_full_out) *n_full_out = n_full;
    return true;
}

// Sum of two per-thread values over the workgroup (keyframe test, point counts). Ends with a barrier; result uniform.
template <int BLOCK>
__device__ __forceinline__ void block_sum2(float& a, float& b, LmShared& s) {
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(a)), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(b)), 63));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        s.misc[(threadIdx.x >> 6) * 2] = a;
        s.misc[(threadIdx.x >> 6) * 2 + 1] = b;
    }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        ta += s.misc[2 * w];
        tb += s.misc[2 * w + 1];
    }
    a = ta;
    b = tb;
}

// ---- split path bookkeeping (see engine.h, LmSplitWs)
#define SPLIT_COUNT_STRIDE (VORS_SPLIT_MAX_ROUNDS + 2)
// a-th active pair of a round: the full-kind pairs first (front of the array), then the energy-kind ones (back of the array)
__device__ __forceinline__ int split_active(const LmSplitWs& ws, int round, int a) {
    const int nf = ws.count[round];
    return a < nf ? ws.list[round & 1][a] : ws.list[round & 1][ws.cap - 1 - (a - nf)];
}
__device__ __forceinline__ int split_n_active(const LmSplitWs& ws, int round) { return ws.count[round] + ws.count[SPLIT_COUNT_STRIDE + round]; }
__device__ __forceinline__ void split_append(const LmSplitWs& ws, int round, bool energy_kind, int pair) {
    if (energy_kind) ws.list[round & 1][ws.cap - 1 - atomicAdd(&ws.count[SPLIT_COUNT_STRIDE + round], 1)] = pair;
    else ws.list[round & 1][atomicAdd(&ws.count[round], 1)] = pair;
}

// ------------------------------------------------------------------------------------------------------------
// Tracker::track for a batch: one workgroup per frame pair, all levels, all LM iterations, keyframe test.
// ------------------------------------------------------------------------------------------------------------
// Per-level dispatch: build the point source of a level and run `f(src, n_slots)`.
template <bool DENSE, bool QUADS, bool FUSED, class F>
__device__ __forceinline__ void with_level_source(const Geom& g, int lvl, int pair, const uint8_t* kf0, const uint8_t* kfu,
                                                  const uint16_t* kf_depth, const Records& rec, F&& f) {
    const LevelGeom lg = g.lv[lvl];
    if constexpr (DENSE && FUSED) {
        const uint8_t* kimg = level_ptr(g, kf0, kfu, pair, lvl);
        const uint8_t* kfine = lvl > 0 ? level_ptr(g, kf0, kfu, pair, lvl - 1) : nullptr;
        const uint16_t* depth = kf_depth + (size_t)pair * g.S0;
        const float* iz = lvl > 0 ? rec.IZ + (size_t)pair * g.slots_total + lg.slot_off : nullptr;
        const int fcols = lvl > 0 ? g.lv[lvl - 1].cols : 0;
        const bool quad_ok = QUADS && g.wide_loads_ok && (lg.cols % 4 == 0) && (lvl == 0 ? (g.S0 % 4 == 0) : (fcols % 8 == 0));
        if (lvl == 0) {
            if (quad_ok) {
                FusedQuadSrc<true> src{{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.cols / 4, IntrFast{lg.k, lg.fu, lg.fv}, rec.LUT}, g.depth_scale};
                f(src, lg.rows * (lg.cols / 4));
            } else {
                FusedPixSrc<true> src{{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.k, g.depth_scale}};
                f(src, lg.n_slots);
            }
        } else {
            if (quad_ok) {
                FusedQuadSrc<false> src{{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.cols / 4, IntrFast{lg.k, lg.fu, lg.fv}, rec.LUT}, g.depth_scale};
                f(src, lg.rows * (lg.cols / 4));
            } else {
                FusedPixSrc<false> src{{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.k, g.depth_scale}};
                f(src, lg.n_slots);
            }
        }
    } else if constexpr (DENSE) {
        const uint8_t* kimg = level_ptr(g, kf0, kfu, pair, lvl);
        const uint8_t* kfine = lvl > 0 ? level_ptr(g, kf0, kfu, pair, lvl - 1) : nullptr;
        const uint16_t* depth = kf_depth + (size_t)pair * g.S0;
        const float* iz = lvl > 0 ? rec.IZ + (size_t)pair * g.slots_total + lg.slot_off : nullptr;
        const int fcols = lvl > 0 ? g.lv[lvl - 1].cols : 0;
        // quads need 4-byte aligned rows in every plane they read (and 16-byte aligned inverse-depth rows)
        const bool quad_ok = QUADS && g.wide_loads_ok && (lg.cols % 4 == 0) && (lvl == 0 ? (g.S0 % 4 == 0) : (fcols % 8 == 0));
        if (lvl == 0) {
            if constexpr (QUADS) {
                if (quad_ok) {
                    if (lg.fu.ok && lg.fv.ok) {  // workgroup-uniform: selects the instantiation, no branch in the hot loop
                        DenseQuadSrc<true, true> src{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.cols / 4, IntrFast{lg.k, lg.fu, lg.fv}, rec.LUT};
                        f(src, lg.rows * (lg.cols / 4));
                    } else {
                        DenseQuadSrc<true, false> src{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.cols / 4, IntrFast{lg.k, lg.fu, lg.fv}, rec.LUT};
                        f(src, lg.rows * (lg.cols / 4));
                    }
                    return;
                }
            }
            DenseSrc<true> src{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.k, g.depth_scale};
            f(src, lg.n_slots);
        } else {
            if constexpr (QUADS) {
                if (quad_ok) {
                    if (lg.fu.ok && lg.fv.ok) {
                        DenseQuadSrc<false, true> src{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.cols / 4, IntrFast{lg.k, lg.fu, lg.fv}, rec.LUT};
                        f(src, lg.rows * (lg.cols / 4));
                    } else {
                        DenseQuadSrc<false, false> src{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.cols / 4, IntrFast{lg.k, lg.fu, lg.fv}, rec.LUT};
                        f(src, lg.rows * (lg.cols / 4));
                    }
                    return;
                }
            }
            DenseSrc<false> src{kimg, kfine, depth, iz, lg.rows, lg.cols, fcols, lg.k, g.depth_scale};
            f(src, lg.n_slots);
        }
    } else {
        const SlimRec* S = rec.S + (size_t)pair * g.slots_total + lg.slot_off;
        const int n = __builtin_amdgcn_readfirstlane(rec.n_used[(size_t)pair * VORS_MAX_LEVELS + lvl]);
        if constexpr (FUSED) {
            FusedSlimSrc src{{S, lg.k, lg.fu, lg.fv}};
            f(src, n);
        } else {
            SlimSrc src{S, lg.k, lg.fu, lg.fv};
            f(src, n);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Tracker::track for a batch: one workgroup per frame pair, all levels, all LM iterations, keyframe test.
// ------------------------------------------------------------------------------------------------------------
template <int BLOCK, bool HUBER, bool DENSE, bool FUSED>
// Register budget: with 256-thread workgroups more resident workgroups per CU hide the latency-bound coarse levels of their
// neighbours (measured at 4096 pairs: dense 5 waves/SIMD = 96 VGPRs +3.7 %, 6 spills; sparse 6 waves/SIMD +4 %).
#ifndef VORS_LM_WAVES
#define VORS_LM_WAVES (BLOCK <= 256 ? (DENSE ? 5 : 6) : (BLOCK == 512 ? 2 : 4))
#endif
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(VORS_LM_WAVES))) void lm_track_kernel(Geom g, const uint8_t* __restrict__ cur0, const uint8_t* __restrict__ curu,
                                                          const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                          const uint16_t* __restrict__ kf_depth, Records rec,
                                                          const float* __restrict__ prev_poses7, const float* __restrict__ kf_poses7,
                                                          float* __restrict__ out_poses7, int32_t* __restrict__ out_status,
                                                          vors_pair_stats* __restrict__ out_stats, int mode, LmSplitWs split) {
    // mode 0: the whole track() of the pair. Split path (dense): mode 1 = levels L-1 .. split.n_split, hands the model over
    // through split.state; mode 3 = workgroup a finishes the a-th pair still iterating after the last evaluation round (a big
    // workgroup each, all of them in parallel); mode 2 = takes the results back: pose + keyframe test + statistics (and
    // finishes whatever mode 3 could not take).
    __shared__ LmShared s;
#ifdef VORS_PROFILE_LEVELS
    const long long t_kernel0 = wall_clock64();
#endif
    int pair = blockIdx.x;
    if (mode == 3) {
        if ((int)blockIdx.x >= split_n_active(split, split.rounds)) return;
        pair = __builtin_amdgcn_readfirstlane(split_active(split, split.rounds, blockIdx.x));
    }
    if (mode == 4) {  // side lane (engine.h LmSplitWs): workgroup k finishes the levels above 0 of the k-th pair handed over by the step kernel
        if ((int)blockIdx.x >= split.count[SPLIT_SIDE_COUNT]) return;
        pair = __builtin_amdgcn_readfirstlane(split.side_list[blockIdx.x]);
    }
    const Iso prev_pose = prev_poses7 ? iso_load(prev_poses7 + 7 * pair) : iso_identity();
    const Iso kf_pose = kf_poses7 ? iso_load(kf_poses7 + 7 * pair) : iso_identity();
    Iso lm_model = iso_uniform(iso_mul(iso_inverse(prev_pose), kf_pose));  // inverse_compositional.rs:177
    bool went_well = true;
    int start_lvl = g.L - 1;
    const LmSplitState* resume = nullptr;
    if (mode >= 2) {  // the finest levels have been solved by the evaluation rounds — or are finished here (stragglers)
        const LmSplitState* st = split.state + pair;
        went_well = st->went_well != 0;
        start_lvl = -1;
        if (went_well && st->phase != 2) {  // still iterating after the last round
            start_lvl = st->lvl;
            lm_model = iso_uniform(iso_load(st->entry));  // stays the result if step() fails
            if (st->phase == 1 || st->phase == 4) resume = st;
        } else {
            lm_model = iso_uniform(iso_load(st->model));
        }
    }
    for (int lvl = start_lvl; lvl >= (mode == 1 ? split.n_split : (mode == 4 ? 1 : 0)); --lvl) {
        ImgCtx c;
        c.img = level_ptr(g, cur0, curu, pair, lvl);
        c.rows = g.lv[lvl].rows;
        c.cols = g.lv[lvl].cols;
        c.k = g.lv[lvl].k;
        c.huber = g.huber_delta;
        c.inv_fu_d = g.lv[lvl].inv_fu_d; c.inv_fv_d = g.lv[lvl].inv_fv_d;
        c.inv_fu = g.lv[lvl].inv_fu; c.inv_fv = g.lv[lvl].inv_fv; c.s_fuv = g.lv[lvl].s_fuv;
        if constexpr (FUSED) {
            // Levels of FEW points run in the EXACT arithmetic (Geom::fused_exact_points): the energy of a few hundred points
            // carries the per-point rounding of the fused warp (~1 ulp of u, v) at 3-10x the reference's own summation noise, enough to
            // fork the accept / stop comparisons of the coarsest levels measurably more often (tools/parity_ablate.py); from a few
            // thousand points on it averages out below that noise. (Workgroup-uniform.)
            const int npts = DENSE ? g.lv[lvl].rows * g.lv[lvl].cols
                                   : __builtin_amdgcn_readfirstlane(rec.n_used[(size_t)pair * VORS_MAX_LEVELS + lvl]);
            const bool few = npts <= g.fused_exact_points;
            // candidate lists: the reference's warp chain alone (Geom::fused_small_warp, the default); dense pixels levels (and
            // VORS_FUSED_SMALL=exact): the whole EXACT evaluation
            c.exact_warp = few && !DENSE && g.fused_small_warp != 0;
            c.force_exact = few && !c.exact_warp;
            c.exact_step = few && g.fused_exact_step != 0;
        }
        int nb_iter = 0, n_full = 0;
        float energy = 0.f, lm_coef = 0.f;
        bool ok = false;
#ifdef VORS_PROFILE_LEVELS
        const long long t_level0 = wall_clock64();
        const long long c_level0 = clock64();
#endif
        with_level_source<DENSE, true, FUSED>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n_slots) {
#ifdef VORS_PROFILE_PHASES
            long long ph[4] = {0, 0, 0, 0};
            ok = solve_level<BLOCK, HUBER>(src, n_slots, c, &lm_model, &nb_iter, &energy, &lm_coef, s, lvl == start_lvl ? resume : nullptr, &n_full, ph);
            if (out_stats && threadIdx.x == 0 && lvl == VORS_PROFILE_PHASES) {  // phases of ONE level, cycles: loop, reduce, step, other
                out_stats[pair].n_points[VORS_MAX_LEVELS - 2] = (int)ph[0];
                out_stats[pair].n_points[VORS_MAX_LEVELS - 1] = (int)ph[1];
                out_stats[pair].nb_iter[VORS_MAX_LEVELS - 2] = (int)ph[2];
                out_stats[pair].nb_iter[VORS_MAX_LEVELS - 1] = (int)ph[3];
            }
#else
            ok = solve_level<BLOCK, HUBER>(src, n_slots, c, &lm_model, &nb_iter, &energy, &lm_coef, s, lvl == start_lvl ? resume : nullptr, &n_full);
#endif
        });
        if (out_stats && threadIdx.x == 0) {
            out_stats[pair].nb_iter[lvl] = ok ? nb_iter : 0;
            out_stats[pair].nb_grad_evals[lvl] = ok ? n_full : 0;
            out_stats[pair].energy[lvl] = ok ? energy : 0.f;
#ifdef VORS_PROFILE_LEVELS
            out_stats[pair].energy[lvl] = (float)(wall_clock64() - t_level0) * 0.01f;  // 100 MHz ticks -> microseconds
#ifdef VORS_PROFILE_CLOCK
            out_stats[pair].energy[lvl] = (float)(clock64() - c_level0) / ((float)(wall_clock64() - t_level0) * 10.0f);  // shader GHz
#endif
#endif
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
    if (mode == 4) {  // level 0 is the rounds' again: hand the pair over like mode 1 does
        if (threadIdx.x == 0) {
            LmSplitState* st = split.state + pair;
            iso_store(lm_model, st->model);
            st->went_well = went_well ? 1 : 0;
            if (went_well) {
                iso_store(lm_model, st->entry);
                st->lvl = 0;
                st->phase = 0;
                st->nb_iter = 0;
                st->n_full = 0;
                if (FUSED) store_fused_ctx(g, 0, lm_model, st);
                split.join_list[atomicAdd(&split.count[SPLIT_JOIN_COUNT], 1)] = pair;
            } else {
                st->phase = 2;
            }
        }
        return;
    }
    if (mode == 3) {  // hand the finished pair back to the bookkeeping of mode 2
        if (threadIdx.x == 0) {
            LmSplitState* st = split.state + pair;
            iso_store(lm_model, st->model);
            st->went_well = went_well ? 1 : 0;
            st->phase = 2;
        }
        return;
    }
    if (mode == 1) {
        if (threadIdx.x == 0) {
            LmSplitState* st = split.state + pair;
            iso_store(lm_model, st->model);
            iso_store(lm_model, st->entry);
            st->lvl = split.n_split - 1;
            st->phase = 0;
            st->nb_iter = 0;
            st->n_full = 0;
            st->went_well = went_well ? 1 : 0;
            if (FUSED && went_well) store_fused_ctx(g, split.n_split - 1, lm_model, st);  // for the workgroups of the first round
            if (went_well) split_append(split, 0, false, pair);
        }
        return;
    }
#ifdef VORS_PROFILE_LEVELS
    const long long t_epilogue0 = wall_clock64();
#endif
    // keyframe test on the coarsest level (inverse_compositional.rs:211-224), with the last lm_model even after a failure
    float flow_sum = 0.f, flow_n = 0.f;
    {
        const int lvl = g.L - 1;
        const Intr k = g.lv[lvl].k;
        if constexpr (DENSE) {
            with_level_source<true, false, false>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n_slots) {
                for (auto cu = src.template begin<BLOCK>(); cu.i < n_slots; cu = src.template advance<BLOCK>(cu)) {
                    typename std::remove_reference<decltype(src)>::type::Raw r;
                    src.template fetch<BLOCK>(cu, n_slots, r);
                    Pos p[1];
                    src.positions(r, p);
                    if (p[0].tmpl >= 0.f) {
                        float u, v;
                        project_uv(k, iso_transform_point(lm_model, V3{p[0].X, p[0].Y, p[0].Z}), &u, &v);
                        flow_sum += fabsf((float)cu.x - u) + fabsf((float)cu.y - v);
                        flow_n += 1.0f;
                    }
                }
            });
        } else {
            const SlimRec* S = rec.S + (size_t)pair * g.slots_total + g.lv[lvl].slot_off;
            const int n_slots = rec.n_used[(size_t)pair * VORS_MAX_LEVELS + lvl];
            for (int i = threadIdx.x; i < n_slots; i += BLOCK) {
                const SlimRec r = S[i];
                const float x = (float)(r.xy & 0xffffu), y = (float)(r.xy >> 16);
                float u, v;
                project_uv(k, iso_transform_point(lm_model, back_project(k, x, y, 1.0f / r.iz)), &u, &v);  // warp: lm_optimizer.rs:213-219
                flow_sum += fabsf(x - u) + fabsf(y - v);
                flow_n += 1.0f;
            }
        }
        block_sum2<BLOCK>(flow_sum, flow_n, s);
    }
    if (threadIdx.x == 0) {
        const float optical_flow = flow_sum / flow_n;
        const Iso pose = went_well ? iso_mul(kf_pose, iso_inverse(lm_model)) : prev_pose;  // inverse_compositional.rs:206-208
        iso_store(pose, out_poses7 + 7 * pair);
        out_status[pair] = went_well ? VORS_TRACK_OK : VORS_TRACK_OPTIMIZER_FAILED_POSE_KEPT;
        if (out_stats) {
            iso_store(lm_model, out_stats[pair].lm_model);
            out_stats[pair].optical_flow = optical_flow;
            out_stats[pair].change_keyframe = (optical_flow >= 1.0f) ? 1 : 0;
        }
    }
    // usable candidates per level (diagnostics for the byte model)
    if (out_stats) {
        for (int lvl = 0; lvl < g.L; ++lvl) {
            const LevelGeom lg = g.lv[lvl];
            float n = 0.f, dummy = 0.f;
            if constexpr (DENSE) {
                if (rec.n_used && g.L >= 2) {  // counted by the keyframe stage
                    if (threadIdx.x == 0) n = (float)rec.n_used[(size_t)pair * VORS_MAX_LEVELS + lvl];
                } else if (lvl == 0) {
                    const uint16_t* d = kf_depth + (size_t)pair * g.S0;
                    if ((g.S0 & 7) == 0) {  // 8 depth values per 16-byte load
                        const uint4* d8 = reinterpret_cast<const uint4*>(d);
                        for (int i = threadIdx.x; i < lg.n_slots / 8; i += BLOCK) {
                            const uint4 w = d8[i];
                            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k) n += ((ww[k] & 0xffffu) ? 1.0f : 0.f) + ((ww[k] >> 16) ? 1.0f : 0.f);
                        }
                    } else {
                        for (int i = threadIdx.x; i < lg.n_slots; i += BLOCK) n += (d[i] != 0) ? 1.0f : 0.f;
                    }
                } else {
                    const float* z = rec.IZ + (size_t)pair * g.slots_total + lg.slot_off;
                    for (int i = threadIdx.x; i < lg.n_slots; i += BLOCK) {
                        const float v = z[i];
                        n += (v != v) ? 0.f : 1.0f;
                    }
                }
            } else {
                if (threadIdx.x == 0) n = (float)rec.n_used[(size_t)pair * VORS_MAX_LEVELS + lvl];
            }
            block_sum2<BLOCK>(n, dummy, s);
            if (threadIdx.x == 0) out_stats[pair].n_points[lvl] = (int)n;
        }
#ifndef VORS_PROFILE_PHASES
        if (threadIdx.x == 0)
            for (int lvl = g.L; lvl < VORS_MAX_LEVELS; ++lvl) {
                out_stats[pair].nb_iter[lvl] = 0;
                out_stats[pair].nb_grad_evals[lvl] = 0;
                out_stats[pair].n_points[lvl] = 0;
                out_stats[pair].energy[lvl] = 0.f;
            }
#endif
#ifdef VORS_PROFILE_LEVELS
        if (threadIdx.x == 0 && g.L <= VORS_MAX_LEVELS - 2) {
            out_stats[pair].energy[VORS_MAX_LEVELS - 2] = (float)(wall_clock64() - t_epilogue0) * 0.01f;  // epilogue, microseconds
            out_stats[pair].energy[VORS_MAX_LEVELS - 1] = (float)(wall_clock64() - t_kernel0) * 0.01f;    // whole workgroup
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------------------------
// Split path (dense mode, finest levels): one launch per energy evaluation over (active pairs x chunks), one small launch
// per step. Same arithmetic per point and the same control flow as solve_level; the sums are formed per chunk (fixed
// order: thread -> wavefront -> workgroup -> chunks in index order), so results are deterministic.
// ------------------------------------------------------------------------------------------------------------
#define SPLIT_BLOCK 256
__device__ __forceinline__ ImgCtx level_ctx(const Geom& g, const uint8_t* cur0, const uint8_t* curu, int pair, int lvl) {
    ImgCtx c;
    c.img = level_ptr(g, cur0, curu, pair, lvl);
    c.rows = g.lv[lvl].rows;
    c.cols = g.lv[lvl].cols;
    c.k = g.lv[lvl].k;
    c.huber = g.huber_delta;
    c.inv_fu_d = g.lv[lvl].inv_fu_d; c.inv_fv_d = g.lv[lvl].inv_fv_d;
    c.inv_fu = g.lv[lvl].inv_fu; c.inv_fv = g.lv[lvl].inv_fv; c.s_fuv = g.lv[lvl].s_fuv;
    return c;
}
#ifndef VORS_SPLIT_WAVES
#define VORS_SPLIT_WAVES 5
#endif
__device__ __forceinline__ int split_chunks(const LmSplitWs& ws, int lvl) { return max(1, ws.chunks0 >> (2 * lvl)); }

#ifndef VORS_SPLIT_ENERGY_WAVES
#define VORS_SPLIT_ENERGY_WAVES 8
#endif
// fused arithmetic: measured best at the same occupancies (3 / 4 wavefronts per SIMD without any spill: 10-20 % slower)
#ifndef VORS_FSPLIT_WAVES
#define VORS_FSPLIT_WAVES 5
#endif
#ifndef VORS_FSPLIT_ENERGY_WAVES
#define VORS_FSPLIT_ENERGY_WAVES 8
#endif
// ENERGY = false: full evaluation (energy, g, H) of the pairs at the front of the round's list, at their kept model (init of a
// level, or the g / H of an accepted candidate) or — late rounds — at their candidate. ENERGY = true: a candidate's energy alone
// (eval_energy, lm_optimizer.rs:68-87: a third fewer instructions per point and a third of the registers, hence its own
// launch at a higher occupancy) for the pairs at the back of the list; g and H (compute_eval_data, :90-107) follow in a later
// round only if the candidate is accepted AND the level goes on — like the reference's `eval`, which never builds them for a
// rejected candidate.
template <bool HUBER, bool ENERGY, bool FUSED>
__global__ __launch_bounds__(SPLIT_BLOCK) __attribute__((amdgpu_waves_per_eu(FUSED ? (ENERGY ? VORS_FSPLIT_ENERGY_WAVES : VORS_FSPLIT_WAVES) : (ENERGY ? VORS_SPLIT_ENERGY_WAVES : VORS_SPLIT_WAVES)))) void
lm_split_eval_kernel(Geom g, const uint8_t* __restrict__ cur0, const uint8_t* __restrict__ curu, const uint8_t* __restrict__ kf0,
                     const uint8_t* __restrict__ kfu, const uint16_t* __restrict__ kf_depth, Records rec, LmSplitWs ws, int round) {
    __shared__ LmShared s;
    const int n_active = ws.count[(ENERGY ? SPLIT_COUNT_STRIDE : 0) + round], n_items = n_active * ws.chunks0;
    const int* list = ws.list[round & 1];
    // item -> (chunk, pair) with the pair index fastest: the chunks a coarser level does not use are the tail of the grid, so the
    // working workgroups stay contiguous in blockIdx (spread over all XCDs)
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int chunk = item / n_active, a = item - chunk * n_active;
        const int pair = __builtin_amdgcn_readfirstlane(ENERGY ? list[ws.cap - 1 - a] : list[a]);
        const LmSplitState* st = ws.state + pair;
        const int lvl = __builtin_amdgcn_readfirstlane(st->lvl);
        const int chunks = split_chunks(ws, lvl);
        if (chunk >= chunks) continue;  // coarser levels are cut into fewer chunks
        const int phase = __builtin_amdgcn_readfirstlane(st->phase);
        const Iso model = iso_uniform(iso_load(phase == 1 ? st->cand : st->model));
        const ImgCtx c = level_ctx(g, cur0, curu, pair, lvl);
        float acc[NACC];
        float* out = ws.partials + ((size_t)pair * ws.chunks + chunk) * 32;
        with_level_source<true, true, FUSED>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n_units) {
            // chunk c = units [c * per + min(c, rem), ...): one 32-bit scalar division (the 64-bit n * c / chunks costs ~150 scalar
            // instructions apiece, in every workgroup)
            const unsigned per = (unsigned)n_units / (unsigned)chunks, rem = (unsigned)n_units - per * (unsigned)chunks;
            const int first = (int)((unsigned)chunk * per + min((unsigned)chunk, rem));
            const int last = (int)((unsigned)(chunk + 1) * per + min((unsigned)(chunk + 1), rem));
            eval_accumulate<SPLIT_BLOCK, HUBER, false, typename std::remove_cv<typename std::remove_reference<decltype(src)>::type>::type, ENERGY>(
                src, last, c, model, acc, nullptr, first, FUSED ? st : nullptr);
        });
        if (ENERGY) {
            float e = acc[0], cnt = acc[1];
            block_sum2<SPLIT_BLOCK>(e, cnt, s);
            if (threadIdx.x == 0) {
                out[0] = e;
                out[1] = cnt;
            }
        } else {
            block_reduce<SPLIT_BLOCK>(acc, s, 0);
            if (threadIdx.x < NACC) out[threadIdx.x] = s.sums[0][threadIdx.x];
        }
        __syncthreads();
    }
}
#if !VORS_FUSED
// (arithmetic-independent: compiled once, in the exact object)
// One wavefront per active pair: chunk partials -> sums, then LMOptimizerState::eval's verdict + stop_criterion + the next
// step() (lm_optimizer.rs:123-192), exactly as solve_level sequences them; a finished level hands over to the next one
// (statistics, inverse_compositional.rs:190-200). Pairs that continue are appended to the next round's list.
// STEP_WAVES pairs per workgroup, one wavefront each: their appends to the next round's list share ONE atomic per kind and workgroup —
// with a workgroup per pair the 4096 same-address atomics of a full round were the kernel (55 us; ~13 ns apiece).
// (The late rounds concern a handful of pairs and are latency: one pair per workgroup there.)
template <int STEP_WAVES>
__global__ __launch_bounds__(64 * STEP_WAVES) void lm_split_step_kernel(Geom g, LmSplitWs ws, vors_pair_stats* __restrict__ out_stats, int round, int late,
                                                                        int next_late) {
    __shared__ float red_all[STEP_WAVES][32];
    __shared__ int s_kind[STEP_WAVES], s_base[3];
    const int n_active = split_n_active(ws, round);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* red = red_all[wave];
    for (int a0 = blockIdx.x * STEP_WAVES; a0 < n_active; a0 += gridDim.x * STEP_WAVES) {
        const int a = a0 + wave;
        const bool live = a < n_active;
        int append_kind = -1;  // -1: the pair does not go on; 0: full evaluation next; 1: energy-only evaluation next
        const int pair = split_active(ws, round, live ? a : 0);
        LmSplitState* st = ws.state + pair;
        const int lvl = st->lvl;
        const int chunks = split_chunks(ws, lvl);
        const int phase = st->phase;
        const bool have_full = phase != 1 || late;  // an energy-only round wrote the first two sums only
        const int n_sums = have_full ? NACC : 2;
        if (live && lane < n_sums) {  // chunks in index order; the loads of a batch of 8 are independent, the additions stay sequential
            const float* pp = ws.partials + (size_t)pair * ws.chunks * 32 + lane;
            float t = 0.f;
            int ch = 0;
            for (; ch + 8 <= chunks; ch += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = pp[(ch + q) * 32];
#pragma unroll
                for (int q = 0; q < 8; ++q) t += v[q];
            }
            for (; ch < chunks; ++ch) t += pp[ch * 32];
            red[lane] = t;
        }
        __syncthreads();
        if (live && lane == 0) {
            const float energy = red[0] / red[1];  // energy_sum / residuals.len(): 0/0 = NaN like the reference
            Iso cur_model = iso_load(st->model);
            float lm_coef = st->lm_coef, cur_energy = st->cur_energy;
            int nb_iter = st->nb_iter;
            int n_full = st->n_full;  // the initial evaluation + every accepted candidate of this level so far (statistics)
            bool take = false, done = false, need_gh = false;
            if (phase == 0) {  // init: lm_optimizer.rs:113-118
                take = true;
                cur_energy = energy;
                lm_coef = 0.1f;
                nb_iter = 0;
                n_full = 1;
            } else if (phase == 4) {  // g, H of the candidate accepted one round ago (now the kept model): on to its step()
                take = true;
            } else {
                const bool too_many_iterations = nb_iter > 20;  // stop_criterion: lm_optimizer.rs:156-192
                if (energy > cur_energy) {                      // Err(energy)
                    if (too_many_iterations) done = true;
                    else lm_coef *= 10.0f;
                } else {
                    const float d_energy = cur_energy - energy;
                    n_full += 1;  // accepted: the reference forms its g and H (lm_optimizer.rs:147)
                    cur_energy = energy;
                    cur_model = iso_load(st->cand);
                    if (too_many_iterations) {
                        done = true;
                    } else {
                        lm_coef = 0.1f * lm_coef;
                        if (!(d_energy > 1.0f)) done = true;
                        else if (have_full) take = true;  // the level goes on from this candidate, its g and H are at hand
                        else need_gh = true;              // ... or are the business of the next round
                    }
                }
            }
            bool again = false;
            if (done) {  // level finished
                if (out_stats) {
                    out_stats[pair].nb_iter[lvl] = nb_iter;
                    out_stats[pair].nb_grad_evals[lvl] = n_full;
                    out_stats[pair].energy[lvl] = cur_energy;
                }
                iso_store(cur_model, st->model);
                if (lvl > 0) {
                    iso_store(cur_model, st->entry);
                    st->lvl = lvl - 1;
                    st->phase = 0;
                    nb_iter = 0;
                    n_full = 0;
                    again = true;
                } else {
                    st->phase = 2;
                }
            } else if (need_gh) {
                iso_store(cur_model, st->model);
                st->cur_energy = cur_energy;
                st->lm_coef = lm_coef;
                st->phase = 4;
                again = true;
            } else {  // step(): lm_optimizer.rs:123-136 on the kept state's sums
                if (take) {
                    for (int q = 0; q < NACC; ++q) st->sums[q] = red[q];
                    iso_store(cur_model, st->model);
                    st->cur_energy = cur_energy;
                }
                st->lm_coef = lm_coef;
                nb_iter += 1;
                const float* sm = take ? red : st->sums;
                float h[36], gr[6];
                for (int q = 0; q < 6; ++q) gr[q] = sm[2 + q];
                int k = 8;
                for (int q = 0; q < 6; ++q)
                    for (int r = q; r < 6; ++r) {
                        h[q * 6 + r] = sm[k];
                        h[r * 6 + q] = sm[k];
                        ++k;
                    }
                Iso cand;
                if ((g.arith == VORS_ARITH_FUSED && !VORS_ABL_EXACT_STEP) ? lm_step_fast(h, gr, cur_model, lm_coef, &cand) : lm_step(h, gr, cur_model, lm_coef, &cand)) {
                    iso_store(cand, st->cand);
                    st->phase = 1;
                    again = true;
                } else {  // Cholesky failed: the level's progress is discarded and tracking stops (inverse_compositional.rs:195-199)
                    for (int q = 0; q < 7; ++q) st->model[q] = st->entry[q];
                    st->went_well = 0;
                    st->phase = 2;
                    if (out_stats)
                        for (int l2 = lvl; l2 >= 0; --l2) {
                            out_stats[pair].nb_iter[l2] = 0;
                            out_stats[pair].nb_grad_evals[l2] = 0;
                            out_stats[pair].energy[l2] = 0.f;
                        }
                }
            }
            st->nb_iter = nb_iter;
            st->n_full = n_full;
            // FUSED arithmetic: the context of the evaluation this pair is due next, for all its workgroups (engine.h LmSplitState::fctx)
            if (again && g.arith == VORS_ARITH_FUSED) store_fused_ctx(g, st->lvl, iso_load(st->phase == 1 ? st->cand : st->model), st);
            // a candidate goes to the energy-only launch of the next round, unless that round is a late one (full evaluations only)
            if (again) append_kind = (round == ws.side_round && st->lvl > 0) ? 2 : ((st->phase == 1 && !next_late) ? 1 : 0);  // 2: side lane
        }
        if (lane == 0) s_kind[wave] = append_kind;
        __syncthreads();
        if (threadIdx.x == 0) {  // one reservation per kind for the whole workgroup
            int n_full = 0, n_energy = 0, n_side = 0;
            for (int w = 0; w < STEP_WAVES; ++w) {
                n_full += s_kind[w] == 0 ? 1 : 0;
                n_energy += s_kind[w] == 1 ? 1 : 0;
                n_side += s_kind[w] == 2 ? 1 : 0;
            }
            s_base[0] = n_full ? atomicAdd(&ws.count[round + 1], n_full) : 0;
            s_base[1] = n_energy ? atomicAdd(&ws.count[SPLIT_COUNT_STRIDE + round + 1], n_energy) : 0;
            s_base[2] = n_side ? atomicAdd(&ws.count[SPLIT_SIDE_COUNT], n_side) : 0;
        }
        __syncthreads();
        if (lane == 0 && append_kind >= 0) {
            int rank = 0;
            for (int w = 0; w < wave; ++w) rank += s_kind[w] == append_kind ? 1 : 0;
            if (append_kind == 2) ws.side_list[s_base[2] + rank] = pair;
            else if (append_kind == 1) ws.list[(round + 1) & 1][ws.cap - 1 - (s_base[1] + rank)] = pair;
            else ws.list[(round + 1) & 1][s_base[0] + rank] = pair;
        }
        __syncthreads();
    }
}

// The pairs the side lane has brought to level 0 enter round `round` as pairs due a full evaluation (a level's init).
__global__ __launch_bounds__(256) void lm_split_merge_kernel(LmSplitWs ws, int round) {
    const int n = ws.count[SPLIT_JOIN_COUNT];
    for (int i = threadIdx.x; i < n; i += 256) ws.list[round & 1][atomicAdd(&ws.count[round], 1)] = ws.join_list[i];
}
void launch_lm_split_merge(LmSplitWs ws, int round, hipStream_t s) { hipLaunchKernelGGL(lm_split_merge_kernel, dim3(1), dim3(256), 0, s, ws, round); }
void launch_lm_split_step(const Geom& g, LmSplitWs ws, vors_pair_stats* out_stats, int round, int late, int next_late, int grid, hipStream_t s) {
    if (grid >= 1024 && !late)
        hipLaunchKernelGGL(lm_split_step_kernel<8>, dim3((grid + 7) / 8), dim3(512), 0, s, g, ws, out_stats, round, late, next_late);
    else
        hipLaunchKernelGGL(lm_split_step_kernel<1>, dim3(grid), dim3(64), 0, s, g, ws, out_stats, round, late, next_late);
}
#else
// host-side launchers of the (arithmetic-independent) step and merge kernels, defined in the exact object
void launch_lm_split_merge(LmSplitWs ws, int round, hipStream_t s);
void launch_lm_split_step(const Geom& g, LmSplitWs ws, vors_pair_stats* out_stats, int round, int late, int next_late, int grid, hipStream_t s);
#endif

#define VORS_LM_KARGS g, cur.level0, cur.upper, kf.level0, kf.upper, kf_depth, rec, prev_poses7, kf_poses7, out_poses7, out_status, out_stats
template <int BLOCK, bool DENSE>
static void launch_lm_track_block(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7,
                                  const float* kf_poses7, float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats,
                                  int n_pairs, int mode, LmSplitWs split, hipStream_t s) {
    if (g.huber_delta > 0.f)
        hipLaunchKernelGGL((lm_track_kernel<BLOCK, true, DENSE, kFused>), dim3(n_pairs), dim3(BLOCK), 0, s, VORS_LM_KARGS, mode, split);
    else
        hipLaunchKernelGGL((lm_track_kernel<BLOCK, false, DENSE, kFused>), dim3(n_pairs), dim3(BLOCK), 0, s, VORS_LM_KARGS, mode, split);
}
static void launch_lm_track_mode(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7,
                                 const float* kf_poses7, float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs,
                                 int block, int mode, LmSplitWs split, hipStream_t s) {
#define VORS_LM_ARGS g, cur, kf, kf_depth, rec, prev_poses7, kf_poses7, out_poses7, out_status, out_stats, n_pairs, mode, split, s
    if (g.mode == VORS_CANDIDATES_DENSE) {
        if (block >= 1024) launch_lm_track_block<1024, true>(VORS_LM_ARGS);
        else if (block >= 512) launch_lm_track_block<512, true>(VORS_LM_ARGS);
        else if (block >= 256) launch_lm_track_block<256, true>(VORS_LM_ARGS);
        else if (block >= 128) launch_lm_track_block<128, true>(VORS_LM_ARGS);
        else launch_lm_track_block<64, true>(VORS_LM_ARGS);
    } else {
        if (block >= 1024) launch_lm_track_block<1024, false>(VORS_LM_ARGS);
        else if (block >= 512) launch_lm_track_block<512, false>(VORS_LM_ARGS);
        else if (block >= 256) launch_lm_track_block<256, false>(VORS_LM_ARGS);
        else if (block >= 128) launch_lm_track_block<128, false>(VORS_LM_ARGS);
        else launch_lm_track_block<64, false>(VORS_LM_ARGS);
    }
#undef VORS_LM_ARGS
}

// ------------------------------------------------------------------------------------------------------------
// One evaluation (eval_energy + compute_eval_data, lm_optimizer.rs:68-107) of ONE level of ONE pair of the handle at an explicit
// model, in this object's arithmetic: the operator-level view of the tracker's own point sources (inspection / parity tests:
// EXACT vs FUSED sums). out29 = sum r^2 (or Huber loss), n_inside, g[6], H upper triangle[21].
// ------------------------------------------------------------------------------------------------------------
template <bool HUBER, bool DENSE, bool FUSED>
__global__ __launch_bounds__(256) void lm_eval_level_kernel(Geom g, const uint8_t* __restrict__ cur0, const uint8_t* __restrict__ curu,
                                                             const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                             const uint16_t* __restrict__ kf_depth, Records rec, int pair, int lvl,
                                                             const float* __restrict__ model7, float* __restrict__ out29) {
    __shared__ LmShared s;
    const Iso model = iso_uniform(iso_load(model7));
    const ImgCtx c = level_ctx(g, cur0, curu, pair, lvl);
    float acc[NACC];
    with_level_source<DENSE, true, FUSED>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n_units) {
        eval_accumulate<256, HUBER, false>(src, n_units, c, model, acc, nullptr);
    });
    block_reduce<256>(acc, s, 0);
    if (threadIdx.x < NACC) out29[threadIdx.x] = s.sums[0][threadIdx.x];
}
#if VORS_FUSED
#define VORS_LAUNCH_LM_EVAL_LEVEL launch_lm_eval_level_fused
#else
#define VORS_LAUNCH_LM_EVAL_LEVEL launch_lm_eval_level_exact
#endif
void VORS_LAUNCH_LM_EVAL_LEVEL(const Geom& g_in, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, int pair, int lvl,
                               const float* model7, float* out29, hipStream_t s) {
    Geom g = g_in;
    g.wide_loads_ok = (((uintptr_t)kf.level0 | (uintptr_t)kf.upper | (uintptr_t)kf_depth | (uintptr_t)rec.IZ) % 16 == 0) ? 1 : 0;
#define VORS_EL_ARGS dim3(1), dim3(256), 0, s, g, cur.level0, cur.upper, kf.level0, kf.upper, kf_depth, rec, pair, lvl, model7, out29
    const bool dense = g.mode == VORS_CANDIDATES_DENSE, huber = g.huber_delta > 0.f;
    if (dense && huber) hipLaunchKernelGGL((lm_eval_level_kernel<true, true, kFused>), VORS_EL_ARGS);
    else if (dense) hipLaunchKernelGGL((lm_eval_level_kernel<false, true, kFused>), VORS_EL_ARGS);
    else if (huber) hipLaunchKernelGGL((lm_eval_level_kernel<true, false, kFused>), VORS_EL_ARGS);
    else hipLaunchKernelGGL((lm_eval_level_kernel<false, false, kFused>), VORS_EL_ARGS);
#undef VORS_EL_ARGS
}

#if VORS_FUSED
#define VORS_LAUNCH_LM_TRACK launch_lm_track_fused
#else
#define VORS_LAUNCH_LM_TRACK launch_lm_track_exact
#endif
void VORS_LAUNCH_LM_TRACK(const Geom& g_in, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7,
                          const float* kf_poses7, float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs, int block,
                          LmSplitWs split, hipStream_t s) {
    Geom g = g_in;
    g.wide_loads_ok = (((uintptr_t)kf.level0 | (uintptr_t)kf.upper | (uintptr_t)kf_depth | (uintptr_t)rec.IZ) % 16 == 0) ? 1 : 0;
#define VORS_LM_MARGS g, cur, kf, kf_depth, rec, prev_poses7, kf_poses7, out_poses7, out_status, out_stats, n_pairs, block
    if (g.mode != VORS_CANDIDATES_DENSE || split.chunks <= 0) {
        launch_lm_track_mode(VORS_LM_MARGS, 0, split, s);
        return;
    }
    // coarse levels per pair, then the evaluation rounds on the finest levels, then the per-pair epilogue (which also finishes
    // the rare pairs still iterating after the last round)
    split.n_split = std::max(1, std::min(split.n_split, g.L));
    split.rounds = std::max(1, std::min(split.rounds, VORS_SPLIT_MAX_ROUNDS));
    (void)hipMemsetAsync(split.count, 0, SPLIT_COUNT_INTS * sizeof(int), s);
    const int join_round = split.side_round >= 0 ? std::min(split.side_round + 3, split.rounds) : -1;
    launch_lm_track_mode(VORS_LM_MARGS, 1, split, s);
    const int base_chunks = std::max(1, split.chunks / 4);
    for (int r = 0; r < split.rounds; ++r) {
        // every pair needs at least two evaluations per level: full grids. Later rounds concern fewer and fewer pairs, finally a
        // handful of stragglers whose evaluations are pure latency: they are cut into 4x more chunks and get small grids
        // (grid-stride loops keep any count correct).
        static const int late_env = getenv("VORS_LM_LATE_FROM") ? atoi(getenv("VORS_LM_LATE_FROM")) : -1;  // (development knob)
        const int late_from = late_env >= 0 ? late_env : 2 * split.n_split + 2;
        const int late = r >= late_from, next_late = r + 1 >= late_from;
        if (r == join_round) {  // the side lane's pairs come back (it has had two long rounds to finish: no wait in practice)
            (void)hipStreamWaitEvent(s, split.ev_join, 0);
            launch_lm_split_merge(split, r, s);
        }
        static const int late_chunks_env = getenv("VORS_LM_LATE_CHUNKS") ? atoi(getenv("VORS_LM_LATE_CHUNKS")) : 0;  // (development knob)
        split.chunks0 = late ? (late_chunks_env > 0 ? std::min(late_chunks_env, split.chunks) : split.chunks) : base_chunks;
        const int full = n_pairs * split.chunks0;
        const int shrink = r < 2 * split.n_split ? 1 : (late ? 16 : 2);
        const int grid = std::max(std::min(full, 256), full / shrink);
        // Until the late rounds the pairs march roughly in step: full evaluations (a level's init, or the g / H of a candidate
        // that goes on) dominate the even rounds, candidates' energies the odd ones; the other kind gets a small grid.
        const int minor = std::max(std::min(full, 256), grid / 8);
        const int grid_full = late ? grid : ((r & 1) ? minor : grid), grid_energy = (r & 1) ? grid : minor;
#define VORS_SPLIT_KARGS g, cur.level0, cur.upper, kf.level0, kf.upper, kf_depth, rec, split, r
        if (g.huber_delta > 0.f) {
            hipLaunchKernelGGL((lm_split_eval_kernel<true, false, kFused>), dim3(grid_full), dim3(SPLIT_BLOCK), 0, s, VORS_SPLIT_KARGS);
            if (!late && r > 0) hipLaunchKernelGGL((lm_split_eval_kernel<true, true, kFused>), dim3(grid_energy), dim3(SPLIT_BLOCK), 0, s, VORS_SPLIT_KARGS);
        } else {
            hipLaunchKernelGGL((lm_split_eval_kernel<false, false, kFused>), dim3(grid_full), dim3(SPLIT_BLOCK), 0, s, VORS_SPLIT_KARGS);
            if (!late && r > 0) hipLaunchKernelGGL((lm_split_eval_kernel<false, true, kFused>), dim3(grid_energy), dim3(SPLIT_BLOCK), 0, s, VORS_SPLIT_KARGS);
        }
#undef VORS_SPLIT_KARGS
        launch_lm_split_step(g, split, out_stats, r, late, next_late, std::max(1, n_pairs / shrink), s);
        if (r == split.side_round) {  // fork: the pairs still above level 0 finish those levels on the side stream, one workgroup each
            (void)hipEventRecord(split.ev_fork, s);
            (void)hipStreamWaitEvent(split.side_stream, split.ev_fork, 0);
            launch_lm_track_mode(g, cur, kf, kf_depth, rec, prev_poses7, kf_poses7, out_poses7, out_status, out_stats, n_pairs, 1024, 4, split,
                                 split.side_stream);
            (void)hipEventRecord(split.ev_join, split.side_stream);
        }
    }
    if (join_round == split.rounds) {
        (void)hipStreamWaitEvent(s, split.ev_join, 0);
        launch_lm_split_merge(split, split.rounds, s);
    }
    // the pairs still iterating (a handful, each with a long serial tail) finish in parallel, one 1024-thread workgroup each (the grid grows with
    // the batch: a hard batch may leave more than 256 of them, and a workgroup beyond the active list returns at once)
    launch_lm_track_mode(g, cur, kf, kf_depth, rec, prev_poses7, kf_poses7, out_poses7, out_status, out_stats, std::min(n_pairs, std::max(256, n_pairs / 8)), 1024, 3,
                         split, s);
    launch_lm_track_mode(VORS_LM_MARGS, 2, split, s);
#undef VORS_LM_MARGS
}

#if !VORS_FUSED
void launch_lm_track(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7,
                     const float* kf_poses7, float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs, int block,
                     LmSplitWs split, hipStream_t s) {
    if (g.arith == VORS_ARITH_FUSED)
        launch_lm_track_fused(g, cur, kf, kf_depth, rec, prev_poses7, kf_poses7, out_poses7, out_status, out_stats, n_pairs, block, split, s);
    else
        launch_lm_track_exact(g, cur, kf, kf_depth, rec, prev_poses7, kf_poses7, out_poses7, out_status, out_stats, n_pairs, block, split, s);
}

// ------------------------------------------------------------------------------------------------------------
// Operator level (one pyramid level, explicit observations)
// ------------------------------------------------------------------------------------------------------------
#define OP_BLOCK 256
__global__ void records_from_obs_kernel(Intr k, int rows, int cols, const uint8_t* __restrict__ tmpl, int n,
                                        const int32_t* __restrict__ xy, const float* __restrict__ iz,
                                        const float* __restrict__ jac, Records rec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = xy[2 * i], y = xy[2 * i + 1];
    const V3 P = back_project(k, (float)x, (float)y, 1.0f / iz[i]);
    rec.A[i] = make_float4(P.x, P.y, P.z, (float)tmpl[(size_t)y * cols + x]);
    rec.B[i] = make_float4(jac[6 * i], jac[6 * i + 1], jac[6 * i + 2], jac[6 * i + 3]);
    rec.C[i] = make_float2(jac[6 * i + 4], jac[6 * i + 5]);
    rec.XY[i] = (uint32_t)x | ((uint32_t)y << 16);
    rec.IZ[i] = iz[i];
}
void launch_records_from_obs(Intr k, int rows, int cols, const uint8_t* tmpl, int n, const int32_t* xy, const float* iz,
                             const float* jac, Records rec, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(records_from_obs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, k, rows, cols, tmpl, n, xy, iz, jac, rec);
}

template <bool HUBER>
__global__ __launch_bounds__(OP_BLOCK) void lm_eval_obs_kernel(RecSrc src, int n, ImgCtx c, const float* __restrict__ model7,
                                                                float* __restrict__ out, float* __restrict__ residuals) {
    __shared__ LmShared s;
    float acc[NACC];
    const Iso model = iso_load(model7);
    if (residuals)
        eval_accumulate<OP_BLOCK, HUBER, true>(src, n, c, model, acc, residuals);
    else
        eval_accumulate<OP_BLOCK, HUBER, false>(src, n, c, model, acc, nullptr);
    block_reduce<OP_BLOCK>(acc, s, 0);
    if (threadIdx.x == 0) {
        const float* a = s.sums[0];
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
static ImgCtx make_ctx(Intr k, int rows, int cols, const uint8_t* image, float huber) {
    ImgCtx c;
    c.img = image;
    c.rows = rows;
    c.cols = cols;
    c.k = k;
    c.huber = huber;
    c.inv_fu_d = c.inv_fv_d = 0.0;  // operator level (exact arithmetic): unused
    c.inv_fu = c.inv_fv = c.s_fuv = 0.f;
    return c;
}
void launch_lm_eval_obs(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta,
                        const float* model7, float* out, float* residuals, hipStream_t s) {
    RecSrc src{rec.A, rec.B, rec.C};
    if (huber_delta > 0.f)
        hipLaunchKernelGGL(lm_eval_obs_kernel<true>, dim3(1), dim3(OP_BLOCK), 0, s, src, n, make_ctx(k, rows, cols, image, huber_delta),
                           model7, out, residuals);
    else
        hipLaunchKernelGGL(lm_eval_obs_kernel<false>, dim3(1), dim3(OP_BLOCK), 0, s, src, n, make_ctx(k, rows, cols, image, huber_delta),
                           model7, out, residuals);
}

template <bool HUBER>
__global__ __launch_bounds__(OP_BLOCK) void lm_solve_obs_kernel(RecSrc src, int n, ImgCtx c, const float* __restrict__ model7,
                                                                 float* __restrict__ out) {
    __shared__ LmShared s;
    Iso model = iso_load(model7);
    int nb_iter = 0;
    float energy = 0.f, lm_coef = 0.f;
    const bool ok = solve_level<OP_BLOCK, HUBER>(src, n, c, &model, &nb_iter, &energy, &lm_coef, s);
    if (threadIdx.x == 0) {
        iso_store(model, out);
        out[7] = (float)nb_iter;
        out[8] = energy;
        out[9] = lm_coef;
        out[10] = ok ? 0.f : 1.f;
    }
}
void launch_lm_solve_obs(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta,
                         const float* model7, float* out, hipStream_t s) {
    RecSrc src{rec.A, rec.B, rec.C};
    if (huber_delta > 0.f)
        hipLaunchKernelGGL(lm_solve_obs_kernel<true>, dim3(1), dim3(OP_BLOCK), 0, s, src, n, make_ctx(k, rows, cols, image, huber_delta),
                           model7, out);
    else
        hipLaunchKernelGGL(lm_solve_obs_kernel<false>, dim3(1), dim3(OP_BLOCK), 0, s, src, n, make_ctx(k, rows, cols, image, huber_delta),
                           model7, out);
}

#endif  // !VORS_FUSED

}  // namespace vors
