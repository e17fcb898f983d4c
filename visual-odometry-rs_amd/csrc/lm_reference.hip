// REFERENCE arithmetic (vors_config.arithmetic = VORS_ARITH_REFERENCE): the LM evaluation with the reference's OWN SUMMATION ORDER.
//
// The reference sums sequentially in f32, candidate after candidate in extract_z's column-major order (x outer, y inner:
// src/core/track/inverse_compositional.rs:260-279):
//     energy_sum += r * r                      src/core/track/lm_optimizer.rs:72-84
//     gradient   += jac * r ; hessian += hes   lm_optimizer.rs:94-100   (hes = jac * jac^T precomputed, inverse_compositional.rs:345-348)
// i.e. one rounding for the product, one for the addition, no FMA, outside points skipped. The accept / stop comparisons of the LM loop
// (lm_optimizer.rs:144,179) are decided at ties, so ANY other order of these additions forks the loop in ~60 % of the pairs and leaves
// a 0.05-0.4 % tail beyond 1e-4. The EXACT arithmetic already gives bit-identical per-point residuals and Jacobians;
// this file adds the order:
//
//   * candidate lists in column-major order: sort_colmajor_kernel re-orders every (pair, level) list of the sparse modes by the key
//     x * rows + y (an LDS bitmap of the level + prefix popcounts = the rank of each candidate: keys are unique); the dense mode
//     enumerates pixels column by column;
//   * a PRODUCER / CONSUMER workgroup per frame pair: three wavefronts evaluate the points of a chunk in the reference's per-point
//     arithmetic and leave (r, J[6]) per point, IN LIST ORDER, in LDS; meanwhile 28 lanes of the fourth wavefront — one per sum:
//     sum r^2, 6 g, 21 H — walk the previous chunk point by point: two LDS reads, one multiplication, one addition. An outside point
//     is stored as zeros: adding +0 is exact (a running sum that starts at +0 never becomes -0), so it equals skipping it.
//   * step() with the IEEE Cholesky of lie.h and sinf / cosf as glibc computes them (lie.h ref_sinf / ref_cosf), the optical-flow sum
//     of the keyframe test in list order as well (inverse_compositional.rs:213-221).
//
// Result: the device follows the oracle's LM path decision for decision — iteration counts equal at every level, poses bit-identical
// (tests/test_gpu_reference.py asserts equality, not a tolerance). Cost: the chain of dependent additions (~5 cycles per point and
// evaluation) — 2-3x the time of EXACT. It is the parity anchor, not the throughput mode.
//
// Compile with -ffp-contract=off (no FMA may be formed from the consumer's multiply + add).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "device_common.h"
#include "engine.h"

namespace vors {

#define REF_BLOCK 256
#define REF_PROD (REF_BLOCK - 64)  // producer threads: wavefronts 1..3
#define REF_G 2                    // points per producer thread and chunk
#define REF_CH (REF_PROD * REF_G)  // points per chunk
template <bool HUBER>
struct RefRow {
    // plain: [r, J0..J5, -]        Huber: [loss term, w r, J0..J5, w, 1, -, -]
    static constexpr int RS = HUBER ? 12 : 8;
};

struct RefShared {
    float sums[2][32];  // ping-pong totals as in lm_kernels.hip: 0 = sum r^2, 1 = n_inside, 2..7 = g, 8..28 = H upper triangle row-wise
    float cand[8];      // candidate model (7) + step-ok flag
    int cnt;
};

struct RefImg {  // the current image of a level + its intrinsics
    const uint8_t* img;
    int rows, cols;
    Intr k;
    float huber;
};

struct RefPt {
    V3 P;        // back-projected keyframe point (camera.rs:135-140)
    float tmpl;  // template grey level
    float J[6];  // warp Jacobian (inverse_compositional.rs:313-341)
    bool valid;
};

// ---- point providers: point i of the level, in the reference's order --------------------------------------------------------------
struct RefSlimSrc {  // sparse modes: the sorted 12-byte lists
    const SlimRec* S;
    Intr k;
    FastDiv fu, fv;  // the focal lengths as verified fast divisors (lie.h div_uniform: bit-identical to the IEEE quotient, or `ok` = 0)
    __device__ __forceinline__ void get(int i, RefPt& p) const {
        const SlimRec r = S[(unsigned)i];
        const float x = (float)(r.xy & 0xffffu), y = (float)(r.xy >> 16);
        p.P = back_project_rt(IntrFast{k, fu, fv}, x, y, 1.0f / r.iz);
        p.tmpl = (float)(r.tg & 0xffu);
        warp_jacobian_at_rt((float)slim_gx(r.tg), (float)slim_gy(r.tg), x, y, r.iz, IntrFast{k, fu, fv}, p.J);
        p.valid = true;
    }
    __device__ __forceinline__ void xy_iz(int i, float* x, float* y, float* iz, bool* valid) const {
        const SlimRec r = S[(unsigned)i];
        *x = (float)(r.xy & 0xffffu);
        *y = (float)(r.xy >> 16);
        *iz = r.iz;
        *valid = true;
    }
};
struct RefDenseSrc {  // dense mode: pixel i of the column-major enumeration, recomputed from the keyframe pyramid + depth / IZ plane
    const Geom* g;
    const uint8_t* kf0;
    const uint8_t* kfu;
    const uint16_t* depth;  // level 0, this pair
    const float* iz;        // levels >= 1, this pair and level (row-major), NaN = Unknown
    int pair, lvl, rows, cols;
    Intr k;
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
    __device__ __forceinline__ void get(int i, RefPt& p) const {
        int x, y;
        float izv;
        raw(i, &x, &y, &izv, &p.valid);
        int gx, gy, tm;
        grad_tmpl_at(*g, kf0, kfu, pair, lvl, x, y, &gx, &gy, &tm);
        p.P = back_project(k, (float)x, (float)y, 1.0f / izv);
        p.tmpl = (float)tm;
        warp_jacobian_at((float)gx, (float)gy, (float)x, (float)y, izv, k, p.J);
    }
    __device__ __forceinline__ void xy_iz(int i, float* x, float* y, float* izv, bool* valid) const {
        int xi, yi;
        raw(i, &xi, &yi, izv, valid);
        *x = (float)xi;
        *y = (float)yi;
    }
};
struct RefObsSrc {  // operator level: explicit observations in the caller's order (Obs, lm_optimizer.rs:43-58)
    const float4* A;
    const float4* B;
    const float2* C;
    __device__ __forceinline__ void get(int i, RefPt& p) const {
        const float4 a = A[(unsigned)i], b = B[(unsigned)i];
        const float2 c = C[(unsigned)i];
        p.P = V3{a.x, a.y, a.z};
        p.tmpl = a.w;
        p.J[0] = b.x; p.J[1] = b.y; p.J[2] = b.z; p.J[3] = b.w; p.J[4] = c.x; p.J[5] = c.y;
        p.valid = a.w >= 0.f;
    }
};

// warp (lm_optimizer.rs:213-219) + interpolate (lm_optimizer.rs:227-251) + residual of one point -> its row. Returns inside.
template <bool HUBER>
__device__ __forceinline__ bool ref_point_row(const RefPt& p, const RefImg& c, const Iso& model, float* row, float* res_out) {
    float u, v;
    project_uv(c.k, iso_transform_point(model, p.P), &u, &v);
    const float uf = floorf(u), vf = floorf(v);
    const bool inside = p.valid && (uf >= 0.f) && (uf < (float)(c.cols - 2)) && (vf >= 0.f) && (vf < (float)(c.rows - 2));
    const unsigned off = inside ? (unsigned)((int)vf * c.cols + (int)uf) : 0u;
    const float vu_00 = (float)c.img[off], vu_01 = (float)c.img[off + 1u];
    const float vu_10 = (float)c.img[off + (unsigned)c.cols], vu_11 = (float)c.img[off + (unsigned)c.cols + 1u];
    const float fa = u - uf, fb = v - vf;
    const float im = (1.0f - fb) * (1.0f - fa) * vu_00 + fb * (1.0f - fa) * vu_10 + (1.0f - fb) * fa * vu_01 + fb * fa * vu_11;
    const float r = im - p.tmpl;
    if (res_out) *res_out = inside ? r : __builtin_nanf("");
    if (HUBER) {  // extension (oracle: lm_optimizer restatement, eval_energy / compute_eval_data with huber_delta)
        const float ar = fabsf(r);
        const bool quad = ar <= c.huber;
        const float e = quad ? r * r : c.huber * (2.0f * ar - c.huber);
        const float w = quad ? 1.0f : c.huber / ar;
        float4* q = reinterpret_cast<float4*>(row);
        q[0] = inside ? make_float4(e, w * r, p.J[0], p.J[1]) : make_float4(0.f, 0.f, 0.f, 0.f);
        q[1] = inside ? make_float4(p.J[2], p.J[3], p.J[4], p.J[5]) : make_float4(0.f, 0.f, 0.f, 0.f);
        q[2] = make_float4(inside ? w : 1.0f, 1.0f, 0.f, 0.f);
    } else {
        float4* q = reinterpret_cast<float4*>(row);
        q[0] = inside ? make_float4(r, p.J[0], p.J[1], p.J[2]) : make_float4(0.f, 0.f, 0.f, 0.f);
        q[1] = inside ? make_float4(p.J[3], p.J[4], p.J[5], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return inside;
}

// Which two (three) row entries the consumer lane multiplies: lane 0 = energy, 1..6 = g, 7..27 = H upper triangle row-wise.
template <bool HUBER>
__device__ __forceinline__ void ref_lane_slots(int lane, int* a, int* b, int* w) {
    int q = 0, s2 = 0;
    if (lane >= 7) {  // h = lane - 7 -> (q, s2), q <= s2, row-wise
        int h = lane - 7;
        q = 0;
        int len = 6;
        while (h >= len && q < 5) {
            h -= len;
            len -= 1;
            q += 1;
        }
        s2 = q + h;
    }
    if (HUBER) {
        if (lane == 0) { *a = 0; *b = 9; *w = 9; }            // loss term * 1 * 1
        else if (lane < 7) { *a = 1 + lane; *b = 1; *w = 9; }  // jac[q] * (w r) * 1
        else { *a = 2 + q; *b = 2 + s2; *w = 8; }              // w * (jac[q] * jac[s])
        if (lane >= 28) { *a = 9; *b = 9; *w = 9; }
    } else {
        if (lane == 0) { *a = 0; *b = 0; }               // r * r
        else if (lane < 7) { *a = lane; *b = 0; }        // jac[q] * r
        else { *a = 1 + q; *b = 1 + s2; }                // jac[q] * jac[s]
        if (lane >= 28) { *a = 7; *b = 7; }
        *w = 0;
    }
}

// One evaluation: eval_energy + compute_eval_data (lm_optimizer.rs:68-107) of the n points of `src` at `model`, sums in list order
// -> s.sums[dst][0..28]. All REF_BLOCK threads call it; ends with a barrier.
template <bool HUBER, class Src>
__device__ void ref_eval(const Src& src, int n, const RefImg& c, const Iso& model, float* buf, RefShared& s, int dst, float* residuals = nullptr) {
    constexpr int RS = RefRow<HUBER>::RS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nchunks = (n + REF_CH - 1) / REF_CH;
    int sa, sb, sw;
    ref_lane_slots<HUBER>(lane, &sa, &sb, &sw);
    float acc = 0.f;
    int cnt = 0;
    if (threadIdx.x == 0) s.cnt = 0;
    for (int ch = 0; ch <= nchunks; ++ch) {
        if (wave != 0) {
            if (ch < nchunks) {  // produce chunk ch
                float* rows = buf + (ch & 1) * (REF_CH * RS);
                const int p = (int)threadIdx.x - 64;
#pragma unroll
                for (int g2 = 0; g2 < REF_G; ++g2) {
                    const int li = p + g2 * REF_PROD, i = ch * REF_CH + li;
                    if (i < n) {
                        RefPt pt;
                        src.get(i, pt);
                        float res;
                        const bool in = ref_point_row<HUBER>(pt, c, model, rows + li * RS, residuals ? &res : nullptr);
                        if (residuals) residuals[i] = res;
                        cnt += in ? 1 : 0;
                    }
                }
            }
        } else if (ch > 0 && lane < 32) {  // consume chunk ch - 1: the sequential sums, one lane per sum (the upper half-wavefront sits out: its LDS requests would cost the same again)
            const float* rows = buf + ((ch - 1) & 1) * (REF_CH * RS);
            const int m = min(REF_CH, n - (ch - 1) * REF_CH);
            int i = 0;
            for (; i + 8 <= m; i += 8) {
                float va[8], vb[8], vw[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    va[j] = rows[(i + j) * RS + sa];
                    vb[j] = rows[(i + j) * RS + sb];
                    if (HUBER) vw[j] = rows[(i + j) * RS + sw];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float pr = va[j] * vb[j];
                    if (HUBER) pr = vw[j] * pr;
                    acc = acc + pr;
                }
            }
            for (; i < m; ++i) {
                float pr = rows[i * RS + sa] * rows[i * RS + sb];
                if (HUBER) pr = rows[i * RS + sw] * pr;
                acc = acc + pr;
            }
        }
        __syncthreads();
    }
    if (wave != 0 && cnt != 0) atomicAdd(&s.cnt, cnt);
    if (wave == 0 && lane < 28) s.sums[dst][lane == 0 ? 0 : lane + 1] = acc;
    __syncthreads();
    if (threadIdx.x == 0) s.sums[dst][1] = (float)s.cnt;
    __syncthreads();
}

__device__ __forceinline__ float ref_uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ Iso ref_iso_uniform(const Iso& m) {
    return Iso{V3{ref_uniform_f(m.t.x), ref_uniform_f(m.t.y), ref_uniform_f(m.t.z)},
               Quat{ref_uniform_f(m.q.i), ref_uniform_f(m.q.j), ref_uniform_f(m.q.k), ref_uniform_f(m.q.w)}};
}

// step() (lm_optimizer.rs:123-136) by one lane on the kept state's sums.
__device__ __forceinline__ void ref_step_lane0(RefShared& s, int cur, const Iso& model, float lm_coef) {
    if (threadIdx.x == 0) {
        const float* a = s.sums[cur];
        float h[36], g[6];
        for (int q = 0; q < 6; ++q) g[q] = a[2 + q];
        int k = 8;
        for (int q = 0; q < 6; ++q)
            for (int r = q; r < 6; ++r) {
                h[q * 6 + r] = a[k];
                h[r * 6 + q] = a[k];
                ++k;
            }
        Iso cand;
        const bool ok = lm_step(h, g, model, lm_coef, &cand);
        iso_store(cand, s.cand);
        s.cand[7] = ok ? 1.0f : 0.0f;
    }
    __syncthreads();
}

// optimizer::State::iterative_solve (optimizer.rs:57-70) with init / step / eval / stop_criterion of lm_optimizer.rs:113-192.
template <bool HUBER, class Src>
__device__ bool ref_solve_level(const Src& src, int n, const RefImg& c, Iso* model, int* nb_iter_out, float* energy_out, float* lm_coef_out,
                                int* n_full_out, float* buf, RefShared& s) {
    Iso cur_model = *model;
    int cur = 0;
    ref_eval<HUBER>(src, n, c, cur_model, buf, s, cur);  // init: lm_optimizer.rs:113-118
    float cur_energy = ref_uniform_f(s.sums[cur][0] / s.sums[cur][1]);  // energy_sum / residuals.len(): 0 / 0 = NaN like the reference
    float lm_coef = 0.1f;
    int nb_iter = 0, n_full = 1;
    for (;;) {
        nb_iter += 1;
        ref_step_lane0(s, cur, cur_model, lm_coef);
        if (ref_uniform_f(s.cand[7]) == 0.0f) return false;
        const Iso cand = ref_iso_uniform(iso_load(s.cand));
        const bool too_many_iterations = nb_iter > 20;  // stop_criterion: lm_optimizer.rs:156-192
        ref_eval<HUBER>(src, n, c, cand, buf, s, 1 - cur);  // eval(): lm_optimizer.rs:140-149
        const float energy = ref_uniform_f(s.sums[1 - cur][0] / s.sums[1 - cur][1]);
        if (energy > cur_energy) {  // Err(energy)
            if (too_many_iterations) break;
            lm_coef *= 10.0f;
            continue;
        }
        const float d_energy = cur_energy - energy;
        n_full += 1;
        cur = 1 - cur;
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
    *n_full_out = n_full;
    return true;
}

__device__ __forceinline__ RefImg ref_level_img(const Geom& g, const uint8_t* cur0, const uint8_t* curu, int pair, int lvl) {
    RefImg c;
    c.img = level_ptr(g, cur0, curu, pair, lvl);
    c.rows = g.lv[lvl].rows;
    c.cols = g.lv[lvl].cols;
    c.k = g.lv[lvl].k;
    c.huber = g.huber_delta;
    return c;
}
template <bool DENSE, class F>
__device__ __forceinline__ void ref_with_source(const Geom& g, int lvl, int pair, const uint8_t* kf0, const uint8_t* kfu, const uint16_t* kf_depth,
                                                const Records& rec, F&& f) {
    const LevelGeom lg = g.lv[lvl];
    if constexpr (DENSE) {
        RefDenseSrc src{&g, kf0, kfu, kf_depth + (size_t)pair * g.S0, lvl > 0 ? rec.IZ + (size_t)pair * g.slots_total + lg.slot_off : nullptr,
                        pair, lvl, lg.rows, lg.cols, lg.k};
        f(src, lg.rows * lg.cols);
    } else {
        RefSlimSrc src{rec.S + (size_t)pair * g.slots_total + lg.slot_off, lg.k, lg.fu, lg.fv};
        f(src, __builtin_amdgcn_readfirstlane(rec.n_used[(size_t)pair * VORS_MAX_LEVELS + lvl]));
    }
}

// Tracker::track for a batch (inverse_compositional.rs:177-224): one workgroup per frame pair, all levels.
template <bool HUBER, bool DENSE>
__global__ __launch_bounds__(REF_BLOCK) void lm_ref_track_kernel(Geom g, const uint8_t* __restrict__ cur0, const uint8_t* __restrict__ curu,
                                                                 const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                                 const uint16_t* __restrict__ kf_depth, Records rec,
                                                                 const float* __restrict__ prev_poses7, const float* __restrict__ kf_poses7,
                                                                 float* __restrict__ out_poses7, int32_t* __restrict__ out_status,
                                                                 vors_pair_stats* __restrict__ out_stats) {
    constexpr int RS = RefRow<HUBER>::RS;
    __shared__ __attribute__((aligned(16))) float buf[2 * REF_CH * RS];
    __shared__ RefShared s;
    const int pair = blockIdx.x;
    const Iso prev_pose = prev_poses7 ? iso_load(prev_poses7 + 7 * pair) : iso_identity();
    const Iso kf_pose = kf_poses7 ? iso_load(kf_poses7 + 7 * pair) : iso_identity();
    Iso lm_model = ref_iso_uniform(iso_mul(iso_inverse(prev_pose), kf_pose));  // inverse_compositional.rs:177
    bool went_well = true;
    for (int lvl = g.L - 1; lvl >= 0; --lvl) {
        const RefImg c = ref_level_img(g, cur0, curu, pair, lvl);
        int nb_iter = 0, n_full = 0;
        float energy = 0.f, lm_coef = 0.f;
        bool ok = false;
        ref_with_source<DENSE>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n) {
            ok = ref_solve_level<HUBER>(src, n, c, &lm_model, &nb_iter, &energy, &lm_coef, &n_full, buf, s);
        });
        if (out_stats && threadIdx.x == 0) {
            out_stats[pair].nb_iter[lvl] = ok ? nb_iter : 0;
            out_stats[pair].nb_grad_evals[lvl] = ok ? n_full : 0;
            out_stats[pair].energy[lvl] = ok ? energy : 0.f;
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
    // keyframe test on the coarsest level (inverse_compositional.rs:211-224): mean L1 displacement, summed in list order
    float flow_sum = 0.f;
    int flow_n = 0;
    {
        const int lvl = g.L - 1;
        const Intr k = g.lv[lvl].k;
        __syncthreads();
        ref_with_source<DENSE>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n) {
            constexpr int CAP = 2 * REF_CH * RS;
            for (int base = 0; base < n; base += CAP) {
                const int m = min(CAP, n - base);
                for (int j = threadIdx.x; j < m; j += REF_BLOCK) {
                    float x, y, iz;
                    bool valid;
                    src.xy_iz(base + j, &x, &y, &iz, &valid);
                    float u, v;
                    project_uv(k, iso_transform_point(lm_model, back_project(k, x, y, 1.0f / iz)), &u, &v);  // warp: lm_optimizer.rs:213-219
                    buf[j] = valid ? fabsf(x - u) + fabsf(y - v) : -1.0f;  // (a displacement is never negative: -1 = not a candidate)
                }
                __syncthreads();
                if (threadIdx.x == 0)
                    for (int j = 0; j < m; ++j) {
                        const float f = buf[j];
                        if (!(f < 0.f)) {  // (NaN counts as a candidate, like the reference)
                            flow_sum = flow_sum + f;
                            flow_n += 1;
                        }
                    }
                __syncthreads();
            }
        });
    }
    if (threadIdx.x == 0) {
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
            if constexpr (DENSE) {
                __syncthreads();
                if (threadIdx.x == 0) s.cnt = 0;
                __syncthreads();
                int mine = 0;
                ref_with_source<true>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n) {
                    for (int i = threadIdx.x; i < n; i += REF_BLOCK) {
                        int x, y;
                        float iz;
                        bool valid;
                        src.raw(i, &x, &y, &iz, &valid);
                        mine += valid ? 1 : 0;
                    }
                });
                if (mine) atomicAdd(&s.cnt, mine);
                __syncthreads();
                if (threadIdx.x == 0) out_stats[pair].n_points[lvl] = s.cnt;
            } else {
                if (threadIdx.x == 0) out_stats[pair].n_points[lvl] = rec.n_used[(size_t)pair * VORS_MAX_LEVELS + lvl];
            }
        }
        if (threadIdx.x == 0)
            for (int lvl = g.L; lvl < VORS_MAX_LEVELS; ++lvl) {
                out_stats[pair].nb_iter[lvl] = 0;
                out_stats[pair].nb_grad_evals[lvl] = 0;
                out_stats[pair].n_points[lvl] = 0;
                out_stats[pair].energy[lvl] = 0.f;
            }
    }
}

void launch_lm_track_reference(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7,
                               const float* kf_poses7, float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs,
                               hipStream_t s) {
#define VORS_REF_ARGS dim3(n_pairs), dim3(REF_BLOCK), 0, s, g, cur.level0, cur.upper, kf.level0, kf.upper, kf_depth, rec, prev_poses7, kf_poses7, \
                      out_poses7, out_status, out_stats
    const bool dense = g.mode == VORS_CANDIDATES_DENSE, huber = g.huber_delta > 0.f;
    if (dense && huber) hipLaunchKernelGGL((lm_ref_track_kernel<true, true>), VORS_REF_ARGS);
    else if (dense) hipLaunchKernelGGL((lm_ref_track_kernel<false, true>), VORS_REF_ARGS);
    else if (huber) hipLaunchKernelGGL((lm_ref_track_kernel<true, false>), VORS_REF_ARGS);
    else hipLaunchKernelGGL((lm_ref_track_kernel<false, false>), VORS_REF_ARGS);
#undef VORS_REF_ARGS
}

// One evaluation of one level of one pair of a prepared batch (vors_batch_eval_level in the REFERENCE arithmetic) -> 29 sums.
template <bool HUBER, bool DENSE>
__global__ __launch_bounds__(REF_BLOCK) void lm_ref_eval_level_kernel(Geom g, const uint8_t* __restrict__ cur0, const uint8_t* __restrict__ curu,
                                                                      const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                                      const uint16_t* __restrict__ kf_depth, Records rec, int pair, int lvl,
                                                                      const float* __restrict__ model7, float* __restrict__ out29) {
    constexpr int RS = RefRow<HUBER>::RS;
    __shared__ __attribute__((aligned(16))) float buf[2 * REF_CH * RS];
    __shared__ RefShared s;
    const Iso model = ref_iso_uniform(iso_load(model7));
    const RefImg c = ref_level_img(g, cur0, curu, pair, lvl);
    ref_with_source<DENSE>(g, lvl, pair, kf0, kfu, kf_depth, rec, [&](const auto& src, int n) { ref_eval<HUBER>(src, n, c, model, buf, s, 0); });
    if (threadIdx.x < 29) out29[threadIdx.x] = s.sums[0][threadIdx.x];
}
void launch_lm_eval_level_reference(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, int pair, int lvl,
                                    const float* model7, float* out29, hipStream_t s) {
#define VORS_REF_ARGS dim3(1), dim3(REF_BLOCK), 0, s, g, cur.level0, cur.upper, kf.level0, kf.upper, kf_depth, rec, pair, lvl, model7, out29
    const bool dense = g.mode == VORS_CANDIDATES_DENSE, huber = g.huber_delta > 0.f;
    if (dense && huber) hipLaunchKernelGGL((lm_ref_eval_level_kernel<true, true>), VORS_REF_ARGS);
    else if (dense) hipLaunchKernelGGL((lm_ref_eval_level_kernel<false, true>), VORS_REF_ARGS);
    else if (huber) hipLaunchKernelGGL((lm_ref_eval_level_kernel<true, false>), VORS_REF_ARGS);
    else hipLaunchKernelGGL((lm_ref_eval_level_kernel<false, false>), VORS_REF_ARGS);
#undef VORS_REF_ARGS
}

// ---- operator level on explicit observations, sums in the order of the observations (the reference's eval on that Obs) --------------
template <bool HUBER>
__global__ __launch_bounds__(REF_BLOCK) void lm_ref_eval_obs_kernel(RefObsSrc src, int n, RefImg c, const float* __restrict__ model7,
                                                                    float* __restrict__ out, float* __restrict__ residuals) {
    constexpr int RS = RefRow<HUBER>::RS;
    __shared__ __attribute__((aligned(16))) float buf[2 * REF_CH * RS];
    __shared__ RefShared s;
    const Iso model = ref_iso_uniform(iso_load(model7));
    ref_eval<HUBER>(src, n, c, model, buf, s, 0, residuals);
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
template <bool HUBER>
__global__ __launch_bounds__(REF_BLOCK) void lm_ref_solve_obs_kernel(RefObsSrc src, int n, RefImg c, const float* __restrict__ model7,
                                                                     float* __restrict__ out) {
    constexpr int RS = RefRow<HUBER>::RS;
    __shared__ __attribute__((aligned(16))) float buf[2 * REF_CH * RS];
    __shared__ RefShared s;
    Iso model = ref_iso_uniform(iso_load(model7));
    int nb_iter = 0, n_full = 0;
    float energy = 0.f, lm_coef = 0.f;
    const bool ok = ref_solve_level<HUBER>(src, n, c, &model, &nb_iter, &energy, &lm_coef, &n_full, buf, s);
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
    if (huber_delta > 0.f) hipLaunchKernelGGL(lm_ref_eval_obs_kernel<true>, dim3(1), dim3(REF_BLOCK), 0, s, src, n, c, model7, out, residuals);
    else hipLaunchKernelGGL(lm_ref_eval_obs_kernel<false>, dim3(1), dim3(REF_BLOCK), 0, s, src, n, c, model7, out, residuals);
}
void launch_lm_solve_obs_reference(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta, const float* model7,
                                   float* out, hipStream_t s) {
    const RefObsSrc src{rec.A, rec.B, rec.C};
    const RefImg c{image, rows, cols, k, huber_delta};
    if (huber_delta > 0.f) hipLaunchKernelGGL(lm_ref_solve_obs_kernel<true>, dim3(1), dim3(REF_BLOCK), 0, s, src, n, c, model7, out);
    else hipLaunchKernelGGL(lm_ref_solve_obs_kernel<false>, dim3(1), dim3(REF_BLOCK), 0, s, src, n, c, model7, out);
}

// ------------------------------------------------------------------------------------------------------------
// Candidate lists into extract_z's order (inverse_compositional.rs:260-279): column-major, key = x * rows + y. Keys are unique (one
// candidate per pixel), so the rank of a candidate is the number of set bits below its key in a bitmap of the level: segments of 2^18
// keys in LDS, per-thread word totals + a block scan. One workgroup per (level, pair); out of place into `tmp`, then copied back.
// ------------------------------------------------------------------------------------------------------------
#define SORT_WORDS 4096  // 131,072 keys per segment: 16 KB of bits + 16 KB of word prefixes
#define SORT_BLOCK 512
#define SORT_U 8          // independent loads in flight per thread (a pass is a chain of global round trips otherwise)
__global__ __launch_bounds__(SORT_BLOCK) __attribute__((amdgpu_waves_per_eu(6))) void sort_colmajor_kernel(Geom g, Records rec, int reg_cap) {
    __shared__ uint32_t bits[SORT_WORDS];
    __shared__ int wpre[SORT_WORDS];  // set bits in the words before this one (within the segment)
    __shared__ int wsum[SORT_BLOCK / 64];
    __shared__ int s_base;
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int l = blockIdx.x;
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
                wpre[w] = run;
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
void launch_sort_colmajor(const Geom& g, Records rec, int n_pairs, hipStream_t s) {
    if (g.mode == VORS_CANDIDATES_DENSE || !rec.sort_tmp) return;
    // VORS_REF_SORT_REGCAP (read per launch; tests): lists longer than this take the multi-pass form through the scratch copy — 0 forces it
    // for every list; the default is what 8 records per thread hold
    const char* e = getenv("VORS_REF_SORT_REGCAP");
    int reg_cap = SORT_BLOCK * SORT_U;
    if (e) reg_cap = std::max(0, std::min(reg_cap, atoi(e)));
    hipLaunchKernelGGL(sort_colmajor_kernel, dim3(g.L, n_pairs), dim3(SORT_BLOCK), 0, s, g, rec, reg_cap);
}

}  // namespace vors
