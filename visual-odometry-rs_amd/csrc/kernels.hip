// HIP kernels (gfx950 / CDNA4, wave64) of the direct-alignment hot path.
//
//   halve_mean_kernel        multires::mean_pyramid / halve            (reference src/core/multires.rs:21-31,67-88)
//   keyframe_sparse_kernel   gradient::centered / bloc_x / bloc_y / squared_norm (src/core/gradient.rs:15-44,74-93),
//                            candidates::coarse_to_fine::select (src/core/candidates/coarse_to_fine.rs:15-89),
//                            inverse-depth pyramid (src/core/inverse_depth.rs:24-29,49-98), extract_z / warp_jacobians
//                            (src/core/track/inverse_compositional.rs:105-161,260-341) — one wavefront per quad-tree root
//   keyframe_dense_kernel    same with an all-true level-0 mask (extension)
//   lm_track_kernel          Tracker::track's coarse->fine loop (inverse_compositional.rs:177-224) with the whole
//                            optimizer::State loop (src/math/optimizer.rs:57-70, src/core/track/lm_optimizer.rs:68-193)
//                            on-device: one workgroup per frame pair, 29-scalar wavefront-shuffle + LDS reduction.
//
// Compile with -ffp-contract=off: per-point arithmetic (warp, bilinear interpolation, residual, Jacobian) follows the
// reference's evaluation order exactly; only the ORDER OF SUMMATION of the normal equations differs (tree vs sequential).
// Where fusing is harmless (accumulating the sums) explicit fmaf is used.
#include <hip/hip_runtime.h>

#include "engine.h"
#include "synth_scene.h"

namespace vors {

#define VORS_INVALID_XY 0xFFFFFFFFu

// ------------------------------------------------------------------------------------------------------------
// image helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ const uint8_t* level_ptr(const Geom& g, const uint8_t* level0, const uint8_t* upper, int pair, int l) {
    return l == 0 ? level0 + (size_t)pair * g.S0 : upper + (size_t)pair * g.upper_stride + g.lv[l].img_off;
}

// Gradient at level l, pixel (x, y), of the pyramid (level0, upper):
//  l == 0: centred difference, truncating /2, 1-px border = 0                      gradient.rs:15-33
//  l >= 1: 2x2 block gradient of the next finer level (a c / b d), truncating /2   gradient.rs:74-93, multires.rs:112-126
__device__ __forceinline__ void grad_at(const Geom& g, const uint8_t* level0, const uint8_t* upper, int pair, int l, int x,
                                        int y, int* gx, int* gy) {
    if (l == 0) {
        const int rows = g.lv[0].rows, cols = g.lv[0].cols;
        if (x == 0 || y == 0 || x == cols - 1 || y == rows - 1) {
            *gx = 0;
            *gy = 0;
            return;
        }
        const uint8_t* p = level0 + (size_t)pair * g.S0 + (size_t)y * cols + x;
        *gx = ((int)p[1] - (int)p[-1]) / 2;
        *gy = ((int)p[cols] - (int)p[-cols]) / 2;
    } else {
        const int fc = g.lv[l - 1].cols;
        const uint8_t* p = level_ptr(g, level0, upper, pair, l - 1) + (size_t)(2 * y) * fc + 2 * x;
        const int a = p[0], c = p[1], b = p[fc], d = p[fc + 1];
        *gx = (c + d - a - b) / 2;
        *gy = (b - a + d - c) / 2;
    }
}

// ------------------------------------------------------------------------------------------------------------
// layout conversion (column-major DMatrix::as_slice() -> row-major device layout)
// ------------------------------------------------------------------------------------------------------------
template <class T>
__global__ void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int rows, int cols) {
    __shared__ T tile[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols;
    // src is cols x rows row-major (== rows x cols column-major)
    int sx = blockIdx.x * 32 + threadIdx.x;  // along rows (fast in src)
    int sy = blockIdx.y * 32 + threadIdx.y;  // along cols
    for (int j = 0; j < 32; j += 8)
        if (sx < rows && sy + j < cols) tile[threadIdx.y + j][threadIdx.x] = src[base + (size_t)(sy + j) * rows + sx];
    __syncthreads();
    int dx = blockIdx.y * 32 + threadIdx.x;  // col
    int dy = blockIdx.x * 32 + threadIdx.y;  // row
    for (int j = 0; j < 32; j += 8)
        if (dx < cols && dy + j < rows) dst[base + (size_t)(dy + j) * cols + dx] = tile[threadIdx.x][threadIdx.y + j];
}
void launch_transpose_u8(const uint8_t* src, uint8_t* dst, int rows, int cols, int n, hipStream_t s) {
    dim3 grid((rows + 31) / 32, (cols + 31) / 32, n), block(32, 8);
    hipLaunchKernelGGL(transpose_kernel<uint8_t>, grid, block, 0, s, src, dst, rows, cols);
}
void launch_transpose_u16(const uint16_t* src, uint16_t* dst, int rows, int cols, int n, hipStream_t s) {
    dim3 grid((rows + 31) / 32, (cols + 31) / 32, n), block(32, 8);
    hipLaunchKernelGGL(transpose_kernel<uint16_t>, grid, block, 0, s, src, dst, rows, cols);
}

// ------------------------------------------------------------------------------------------------------------
// mean pyramid: one level per launch, 4 output pixels per thread.  ((a+b+c+d) as u16 / 4) as u8
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void halve_mean_kernel(const uint8_t* __restrict__ src, size_t src_stride, int src_cols,
                                                          uint8_t* __restrict__ dst, size_t dst_stride, int dst_rows,
                                                          int dst_cols, int vec_ok) {
    const int groups = (dst_cols + 3) >> 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= groups * dst_rows) return;
    const int y = t / groups, x4 = (t - y * groups) * 4;
    const uint8_t* s0 = src + (size_t)blockIdx.y * src_stride + (size_t)(2 * y) * src_cols + 2 * x4;
    const uint8_t* s1 = s0 + src_cols;
    uint8_t* d = dst + (size_t)blockIdx.y * dst_stride + (size_t)y * dst_cols + x4;
    if (vec_ok) {
        const uint2 r0 = *reinterpret_cast<const uint2*>(s0);
        const uint2 r1 = *reinterpret_cast<const uint2*>(s1);
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t w0 = k < 2 ? r0.x : r0.y, w1 = k < 2 ? r1.x : r1.y;
            const int sh = (k & 1) * 16;
            const uint32_t a = (w0 >> sh) & 0xff, c = (w0 >> (sh + 8)) & 0xff;
            const uint32_t b = (w1 >> sh) & 0xff, dd = (w1 >> (sh + 8)) & 0xff;
            out |= (((a + b + c + dd) >> 2) & 0xff) << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(d) = out;
    } else {
        for (int k = 0; k < 4 && x4 + k < dst_cols; ++k) {
            const uint32_t a = s0[2 * k], c = s0[2 * k + 1], b = s1[2 * k], dd = s1[2 * k + 1];
            d[k] = (uint8_t)((a + b + c + dd) >> 2);
        }
    }
}

void launch_pyramid(const Geom& g, Pyramid pyr, int n_pairs, hipStream_t s) {
    for (int l = 1; l < g.L; ++l) {
        const LevelGeom& src = g.lv[l - 1];
        const LevelGeom& dst = g.lv[l];
        const uint8_t* sp = (l == 1) ? pyr.level0 : pyr.upper + src.img_off;
        const size_t sstride = (l == 1) ? (size_t)g.S0 : (size_t)g.upper_stride;
        uint8_t* dp = pyr.upper + dst.img_off;
        const int groups = (dst.cols + 3) / 4;
        const int total = groups * dst.rows;
        const bool vec_ok = (src.cols % 8 == 0) && (((uintptr_t)sp) % 8 == 0) && (sstride % 8 == 0) && (dst.cols % 4 == 0) &&
                            (((uintptr_t)dp) % 4 == 0) && (g.upper_stride % 4 == 0);
        dim3 grid((total + 255) / 256, n_pairs);
        hipLaunchKernelGGL(halve_mean_kernel, grid, dim3(256), 0, s, sp, sstride, src.cols, dp, (size_t)g.upper_stride, dst.rows,
                           dst.cols, vec_ok ? 1 : 0);
    }
}

// ------------------------------------------------------------------------------------------------------------
// record writing shared by the keyframe kernels
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void write_record(const Records& rec, size_t slot, const Intr& k, int x, int y, float iz, int gx, int gy,
                                             uint8_t tmpl) {
    const V3 P = back_project(k, (float)x, (float)y, 1.0f / iz);  // lm_optimizer.rs:215
    float J[6];
    warp_jacobian_at((float)gx, (float)gy, (float)x, (float)y, iz, k, J);
    rec.A[slot] = make_float4(P.x, P.y, P.z, (float)tmpl);
    rec.B[slot] = make_float4(J[0], J[1], J[2], J[3]);
    rec.C[slot] = make_float2(J[4], J[5]);
    rec.XY[slot] = (uint32_t)x | ((uint32_t)y << 16);
    rec.IZ[slot] = iz;
}
__device__ __forceinline__ void write_empty(const Records& rec, size_t slot) {
    rec.A[slot] = make_float4(0.f, 0.f, 0.f, -1.0f);
    rec.B[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    rec.C[slot] = make_float2(0.f, 0.f);
    rec.XY[slot] = VORS_INVALID_XY;
    rec.IZ[slot] = 0.f;
}

// ------------------------------------------------------------------------------------------------------------
// keyframe precompute, coarse-to-fine candidates: ONE WAVEFRONT PER ROOT (= pixel of the coarsest level).
//
// The reference's masks form a forest: each coarsest-level pixel roots a quad-tree in which every selected parent
// selects its best child, plus the second best iff second > third + thresh (coarse_to_fine.rs:73-89). A node of
// level l therefore has at most 2^(L-1-l) selected descendants per root: the slot grid [root][k] below. Parent
// slot k hands its first choice to child slot 2k and its second to 2k+1. Lanes are (node, child) pairs: 16 nodes x
// 4 children per step, the 4-element ranking done with in-quad shuffles. Inverse depths are then fused bottom-up
// (inverse_depth.rs:81-98; at most two known children per parent here, so the [a,b,c,d] order cannot matter).
// ------------------------------------------------------------------------------------------------------------
#define KF_WAVES 4
__global__ __launch_bounds__(64 * KF_WAVES) void keyframe_sparse_kernel(Geom g, const uint8_t* __restrict__ kf0,
                                                                         const uint8_t* __restrict__ kfu,
                                                                         const uint16_t* __restrict__ depth, Records rec) {
    __shared__ uint32_t s_xy[KF_WAVES][256];
    __shared__ uint32_t s_gr[KF_WAVES][256];
    __shared__ float s_d[KF_WAVES][256];
    __shared__ float s_v[KF_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.y;
    const int n_roots = g.root_rows * g.root_cols;
    const int root = blockIdx.x * KF_WAVES + wave;
    const bool active = root < n_roots;
    const int L = g.L;
    uint32_t* xy = s_xy[wave];
    uint32_t* gr = s_gr[wave];
    float* sd = s_d[wave];
    float* sv = s_v[wave];

    if (lane == 0) {
        if (active) {
            const int rx = root % g.root_cols, ry = root / g.root_cols;
            int gx, gy;
            grad_at(g, kf0, kfu, pair, L - 1, rx, ry, &gx, &gy);
            xy[0] = (uint32_t)rx | ((uint32_t)ry << 16);
            gr[0] = ((uint32_t)gx & 0xffffu) | ((uint32_t)gy << 16);
        } else {
            xy[0] = VORS_INVALID_XY;
            gr[0] = 0;
        }
    }
    __syncthreads();

    // ---- top-down selection: level l -> l-1
    const uint32_t thresh = (uint32_t)g.thresh & 0xffffu;
    for (int l = L - 1; l >= 1; --l) {
        const int cap = 1 << (L - 1 - l);
        const int off = cap - 1, offc = 2 * cap - 1;
        for (int base = 0; base < cap; base += 16) {
            const int k = base + (lane >> 2), c = lane & 3;
            const bool in = k < cap;
            const uint32_t pxy = in ? xy[off + k] : VORS_INVALID_XY;
            const bool pvalid = pxy != VORS_INVALID_XY;
            const int cx = 2 * (int)(pxy & 0xffffu) + (c >> 1), cy = 2 * (int)(pxy >> 16) + (c & 1);
            int gx = 0, gy = 0;
            if (pvalid) grad_at(g, kf0, kfu, pair, l - 1, cx, cy, &gx, &gy);
            const uint32_t g2 = (uint32_t)(gx * gx + gy * gy) & 0xffffu;  // `as u16` wrap, gradient.rs:39-43
            uint32_t v[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) v[m] = (uint32_t)__shfl((int)g2, (lane & ~3) + m);
            int rk[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                int r = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) r += (v[j] < v[m] || (v[j] == v[m] && j < m)) ? 1 : 0;
                rk[m] = r;
            }
            uint32_t second = 0, third = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (rk[m] == 2) second = v[m];
                if (rk[m] == 1) third = v[m];
            }
            const bool keep2 = second > ((third + thresh) & 0xffffu);  // u16 wrapping add, coarse_to_fine.rs:85
            const int myrank = rk[c];
            if (in) {
                const uint32_t cxy = (uint32_t)cx | ((uint32_t)cy << 16);
                const uint32_t cg = ((uint32_t)gx & 0xffffu) | ((uint32_t)gy << 16);
                if (myrank == 3) {
                    xy[offc + 2 * k] = pvalid ? cxy : VORS_INVALID_XY;
                    gr[offc + 2 * k] = cg;
                } else if (myrank == 2) {
                    xy[offc + 2 * k + 1] = (pvalid && keep2) ? cxy : VORS_INVALID_XY;
                    gr[offc + 2 * k + 1] = cg;
                }
            }
        }
        __syncthreads();
    }

    // ---- level 0: inverse depth from the depth map (inverse_depth.rs:24-29); unknown depth -> not a point
    {
        const int cap = 1 << (L - 1), off = cap - 1;
        for (int k = lane; k < cap; k += 64) {
            const uint32_t p = xy[off + k];
            if (p != VORS_INVALID_XY) {
                const uint16_t dz = depth[(size_t)pair * g.S0 + (size_t)(p >> 16) * g.lv[0].cols + (p & 0xffffu)];
                if (dz == 0) {
                    xy[off + k] = VORS_INVALID_XY;
                } else {
                    sd[off + k] = g.depth_scale / (float)dz;
                    sv[off + k] = g.idepth_variance;
                }
            }
        }
    }
    __syncthreads();
    // ---- bottom-up fusion (strategy_dso_mean, inverse_depth.rs:81-98)
    for (int l = 1; l < L; ++l) {
        const int cap = 1 << (L - 1 - l);
        const int off = cap - 1, offc = 2 * cap - 1;
        for (int k = lane; k < cap; k += 64) {
            const bool k1 = xy[offc + 2 * k] != VORS_INVALID_XY, k2 = xy[offc + 2 * k + 1] != VORS_INVALID_XY;
            if (k1 && k2) {
                const float d1 = sd[offc + 2 * k], v1 = sv[offc + 2 * k], d2 = sd[offc + 2 * k + 1], v2 = sv[offc + 2 * k + 1];
                const float sum = v1 + v2;
                sd[off + k] = (d1 * v1 + d2 * v2) / sum;
                sv[off + k] = sum;
            } else if (k1 || k2) {
                const int src = k1 ? offc + 2 * k : offc + 2 * k + 1;
                sd[off + k] = sd[src];
                sv[off + k] = sv[src];
            } else {
                xy[off + k] = VORS_INVALID_XY;
            }
        }
        __syncthreads();
    }
    if (!active) return;
    // ---- records
    for (int l = 0; l < L; ++l) {
        const int cap = 1 << (L - 1 - l), off = cap - 1;
        const uint8_t* img = level_ptr(g, kf0, kfu, pair, l);
        const size_t slot0 = (size_t)pair * g.slots_total + g.lv[l].slot_off + (size_t)root * cap;
        for (int k = lane; k < cap; k += 64) {
            const uint32_t p = xy[off + k];
            if (p != VORS_INVALID_XY) {
                const int x = (int)(p & 0xffffu), y = (int)(p >> 16);
                const uint32_t gg = gr[off + k];
                write_record(rec, slot0 + k, g.lv[l].k, x, y, sd[off + k], (int)(int16_t)(gg & 0xffffu), (int)(int16_t)(gg >> 16),
                             img[(size_t)y * g.lv[l].cols + x]);
            } else {
                write_empty(rec, slot0 + k);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// keyframe precompute, dense candidates (extension): all-true level-0 mask, any image size. Slots are the pixels
// (slot = y*cols + x at each level). The inverse-depth pyramid is built level by level in the IZ / V planes,
// fusing the four children in [a,b,c,d] order (inverse_depth.rs:49-66,81-98); V < 0 marks Unknown.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_idepth_level0_kernel(Geom g, const uint16_t* __restrict__ depth, Records rec) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.lv[0].n_slots) return;
    const size_t slot = (size_t)pair * g.slots_total + g.lv[0].slot_off + t;
    const uint16_t dz = depth[(size_t)pair * g.S0 + t];
    rec.IZ[slot] = dz != 0 ? g.depth_scale / (float)dz : 0.f;  // inverse_depth.rs:24-29
    rec.V[slot] = dz != 0 ? g.idepth_variance : -1.0f;
}
__global__ __launch_bounds__(256) void dense_idepth_halve_kernel(Geom g, int l, Records rec) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols;
    if (t >= rows * cols) return;
    const int y = t / cols, x = t - y * cols;
    const int fc = g.lv[l - 1].cols;
    const size_t cb = (size_t)pair * g.slots_total + g.lv[l - 1].slot_off + (size_t)(2 * y) * fc + 2 * x;
    // children a=(2i,2j) b=(2i+1,2j) c=(2i,2j+1) d=(2i+1,2j+1) with i=row, j=col   (multires.rs:80-83)
    const size_t idx[4] = {cb, cb + fc, cb + 1, cb + fc + 1};
    float dv[4], vv[4];
    int n = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float v = rec.V[idx[m]];
        if (v >= 0.f) {
            dv[n] = rec.IZ[idx[m]];
            vv[n] = v;
            ++n;
        }
    }
    float od = 0.f, ov = -1.0f;
    if (n == 1) {
        od = dv[0];
        ov = vv[0];
    } else if (n == 2) {
        ov = vv[0] + vv[1];
        od = (dv[0] * vv[0] + dv[1] * vv[1]) / ov;
    } else if (n == 3) {
        ov = vv[0] + vv[1] + vv[2];
        od = (dv[0] * vv[0] + dv[1] * vv[1] + dv[2] * vv[2]) / ov;
    } else if (n == 4) {
        ov = vv[0] + vv[1] + vv[2] + vv[3];
        od = (dv[0] * vv[0] + dv[1] * vv[1] + dv[2] * vv[2] + dv[3] * vv[3]) / ov;
    }
    const size_t slot = (size_t)pair * g.slots_total + g.lv[l].slot_off + t;
    rec.IZ[slot] = od;
    rec.V[slot] = ov;
}
__global__ __launch_bounds__(256) void dense_records_kernel(Geom g, int l, const uint8_t* __restrict__ kf0,
                                                             const uint8_t* __restrict__ kfu, Records rec) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols;
    if (t >= rows * cols) return;
    const int y = t / cols, x = t - y * cols;
    const size_t slot = (size_t)pair * g.slots_total + g.lv[l].slot_off + t;
    if (rec.V[slot] >= 0.f) {
        int gx, gy;
        grad_at(g, kf0, kfu, pair, l, x, y, &gx, &gy);
        write_record(rec, slot, g.lv[l].k, x, y, rec.IZ[slot], gx, gy, level_ptr(g, kf0, kfu, pair, l)[t]);
    } else {
        write_empty(rec, slot);
    }
}

void launch_keyframe(const Geom& g, Pyramid kf, const uint16_t* depth, Records rec, int n_pairs, hipStream_t s) {
    const int n_roots = g.root_rows * g.root_cols;
    if (g.mode == VORS_CANDIDATES_DENSE) {
        hipLaunchKernelGGL(dense_idepth_level0_kernel, dim3((g.lv[0].n_slots + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, rec);
        for (int l = 1; l < g.L; ++l)
            hipLaunchKernelGGL(dense_idepth_halve_kernel, dim3((g.lv[l].n_slots + 255) / 256, n_pairs), dim3(256), 0, s, g, l, rec);
        for (int l = 0; l < g.L; ++l)
            hipLaunchKernelGGL(dense_records_kernel, dim3((g.lv[l].n_slots + 255) / 256, n_pairs), dim3(256), 0, s, g, l, kf.level0,
                               kf.upper, rec);
    } else {
        dim3 grid((n_roots + KF_WAVES - 1) / KF_WAVES, n_pairs);
        hipLaunchKernelGGL(keyframe_sparse_kernel, grid, dim3(64 * KF_WAVES), 0, s, g, kf.level0, kf.upper, depth, rec);
    }
}

// ------------------------------------------------------------------------------------------------------------
// LM evaluation: eval_energy + compute_eval_data fused in ONE pass (lm_optimizer.rs:68-107).
// Per valid slot: warp (lm_optimizer.rs:213-219), bilinear interpolation with the reference's strict inside test
// (lm_optimizer.rs:227-251), residual r = I(w(x)) - T(x), then 29 partial sums: sum r^2, count, g = sum J r (6),
// H = sum J J^T (21 unique). J J^T is recomputed from J instead of streaming the reference's precomputed 6x6.
// ------------------------------------------------------------------------------------------------------------
#define NACC 29

struct EvalCtx {
    const uint8_t* img;  // current image of this level, row-major
    int rows, cols;
    Intr k;
    const float4* A;
    const float4* B;
    const float2* C;
    int n_slots;
    float huber;
};

template <int BLOCK, bool WRITE_RES>
__device__ __forceinline__ void eval_accumulate(const EvalCtx& c, const Iso& model, float acc[NACC], float* residuals) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    const float wm2 = (float)(c.cols - 2), hm2 = (float)(c.rows - 2);
    for (int i = threadIdx.x; i < c.n_slots; i += BLOCK) {
        const float4 a = c.A[i];
        float res = __builtin_nanf("");
        if (a.w >= 0.f) {
            const V3 p2 = iso_transform_point(model, V3{a.x, a.y, a.z});
            float u, v;
            project_uv(c.k, p2, &u, &v);
            const float uf = floorf(u), vf = floorf(v);
            if (uf >= 0.f && uf < wm2 && vf >= 0.f && vf < hm2) {
                const int u0 = (int)uf, v0 = (int)vf;
                const uint8_t* p = c.img + (size_t)v0 * c.cols + u0;
                const float vu_00 = (float)p[0], vu_01 = (float)p[1];
                const float vu_10 = (float)p[c.cols], vu_11 = (float)p[c.cols + 1];
                const float fa = u - uf, fb = v - vf;
                const float im = (1.0f - fb) * (1.0f - fa) * vu_00 + fb * (1.0f - fa) * vu_10 + (1.0f - fb) * fa * vu_01 +
                                 fb * fa * vu_11;
                const float r = im - a.w;
                res = r;
                const float4 jb = c.B[i];
                const float2 jc = c.C[i];
                const float J[6] = {jb.x, jb.y, jb.z, jb.w, jc.x, jc.y};
                float w = 1.0f, wr = r;
                if (c.huber > 0.f) {  // extension (not in the reference)
                    const float ar = fabsf(r);
                    if (ar <= c.huber) {
                        acc[0] = fmaf(r, r, acc[0]);
                    } else {
                        acc[0] += c.huber * (2.0f * ar - c.huber);
                        w = c.huber / ar;
                        wr = w * r;
                    }
                } else {
                    acc[0] = fmaf(r, r, acc[0]);
                }
                acc[1] += 1.0f;
#pragma unroll
                for (int q = 0; q < 6; ++q) acc[2 + q] = fmaf(J[q], wr, acc[2 + q]);
                int h = 8;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const float jq = (c.huber > 0.f) ? w * J[q] : J[q];
#pragma unroll
                    for (int s = q; s < 6; ++s) {
                        acc[h] = fmaf(jq, J[s], acc[h]);
                        ++h;
                    }
                }
            }
        }
        if (WRITE_RES) residuals[i] = res;
    }
}

// Workgroup reduction of the 29 partial sums: xor-butterfly inside each wavefront, then a fixed-order sum over the
// wavefronts through LDS. Every thread returns with the totals in acc[]. Deterministic for a given BLOCK.
template <int BLOCK>
__device__ __forceinline__ void block_reduce(float acc[NACC], float* s_part /* [BLOCK/64][32] */, float* s_tot /* [32] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        float v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        acc[i] = v;
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) s_part[wave * 32 + i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) t += s_part[w * 32 + threadIdx.x];
        s_tot[threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = s_tot[i];
}

struct EvalOut {  // EvalData of lm_optimizer.rs:31-40 (model kept separately)
    float energy;
    float g[6];
    float h[36];
};
__device__ __forceinline__ void unpack_eval(const float acc[NACC], EvalOut* e) {
    e->energy = acc[0] / acc[1];  // energy_sum / residuals.len()  (0/0 = NaN as in the reference)
#pragma unroll
    for (int q = 0; q < 6; ++q) e->g[q] = acc[2 + q];
    int h = 8;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int s = q; s < 6; ++s) {
            e->h[q * 6 + s] = acc[h];
            e->h[s * 6 + q] = acc[h];
            ++h;
        }
}

// optimizer::State::iterative_solve (optimizer.rs:57-70) with LMOptimizerState's init / step / eval / stop_criterion
// (lm_optimizer.rs:113-192), executed redundantly by every thread on workgroup-uniform values.
// Returns false when step() fails (Cholesky); *model is then left untouched (the level's progress is discarded like
// the reference's `Err(err) => break`, inverse_compositional.rs:195-199).
template <int BLOCK>
__device__ bool solve_level(const EvalCtx& c, Iso* model, int* nb_iter_out, float* energy_out, float* lm_coef_out, float* s_part,
                            float* s_tot) {
    float acc[NACC];
    EvalOut cur;
    Iso cur_model = *model;
    eval_accumulate<BLOCK, false>(c, cur_model, acc, nullptr);  // init: lm_optimizer.rs:113-118
    block_reduce<BLOCK>(acc, s_part, s_tot);
    unpack_eval(acc, &cur);
    float lm_coef = 0.1f;
    int nb_iter = 0;
    for (;;) {
        nb_iter += 1;
        Iso cand;
        if (!lm_step(cur.h, cur.g, cur_model, lm_coef, &cand)) return false;  // step(): lm_optimizer.rs:123-136
        eval_accumulate<BLOCK, false>(c, cand, acc, nullptr);                  // eval(): lm_optimizer.rs:140-149
        block_reduce<BLOCK>(acc, s_part, s_tot);
        const float energy = acc[0] / acc[1];
        const bool too_many_iterations = nb_iter > 20;  // stop_criterion: lm_optimizer.rs:156-192
        if (energy > cur.energy) {                      // Err(energy)
            if (too_many_iterations) break;
            lm_coef *= 10.0f;
            continue;
        }
        const float d_energy = cur.energy - energy;
        unpack_eval(acc, &cur);
        cur_model = cand;
        if (too_many_iterations) break;
        lm_coef = 0.1f * lm_coef;
        if (!(d_energy > 1.0f)) break;
    }
    *model = cur_model;
    *nb_iter_out = nb_iter;
    *energy_out = cur.energy;
    *lm_coef_out = lm_coef;
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// Tracker::track for a batch: one workgroup per frame pair, all levels, all LM iterations, keyframe test.
// ------------------------------------------------------------------------------------------------------------
#define LM_BLOCK 256
__global__ __launch_bounds__(LM_BLOCK) void lm_track_kernel(Geom g, const uint8_t* __restrict__ cur0,
                                                             const uint8_t* __restrict__ curu, Records rec,
                                                             const float* __restrict__ prev_poses7,
                                                             const float* __restrict__ kf_poses7, float* __restrict__ out_poses7,
                                                             int32_t* __restrict__ out_status,
                                                             vors_pair_stats* __restrict__ out_stats) {
    __shared__ float s_part[(LM_BLOCK / 64) * 32];
    __shared__ float s_tot[32];
    const int pair = blockIdx.x;
    const Iso prev_pose = prev_poses7 ? iso_load(prev_poses7 + 7 * pair) : iso_identity();
    const Iso kf_pose = kf_poses7 ? iso_load(kf_poses7 + 7 * pair) : iso_identity();
    Iso lm_model = iso_mul(iso_inverse(prev_pose), kf_pose);  // inverse_compositional.rs:177
    bool went_well = true;
    const size_t rbase = (size_t)pair * g.slots_total;
    for (int lvl = g.L - 1; lvl >= 0; --lvl) {
        EvalCtx c;
        c.img = level_ptr(g, cur0, curu, pair, lvl);
        c.rows = g.lv[lvl].rows;
        c.cols = g.lv[lvl].cols;
        c.k = g.lv[lvl].k;
        c.A = rec.A + rbase + g.lv[lvl].slot_off;
        c.B = rec.B + rbase + g.lv[lvl].slot_off;
        c.C = rec.C + rbase + g.lv[lvl].slot_off;
        c.n_slots = g.lv[lvl].n_slots;
        c.huber = g.huber_delta;
        int nb_iter = 0;
        float energy = 0.f, lm_coef = 0.f;
        const bool ok = solve_level<LM_BLOCK>(c, &lm_model, &nb_iter, &energy, &lm_coef, s_part, s_tot);
        if (out_stats && threadIdx.x == 0) {
            out_stats[pair].nb_iter[lvl] = ok ? nb_iter : 0;
            out_stats[pair].energy[lvl] = ok ? energy : 0.f;
        }
        if (!ok) {
            went_well = false;
            if (out_stats && threadIdx.x == 0)
                for (int l2 = lvl - 1; l2 >= 0; --l2) {
                    out_stats[pair].nb_iter[l2] = 0;
                    out_stats[pair].energy[l2] = 0.f;
                }
            break;
        }
    }
    // keyframe test on the coarsest level (inverse_compositional.rs:211-224)
    float flow_sum = 0.f, flow_n = 0.f;
    {
        const int lvl = g.L - 1;
        const float4* A = rec.A + rbase + g.lv[lvl].slot_off;
        const uint32_t* XY = rec.XY + rbase + g.lv[lvl].slot_off;
        for (int i = threadIdx.x; i < g.lv[lvl].n_slots; i += LM_BLOCK) {
            const float4 a = A[i];
            if (a.w >= 0.f) {
                const uint32_t p = XY[i];
                const float x = (float)(p & 0xffffu), y = (float)(p >> 16);
                float u, v;
                project_uv(g.lv[lvl].k, iso_transform_point(lm_model, V3{a.x, a.y, a.z}), &u, &v);
                flow_sum += fabsf(x - u) + fabsf(y - v);
                flow_n += 1.0f;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            flow_sum += __shfl_xor(flow_sum, m);
            flow_n += __shfl_xor(flow_n, m);
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            s_part[(threadIdx.x >> 6) * 2] = flow_sum;
            s_part[(threadIdx.x >> 6) * 2 + 1] = flow_n;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float fs = 0.f, fn = 0.f;
        for (int w = 0; w < LM_BLOCK / 64; ++w) {
            fs += s_part[2 * w];
            fn += s_part[2 * w + 1];
        }
        const float optical_flow = fs / fn;
        const Iso pose = went_well ? iso_mul(kf_pose, iso_inverse(lm_model)) : prev_pose;  // inverse_compositional.rs:206-208
        iso_store(pose, out_poses7 + 7 * pair);
        out_status[pair] = went_well ? VORS_TRACK_OK : VORS_TRACK_OPTIMIZER_FAILED_POSE_KEPT;
        if (out_stats) {
            iso_store(lm_model, out_stats[pair].lm_model);
            out_stats[pair].optical_flow = optical_flow;
            out_stats[pair].change_keyframe = (optical_flow >= 1.0f) ? 1 : 0;
        }
    }
    // usable candidates per level (diagnostics)
    if (out_stats) {
        for (int lvl = 0; lvl < g.L; ++lvl) {
            const float4* A = rec.A + rbase + g.lv[lvl].slot_off;
            int n = 0;
            for (int i = threadIdx.x; i < g.lv[lvl].n_slots; i += LM_BLOCK) n += (A[i].w >= 0.f) ? 1 : 0;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) n += __shfl_xor(n, m);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = (float)n;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = 0.f;
                for (int w = 0; w < LM_BLOCK / 64; ++w) t += s_part[w];
                out_stats[pair].n_points[lvl] = (int)t;
            }
        }
        if (threadIdx.x == 0)
            for (int lvl = g.L; lvl < VORS_MAX_LEVELS; ++lvl) {
                out_stats[pair].nb_iter[lvl] = 0;
                out_stats[pair].n_points[lvl] = 0;
                out_stats[pair].energy[lvl] = 0.f;
            }
    }
}

void launch_lm_track(const Geom& g, Pyramid cur, Records rec, const float* prev_poses7, const float* kf_poses7,
                     float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs, hipStream_t s) {
    hipLaunchKernelGGL(lm_track_kernel, dim3(n_pairs), dim3(LM_BLOCK), 0, s, g, cur.level0, cur.upper, rec, prev_poses7, kf_poses7,
                       out_poses7, out_status, out_stats);
}

// ------------------------------------------------------------------------------------------------------------
// Operator level (one pyramid level, explicit observations)
// ------------------------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(LM_BLOCK) void lm_eval_obs_kernel(EvalCtx c, const float* __restrict__ model7, float* __restrict__ out,
                                                                float* __restrict__ residuals) {
    __shared__ float s_part[(LM_BLOCK / 64) * 32];
    __shared__ float s_tot[32];
    float acc[NACC];
    const Iso model = iso_load(model7);
    if (residuals)
        eval_accumulate<LM_BLOCK, true>(c, model, acc, residuals);
    else
        eval_accumulate<LM_BLOCK, false>(c, model, acc, nullptr);
    block_reduce<LM_BLOCK>(acc, s_part, s_tot);
    if (threadIdx.x == 0) {
        EvalOut e;
        unpack_eval(acc, &e);
        out[0] = e.energy;
        out[1] = acc[1];
        for (int q = 0; q < 6; ++q) out[2 + q] = e.g[q];
        for (int q = 0; q < 36; ++q) out[8 + q] = e.h[q];
    }
}
static EvalCtx make_ctx(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber) {
    EvalCtx c;
    c.img = image;
    c.rows = rows;
    c.cols = cols;
    c.k = k;
    c.A = rec.A;
    c.B = rec.B;
    c.C = rec.C;
    c.n_slots = n;
    c.huber = huber;
    return c;
}
void launch_lm_eval_obs(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta,
                        const float* model7, float* out, float* residuals, hipStream_t s) {
    hipLaunchKernelGGL(lm_eval_obs_kernel, dim3(1), dim3(LM_BLOCK), 0, s, make_ctx(k, rows, cols, image, n, rec, huber_delta), model7,
                       out, residuals);
}

__global__ __launch_bounds__(LM_BLOCK) void lm_solve_obs_kernel(EvalCtx c, const float* __restrict__ model7, float* __restrict__ out) {
    __shared__ float s_part[(LM_BLOCK / 64) * 32];
    __shared__ float s_tot[32];
    Iso model = iso_load(model7);
    int nb_iter = 0;
    float energy = 0.f, lm_coef = 0.f;
    const bool ok = solve_level<LM_BLOCK>(c, &model, &nb_iter, &energy, &lm_coef, s_part, s_tot);
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
    hipLaunchKernelGGL(lm_solve_obs_kernel, dim3(1), dim3(LM_BLOCK), 0, s, make_ctx(k, rows, cols, image, n, rec, huber_delta), model7,
                       out);
}

// ------------------------------------------------------------------------------------------------------------
// synthetic scene renderer (bench / test tooling)
// ------------------------------------------------------------------------------------------------------------
__global__ void synth_pairs_kernel(uint64_t seed0, int rows, int cols, vors_synth::CameraD cam, double motion_scale,
                                   int invalid_percent, uint8_t* kf_gray, uint16_t* kf_depth, uint8_t* cur_gray,
                                   uint16_t* cur_depth, float* gt_models7) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t seed = seed0 + (uint64_t)pair;
    double xi[6];
    vors_synth::pair_twist(seed, motion_scale, xi);
    const vors_synth::RigidD m = vors_synth::se3_exp_d(xi);
    if (t == 0 && gt_models7) vors_synth::rigid_to_pose7(m, xi, gt_models7 + 7 * pair);
    if (t >= rows * cols) return;
    const int y = t / cols, x = t - y * cols;
    const size_t o = (size_t)pair * rows * cols + t;
    const double zero[6] = {0, 0, 0, 0, 0, 0};
    const vors_synth::RigidD id = vors_synth::se3_exp_d(zero);
    uint8_t gq;
    uint16_t dq;
    vors_synth::render_pixel(seed, 0, cam, id, x, y, invalid_percent, &gq, &dq);
    kf_gray[o] = gq;
    kf_depth[o] = dq;
    vors_synth::render_pixel(seed, 1, cam, m, x, y, invalid_percent, &gq, &dq);
    cur_gray[o] = gq;
    if (cur_depth) cur_depth[o] = dq;
}
void launch_synth_pairs(uint64_t seed0, int n_pairs, int rows, int cols, const double cam5[5], double motion_scale,
                        int invalid_percent, uint8_t* kf_gray, uint16_t* kf_depth, uint8_t* cur_gray, uint16_t* cur_depth,
                        float* gt_models7, hipStream_t s) {
    vors_synth::CameraD cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    dim3 grid((rows * cols + 255) / 256, n_pairs);
    hipLaunchKernelGGL(synth_pairs_kernel, grid, dim3(256), 0, s, seed0, rows, cols, cam, motion_scale, invalid_percent, kf_gray,
                       kf_depth, cur_gray, cur_depth, gt_models7);
}

}  // namespace vors
