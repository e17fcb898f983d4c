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

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "device_common.h"
#include "engine.h"
#include "synth_scene.h"

namespace vors {

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

// ------------------------------------------------------------------------------------------------------------
// mean pyramid, up to five halvings per launch: a wavefront owns a 32 x 128 tile of level 0 and everything above it (16 x 64 of level 1
// ... 1 x 4 of level 5).  Level 0 is read once with 16-byte loads; every coarser level is written from registers and never read back from
// memory (one level per launch moves 1.67 x the level-0 bytes, this 1.33 x).  Same arithmetic: ((a+b+c+d) as u16 / 4) as u8, and
// halve's floor on odd sizes (multires.rs:67-88) means a pixel of level l only ever depends on pixels INSIDE level l-1.
// ------------------------------------------------------------------------------------------------------------
struct PyrLevels {
    int n;                       // halvings done here (1..5)
    int rows[7], cols[7];        // sizes of levels 0..n
    int off[7];                  // offset of level l (>= 1) inside a pair's upper block
};
__device__ __forceinline__ uint32_t halve_words(uint32_t w0, uint32_t w1) {
    // four pixels of two rows -> two means in bytes 0 and 2 (byte 1 is not clean: callers pick bytes).  v_perm_b32 spreads the even and
    // the odd bytes of a word into 16-bit lanes (selector 0x0c = a zero byte).
    const uint32_t s = __builtin_amdgcn_perm(0, w0, 0x0c020c00u) + __builtin_amdgcn_perm(0, w0, 0x0c030c01u) +
                       __builtin_amdgcn_perm(0, w1, 0x0c020c00u) + __builtin_amdgcn_perm(0, w1, 0x0c030c01u);
    return s >> 2;
}
__device__ __forceinline__ uint32_t pack_means(uint32_t lo, uint32_t hi) {
    // bytes 0 and 2 of lo, then bytes 0 and 2 of hi
    return __builtin_amdgcn_perm(hi, lo, 0x06040200u);
}
__global__ __launch_bounds__(1024) void pyramid_fused_kernel(const uint8_t* __restrict__ level0, size_t stride0, uint8_t* __restrict__ upper,
                                                             size_t upper_stride, PyrLevels lv, int tiles_x, int n_tiles, int wgs_per_image,
                                                             int n_images) {
    // A wavefront owns a tile on its own (no LDS, no barrier) as 8 x 8 lanes: a lane reads four rows x 16 pixels (all four 16-byte
    // loads are issued before anything is used) and holds 2 x 8 of level 1 and 1 x 4 of level 2; the partner rows of the next
    // halvings are 8, 16 and 32 lanes away.
    // Workgroup b runs on XCD b % 8 (observed, MI355X_MICROARCH.md): all workgroups of an image go to ONE XCD, so that the pieces of
    // a 128-byte line of the small levels, which come from several tiles, meet in that XCD's L2 instead of leaving as partial writes
    // from several L2s.  Only speed depends on it.
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x >> 3;
    const int image = (q / wgs_per_image) * 8 + (blockIdx.x & 7);
    const int tile = (q % wgs_per_image) * (blockDim.x >> 6) + (threadIdx.x >> 6);  // a workgroup = one row of tiles when that fits
    if (image >= n_images || tile >= n_tiles) return;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int rp = lane >> 3, cg = lane & 7;
    const int ty = tyi * 32, x0 = txi * 128 + 16 * cg;
    uint8_t* up = upper + (size_t)image * upper_stride;
    const uint8_t* img = level0 + (size_t)image * stride0;
    const int y0 = ty + 4 * rp;
    const bool in_x = x0 < lv.cols[0];  // cols[0] % 16 == 0: a group is inside or outside as a whole
    const bool in0 = in_x && y0 + 1 < lv.rows[0], in1 = in_x && y0 + 3 < lv.rows[0];
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
    const uint8_t* src = img + (uint32_t)(y0 * lv.cols[0] + x0);
    if (in0) {
        r0 = *reinterpret_cast<const uint4*>(src);
        r1 = *reinterpret_cast<const uint4*>(src + lv.cols[0]);
    }
    if (in1) {
        r2 = *reinterpret_cast<const uint4*>(src + 2 * lv.cols[0]);
        r3 = *reinterpret_cast<const uint4*>(src + 3 * lv.cols[0]);
    }
    uint2 a0, a1;
    a0.x = pack_means(halve_words(r0.x, r1.x), halve_words(r0.y, r1.y));
    a0.y = pack_means(halve_words(r0.z, r1.z), halve_words(r0.w, r1.w));
    a1.x = pack_means(halve_words(r2.x, r3.x), halve_words(r2.y, r3.y));
    a1.y = pack_means(halve_words(r2.z, r3.z), halve_words(r2.w, r3.w));
    uint8_t* d1 = up + lv.off[1] + (uint32_t)(((ty >> 1) + 2 * rp) * lv.cols[1] + (x0 >> 1));
    if (in0) *reinterpret_cast<uint2*>(d1) = a0;
    if (in1) *reinterpret_cast<uint2*>(d1 + lv.cols[1]) = a1;
    if (lv.n < 2) return;
    // level 2: a lane's own two rows; 8 pixels -> 4 (cols[2] % 4 == 0)
    const uint32_t c = pack_means(halve_words(a0.x, a1.x), halve_words(a0.y, a1.y));
    if (in1) *reinterpret_cast<uint32_t*>(up + lv.off[2] + (uint32_t)(((ty >> 2) + rp) * lv.cols[2] + (x0 >> 2))) = c;
    if (lv.n < 3) return;
    // level 3: rows (rp, rp + 1) for even rp; 4 pixels -> 2 (cols[3] % 2 == 0)
    const uint32_t d = __builtin_amdgcn_perm(0, halve_words(c, __shfl_down(c, 8)), 0x0c0c0200u);
    if ((rp & 1) == 0) {
        const int y = (ty >> 3) + (rp >> 1), x = x0 >> 3;
        if (y < lv.rows[3] && x < lv.cols[3]) *reinterpret_cast<uint16_t*>(up + lv.off[3] + (uint32_t)(y * lv.cols[3] + x)) = (uint16_t)d;
    }
    if (lv.n < 4) return;
    // level 4: rows (rp, rp + 2) for rp % 4 == 0; 2 pixels -> 1 (v_sad_u8 against 0 adds the bytes of a word)
    const uint32_t e = __builtin_amdgcn_sad_u8(d, 0, __builtin_amdgcn_sad_u8(__shfl_down(d, 16), 0, 0)) >> 2;
    if ((rp & 3) == 0) {
        const int y = (ty >> 4) + (rp >> 2), x = x0 >> 4;
        if (y < lv.rows[4] && x < lv.cols[4]) up[lv.off[4] + (uint32_t)(y * lv.cols[4] + x)] = (uint8_t)e;
    }
    if (lv.n < 5) return;
    // level 5: rows (0, 4) and the neighbouring group
    const uint32_t s = e + __shfl_down(e, 32);
    const uint32_t f = (s + __shfl_xor(s, 1)) >> 2;
    if (rp == 0 && (cg & 1) == 0) {
        const int y = ty >> 5, x = x0 >> 5;
        if (y < lv.rows[5] && x < lv.cols[5]) up[lv.off[5] + (uint32_t)(y * lv.cols[5] + x)] = (uint8_t)f;
    }
}

static int g_pyramid_fused = -1;  // VORS_PYRAMID_FUSED=0 keeps one level per launch (development aid)

void launch_pyramid(const Geom& g, Pyramid pyr, int n_pairs, hipStream_t s) {
    if (g_pyramid_fused < 0) {
        const char* e = getenv("VORS_PYRAMID_FUSED");
        g_pyramid_fused = (e && e[0] == '0') ? 0 : 1;
    }
    int first = 1;
    if (g_pyramid_fused && g.L >= 2 && g.lv[0].cols % 16 == 0 && ((uintptr_t)pyr.level0) % 16 == 0 && g.S0 % 16 == 0 &&
        ((uintptr_t)pyr.upper) % 16 == 0 && g.upper_stride % 16 == 0) {
        PyrLevels lv{};
        lv.n = std::min(g.L - 1, 5);
        bool ok = true;
        for (int l = 0; l <= lv.n; ++l) {
            lv.rows[l] = g.lv[l].rows;
            lv.cols[l] = g.lv[l].cols;
            lv.off[l] = l ? g.lv[l].img_off : 0;
            if (l && (g.lv[l].img_off < 0 || g.lv[l].img_off % 16 != 0)) ok = false;
        }
        if (ok) {
            const int tiles_x = (g.lv[0].cols + 127) / 128, n_tiles = tiles_x * ((g.lv[0].rows + 31) / 32);
            const int waves = tiles_x <= 16 ? tiles_x : 4;
            const int wgs = (n_tiles + waves - 1) / waves;
            dim3 grid((unsigned)(wgs * ((n_pairs + 7) / 8) * 8));
            hipLaunchKernelGGL(pyramid_fused_kernel, grid, dim3(64 * waves), 0, s, pyr.level0, (size_t)g.S0, pyr.upper, (size_t)g.upper_stride, lv,
                               tiles_x, n_tiles, wgs, n_pairs);
            first = lv.n + 1;
        }
    }
    for (int l = first; l < g.L; ++l) {
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
// An empty slot is recognised by tmpl < 0 alone; the other planes of an empty slot are never read.
__device__ __forceinline__ void write_empty(const Records& rec, size_t slot) {
    rec.A[slot] = make_float4(0.f, 0.f, 0.f, -1.0f);
    rec.XY[slot] = VORS_INVALID_XY;
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
#ifndef VORS_KF_BATCH
#define VORS_KF_BATCH 2  // chunks of 64 (node, child) items whose gradient gathers travel together (1 = one round trip per chunk: round 5; measured 1 / 2 / 4: keyframe stage 1.143 / 1.112 / 1.116 ms per 4096 pairs)
#endif
// grad_tmpl_at (device_common.h) in two halves — the REQUESTS for what a pixel's gradient and grey level are made of, and their decoding —
// so that the requests of several independent items can be in flight at once. L0: the pixel is of level 0 (centred differences,
// gradient.rs:15-33), else of a level >= 1 (2x2 block gradient of the next finer level, gradient.rs:74-93; its grey level is that block's
// floored mean, multires.rs:21-31). Same loads, same integer arithmetic as grad_tmpl_at.
struct KfGradRaw {
    uint32_t w0, w1, w2;
};
template <bool L0>
__device__ __forceinline__ KfGradRaw kf_grad_request(const Geom& g, const uint8_t* level0, const uint8_t* upper, int pair, int l, int x, int y) {
    KfGradRaw r;
    if (L0) {
        const int rows = g.lv[0].rows, cols = g.lv[0].cols;
        const uint8_t* p = level0 + (size_t)pair * g.S0;
        const unsigned o = __umul24((unsigned)y, (unsigned)cols) + (unsigned)x;
        const bool interior = !(x == 0 || y == 0 || x == cols - 1 || y == rows - 1);
        const unsigned dy = interior ? (unsigned)cols : 0u;
        const unsigned base = interior ? o - 1u : min(o, (unsigned)g.S0 - 4u);  // left, centre, right as one (unaligned) dword
        __builtin_memcpy(&r.w0, p + base, 4);
        r.w1 = p[o - dy];
        r.w2 = p[o + dy];
    } else {
        const int fc = g.lv[l - 1].cols;
        const uint8_t* pb = level_ptr(g, level0, upper, pair, l - 1);
        const unsigned o = __umul24((unsigned)(2 * y), (unsigned)fc) + (unsigned)(2 * x);
        uint16_t r0, r1;  // the 2x2 block as two (possibly unaligned) 16-bit loads
        __builtin_memcpy(&r0, pb + o, 2);
        __builtin_memcpy(&r1, pb + (o + (unsigned)fc), 2);
        r.w0 = r0;
        r.w1 = r1;
        r.w2 = 0;
    }
    return r;
}
template <bool L0>
__device__ __forceinline__ void kf_grad_decode(const Geom& g, const KfGradRaw& r, int x, int y, int* gx, int* gy, int* tm) {
    if (L0) {
        const int rows = g.lv[0].rows, cols = g.lv[0].cols;
        const unsigned o = __umul24((unsigned)y, (unsigned)cols) + (unsigned)x;
        const bool interior = !(x == 0 || y == 0 || x == cols - 1 || y == rows - 1);
        const unsigned base = interior ? o - 1u : min(o, (unsigned)g.S0 - 4u);
        const uint32_t w = r.w0;
        const int c0 = (int)((w >> (8u * (interior ? 1u : o - base))) & 0xffu);
        const int l0 = interior ? (int)(w & 0xffu) : c0, r0 = interior ? (int)((w >> 16) & 0xffu) : c0;
        const int u0 = (int)r.w1, d0 = (int)r.w2;
        *tm = c0;
        *gx = (r0 - l0) / 2;  // borders: the taps alias the centre pixel -> 0, like gradient.rs:15-33
        *gy = (d0 - u0) / 2;
    } else {
        const int a = r.w0 & 0xff, c = r.w0 >> 8, b = r.w1 & 0xff, d = r.w1 >> 8;
        *gx = (c + d - a - b) / 2;
        *gy = (b - a + d - c) / 2;
        *tm = (a + b + c + d) >> 2;
    }
}
// KF_R roots per wavefront: the early tree steps have only 4 * 2^k (node, child) items per root, so several roots share
// a wavefront (more lanes busy, more independent loads in flight per wave, 1/KF_R as many waves).
// The wavefronts of a keyframe workgroup never share LDS (each owns the tree nodes of its own roots), so the steps of the descent
// only need ordering INSIDE a wavefront: LDS operations of a wavefront execute in order; this keeps the compiler from moving them.
__device__ __forceinline__ void kf_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#ifndef VORS_KF_PAD
#define VORS_KF_PAD 1  // one word of padding per root's tree: the roots of a wavefront no longer sit 2^L words apart — all in the same LDS
                       // banks (round 6 counters: 46 % of the kernel's LDS cycles were bank conflicts)
#endif
__host__ __device__ inline int kf_nodes(int L) { return (1 << L) + VORS_KF_PAD; }
// LC: the number of pyramid levels as a compile-time constant (6 and 7: BASELINE's configurations), 0 = read it from the geometry. With a
// constant the level loops unroll, the slot arithmetic (capacities, offsets, shifts) folds into immediates and the chunk loops become
// straight-line code — the kernel is bound by the instructions a wavefront issues, a third of them scalar loop and address bookkeeping.
template <int KF_R, int LC>
__global__ __launch_bounds__(64 * KF_WAVES) void keyframe_sparse_kernel(Geom g, const uint8_t* __restrict__ kf0,
                                                                         const uint8_t* __restrict__ kfu,
                                                                         const uint16_t* __restrict__ depth, Records rec) {
    const int NODES = kf_nodes(LC ? LC : g.L);  // >= 2^L - 1 tree nodes per root
    extern __shared__ __attribute__((aligned(16))) char kf_smem[];
    uint32_t* s_xy = reinterpret_cast<uint32_t*>(kf_smem);  // [KF_WAVES][KF_R][NODES]
    uint32_t* s_gr = s_xy + KF_WAVES * KF_R * NODES;
    float* s_d = reinterpret_cast<float*>(s_gr + KF_WAVES * KF_R * NODES);
    float* s_v = s_d + KF_WAVES * KF_R * NODES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int n_roots = g.root_rows * g.root_cols;
    const int root0 = (blockIdx.x * KF_WAVES + wave) * KF_R;  // first root of this wavefront
    const int L = LC ? LC : g.L;
    uint32_t* xy = s_xy + wave * KF_R * NODES;
    uint32_t* gr = s_gr + wave * KF_R * NODES;
    float* sd = s_d + wave * KF_R * NODES;
    float* sv = s_v + wave * KF_R * NODES;

    if (lane < KF_R) {
        const int root = root0 + lane;
        if (root < n_roots) {
            const int rx = root % g.root_cols, ry = root / g.root_cols;
            int gx, gy, tm;
            grad_tmpl_at(g, kf0, kfu, pair, L - 1, rx, ry, &gx, &gy, &tm);
            xy[lane * NODES] = (uint32_t)rx | ((uint32_t)ry << 16);
            gr[lane * NODES] = slim_pack_tg(tm, gx, gy);  // the record's third word, carried through the descent (no gather later)
        } else {
            xy[lane * NODES] = VORS_INVALID_XY;
            gr[lane * NODES] = 0;
        }
    }
    kf_wave_sync();

    // ---- top-down selection: level l -> l-1. Items = (root, node k, child c); the 4 children of a node sit in one quad.
    // Round 6: the descent is a chain of dependent gathers (the children's gradients need the parents the previous step selected), and at the
    // two finest steps a wavefront has 2 and 4 chunks of 64 items to look at: their gathers are REQUESTED TOGETHER (kf_grad_request, at most
    // VORS_KF_BATCH chunks) and only then decoded and ranked — one memory round trip per batch instead of one per chunk. Worth 3 % of the stage, no more:
    // the kernel is bound by the ~1500 instructions a wavefront issues for its 4 roots (profiles/r06_experiments), not by these round trips.
    const uint32_t thresh = (uint32_t)g.thresh & 0xffffu;
#pragma unroll
    for (int l = L - 1; l >= 1; --l) {
        const int lc = L - 1 - l, cap = 1 << lc;  // (a power of two: slot arithmetic by shifts, not divisions)
        const int off = cap - 1, offc = 2 * cap - 1;
        const int items = KF_R * cap * 4;
        const bool child0 = l == 1;  // the children are pixels of level 0 (centred gradients), else of a level above (block gradients)
        for (int base = 0; base < items; base += 64 * VORS_KF_BATCH) {
            uint32_t pxy_b[VORS_KF_BATCH];
            KfGradRaw raw_b[VORS_KF_BATCH];
#pragma unroll
            for (int b = 0; b < VORS_KF_BATCH; ++b) {
                const int t = base + 64 * b + lane;
                const bool in = t < items;
                const int rl = in ? (t >> 2) >> lc : 0;  // root slot inside the wavefront
                const int k = (t >> 2) & (cap - 1), c = lane & 3;
                pxy_b[b] = in ? xy[rl * NODES + off + k] : VORS_INVALID_XY;
                const bool pvalid = pxy_b[b] != VORS_INVALID_XY;
                const int cx = pvalid ? 2 * (int)(pxy_b[b] & 0xffffu) + (c >> 1) : 0, cy = pvalid ? 2 * (int)(pxy_b[b] >> 16) + (c & 1) : 0;
                // (an item without a parent requests pixel (0, 0): a safe address, its result is replaced by zeros below)
                if (base + 64 * b < items) raw_b[b] = child0 ? kf_grad_request<true>(g, kf0, kfu, pair, l - 1, cx, cy) : kf_grad_request<false>(g, kf0, kfu, pair, l - 1, cx, cy);
            }
#pragma unroll
            for (int b = 0; b < VORS_KF_BATCH; ++b) {
                if (base + 64 * b >= items) break;  // (uniform)
                const int t = base + 64 * b + lane;
                const bool in = t < items;
                const int rl = in ? (t >> 2) >> lc : 0;
                const int k = (t >> 2) & (cap - 1), c = lane & 3;
                const uint32_t pxy = pxy_b[b];
                const bool pvalid = pxy != VORS_INVALID_XY;
                const int cx = 2 * (int)(pxy & 0xffffu) + (c >> 1), cy = 2 * (int)(pxy >> 16) + (c & 1);
                int gx = 0, gy = 0, tm = 0;
                if (child0) kf_grad_decode<true>(g, raw_b[b], pvalid ? cx : 0, pvalid ? cy : 0, &gx, &gy, &tm);
                else kf_grad_decode<false>(g, raw_b[b], pvalid ? cx : 0, pvalid ? cy : 0, &gx, &gy, &tm);
                gx = pvalid ? gx : 0;
                gy = pvalid ? gy : 0;
                tm = pvalid ? tm : 0;
                const uint32_t g2 = (uint32_t)(gx * gx + gy * gy) & 0xffffu;  // `as u16` wrap, gradient.rs:39-43
                // The four children of a node sit in one quad. Stable ranking (a later child wins a tie: ties rank by child index) = ranking
                // of the DISTINCT keys (g2 << 2 | child); the quad's keys by DPP quad broadcasts, sorted by a 5-exchange network.
                const uint32_t key = (g2 << 2) | (uint32_t)c;
                uint32_t k0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x00, 0xf, 0xf, true);
                uint32_t k1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x55, 0xf, 0xf, true);
                uint32_t k2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0xaa, 0xf, 0xf, true);
                uint32_t k3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0xff, 0xf, 0xf, true);
                {
                    uint32_t lo, hi;
                    lo = min(k0, k1); hi = max(k0, k1); k0 = lo; k1 = hi;
                    lo = min(k2, k3); hi = max(k2, k3); k2 = lo; k3 = hi;
                    lo = min(k0, k2); hi = max(k0, k2); k0 = lo; k2 = hi;
                    lo = min(k1, k3); hi = max(k1, k3); k1 = lo; k3 = hi;
                    lo = min(k1, k2); hi = max(k1, k2); k1 = lo; k2 = hi;
                }
                const uint32_t second = k2 >> 2, third = k1 >> 2;                 // the values of rank 2 and rank 1
                const bool keep2 = second > ((third + thresh) & 0xffffu);         // u16 wrapping add, coarse_to_fine.rs:85
                if (in && key >= k2) {  // rank 3 (the best: always kept) -> slot 2k, rank 2 (kept if it stands out) -> slot 2k + 1
                    const bool best = key == k3;
                    const int dst = rl * NODES + offc + 2 * k + (best ? 0 : 1);
                    xy[dst] = (pvalid && (best || keep2)) ? ((uint32_t)cx | ((uint32_t)cy << 16)) : VORS_INVALID_XY;
                    gr[dst] = slim_pack_tg(tm, gx, gy);
                }
            }
        }
        kf_wave_sync();
    }

    // ---- level 0: inverse depth from the depth map (inverse_depth.rs:24-29); unknown depth -> not a point
    {
        const int cap = 1 << (L - 1), off = cap - 1;
        for (int t = lane; t < KF_R * cap; t += 64) {
            const int rl = t >> (L - 1), k = t & (cap - 1);
            const uint32_t p = xy[rl * NODES + off + k];
            if (p != VORS_INVALID_XY) {
                const uint16_t dz = (depth + (size_t)pair * g.S0)[(p >> 16) * (unsigned)g.lv[0].cols + (p & 0xffffu)];  // uniform base + 32-bit offset
                if (dz == 0) {
                    xy[rl * NODES + off + k] = VORS_INVALID_XY;
                } else {
                    sd[rl * NODES + off + k] = g.depth_scale / (float)dz;
                    sv[rl * NODES + off + k] = g.idepth_variance;
                }
            }
        }
    }
    kf_wave_sync();
    // ---- bottom-up fusion (strategy_dso_mean, inverse_depth.rs:81-98)
#pragma unroll
    for (int l = 1; l < L; ++l) {
        const int cap = 1 << (L - 1 - l);
        const int off = cap - 1, offc = 2 * cap - 1;
        for (int t = lane; t < KF_R * cap; t += 64) {
            const int rl = t >> (L - 1 - l), k = t & (cap - 1);
            const int c1 = rl * NODES + offc + 2 * k, c2 = c1 + 1, dst = rl * NODES + off + k;
            const bool k1 = xy[c1] != VORS_INVALID_XY, k2 = xy[c2] != VORS_INVALID_XY;
            if (k1 && k2) {
                const float d1 = sd[c1], v1 = sv[c1], d2 = sd[c2], v2 = sv[c2];
                const float sum = v1 + v2;
                sd[dst] = (d1 * v1 + d2 * v2) / sum;
                sv[dst] = sum;
            } else if (k1 || k2) {
                const int src = k1 ? c1 : c2;
                sd[dst] = sd[src];
                sv[dst] = sv[src];
            } else {
                xy[dst] = VORS_INVALID_XY;
            }
        }
        kf_wave_sync();
    }
    // ---- records (12 bytes each: coordinates, inverse depth, template + integer gradient; SlimRec). The KF_R roots of a wavefront own
    // KF_R * cap contiguous slots of each level in the STAGING grid; usable points are compacted to the front of that region (fixed
    // order: by root, then tree slot) and the region's count is published; compact_regions_kernel then packs the regions of a pair.
    const int region = blockIdx.x * KF_WAVES + wave;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int cap = 1 << (L - 1 - l), off = cap - 1;
        const int n_here = min(KF_R, max(0, n_roots - root0)) * cap;  // slots of this region that exist
        SlimRec* out = rec.stage + (size_t)pair * g.slots_total + g.lv[l].slot_off + (size_t)root0 * cap;
        int filled = 0;  // wavefront-uniform running count of points written
        for (int base = 0; base < n_here; base += 64) {
            const int t = base + lane;
            const int rl = t >> (L - 1 - l), k = t & (cap - 1);
            const bool in = t < n_here;
            const uint32_t p = in ? xy[rl * NODES + off + k] : VORS_INVALID_XY;
            const bool valid = p != VORS_INVALID_XY;
            const unsigned long long m = __ballot(valid);
            const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (valid) out[(unsigned)(filled + before)] = SlimRec{p, sd[rl * NODES + off + k], gr[rl * NODES + off + k]};
            filled += __popcll(m);
        }
        if (lane == 0) rec.region_cnt[((size_t)pair * VORS_MAX_LEVELS + l) * rec.n_regions + region] = filled;
    }
}

// Per-pair compaction of the staged regions: one workgroup per pair, level after level. Exclusive prefix of the region counts in region order
// (tiles of 256 regions: wave shuffles + LDS), then a cooperative copy — deterministic order (region, then position inside the region),
// no atomics. Publishes n_used[pair][level].
__global__ __launch_bounds__(256) void compact_regions_kernel(Geom g, Records rec) {
    __shared__ int s_cnt[256], s_pre[256], s_wave[4];
    __shared__ int s_base;
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Large batches: all levels of a pair in one workgroup, finest (largest) first — a sixth of the workgroups of a (level, pair) grid, whose
    // small levels were nothing but dispatch (24,576 workgroups at 4096 pairs: 0.26 ms). Small batches and the masked launches of the
    // lock-step trackers (a handful of promoted sequences per frame): one workgroup per (level, pair), gridDim.x = L — the six levels of a
    // pair are a chain of dependent barriers and copies, 32 us in one workgroup, and nothing else is running meanwhile.
    const int l_first = gridDim.x > 1 ? (int)blockIdx.x : 0, l_end = gridDim.x > 1 ? l_first + 1 : g.L;
    for (int l = l_first; l < l_end; ++l) {
        const int cap_r = rec.kf_r << (g.L - 1 - l);  // slots per region at this level
        const int* cnt = rec.region_cnt + ((size_t)pair * VORS_MAX_LEVELS + l) * rec.n_regions;
        const size_t lvl0 = (size_t)pair * g.slots_total + g.lv[l].slot_off;
        const SlimRec* src = rec.stage + lvl0;
        SlimRec* dst = rec.S + lvl0;
        __syncthreads();
        if (threadIdx.x == 0) s_base = 0;
        __syncthreads();
        for (int tile = 0; tile < rec.n_regions; tile += 256) {
            const int r = tile + (int)threadIdx.x;
            const int mine = r < rec.n_regions ? cnt[r] : 0;
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            if (lane == 63) s_wave[wave] = incl;
            __syncthreads();
            int pre = s_base + incl - mine;
            for (int w = 0; w < wave; ++w) pre += s_wave[w];
            s_cnt[threadIdx.x] = mine;
            s_pre[threadIdx.x] = pre;
            __syncthreads();
            const int n_tile = min(256, rec.n_regions - tile);
            const int cap_sh = __ffs(cap_r) - 1;  // (cap_r = roots per region << level distance: a power of two — a shift, not a ~30-instruction division per record)
            for (int j = threadIdx.x; j < n_tile * cap_r; j += 256) {
                const int rr = j >> cap_sh, k = j & (cap_r - 1);
                if (k < s_cnt[rr]) dst[s_pre[rr] + k] = src[(size_t)(tile + rr) * cap_r + k];
            }
            __syncthreads();
            if (threadIdx.x == 255) s_base = pre + mine;
            __syncthreads();
        }
        if (threadIdx.x == 0) rec.n_used[(size_t)pair * VORS_MAX_LEVELS + l] = s_base;
    }
}

// ------------------------------------------------------------------------------------------------------------
// keyframe precompute, dense candidates (extension): all-true level-0 mask, any image size. Candidates are the pixels
// themselves (slot = y*cols + x at each level). NOTHING per-point is stored for level 0: the LM kernel recomputes each
// point from the keyframe image and the depth map (lm_kernels.hip, DenseSrc). For levels >= 1 only the fused inverse depth
// (IZ plane, NaN = Unknown) and its weight (V plane, < 0 = Unknown) are stored: 8 B per pixel of levels >= 1.
// Fusion follows inverse_depth.rs:49-66,81-98 with the four children in [a,b,c,d] order.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fuse_dso_mean(const float dv_in[4], const float vv_in[4], float* od, float* ov) {
    // Branch-free: an Unknown child adds +0 to both sums. Every product d * v is > 0 and x + 0 == x exactly, so these ARE the
    // reference's sequential sums over the known children in [a, b, c, d] order — with ONE division per fused pixel instead of one
    // per distinct child count met in the wavefront. A single known child is passed through (inverse_depth.rs:59-65 never divides it).
    float sd = 0.f, sv = 0.f, single = 0.f;
    int n = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const bool known = vv_in[m] >= 0.f;
        const float prod = dv_in[m] * vv_in[m];
        sd += known ? prod : 0.f;
        sv += known ? vv_in[m] : 0.f;
        single = known ? dv_in[m] : single;
        n += known ? 1 : 0;
    }
    const float q = sd / sv;
    *od = n == 0 ? __builtin_nanf("") : (n == 1 ? single : q);
    *ov = n == 0 ? -1.0f : sv;
}
// from_depth (inverse_depth.rs:24-29): scale / depth. FAST: q = scale * rcp(d), one residual correction q' = fma(fma(-q, d, scale),
// rcp(d), q) — 5 instructions instead of the ~12 of an IEEE division — used only after verify_idepth_kernel has found q' equal to the
// IEEE quotient for EVERY depth 1 .. 65535 at this handle's scale (Geom::fast_idepth); otherwise the division.
template <bool FAST>
__device__ __forceinline__ float idepth_of(float scale, uint32_t dz) {
    const float d = (float)dz;
    if (FAST) {
        const float rc = __builtin_amdgcn_rcpf(d);
        const float q = scale * rc;
        return fmaf(fmaf(-q, d, scale), rc, q);
    }
    return scale / d;
}
// Level 0 -> level 1 (every known child weighs `var`): fuse_dso_mean of from_depth(children), without the per-child selects of the
// generic form — an Unknown child enters as d = 0, w = 0, so its product and its terms of all three sums are +0.
template <bool FAST>
__device__ __forceinline__ void fuse_depths(float scale, float var, const uint32_t dz[4], float* od, float* ov, int& n0) {
    float sd = 0.f, sv = 0.f, s1 = 0.f;
    int n = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const bool known = dz[m] != 0;
        const float d = known ? idepth_of<FAST>(scale, dz[m]) : 0.f;
        sd += d * var;
        sv += known ? var : 0.f;
        s1 += d;  // a single known child: d + 0 + 0 + 0 == d
        n += known ? 1 : 0;
    }
    n0 += n;
    const float q = sd / sv;
    *od = n == 0 ? __builtin_nanf("") : (n == 1 ? s1 : q);
    *ov = n == 0 ? -1.0f : sv;
}
__global__ void verify_idepth_kernel(float scale, int* mismatch) {
    const uint32_t dz = blockIdx.x * blockDim.x + threadIdx.x;
    if (dz == 0 || dz > 65535u) return;
    if (idepth_of<true>(scale, dz) != idepth_of<false>(scale, dz)) atomicExch(mismatch, 1);
}
bool verify_fast_idepth(float scale, hipStream_t s) {
    if (!(scale == scale) || !(fabsf(scale) < 1e30f) || !(fabsf(scale) > 1e-20f)) return false;
    int* flag = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&flag), sizeof(int)) != hipSuccess) return false;
    (void)hipMemsetAsync(flag, 0, sizeof(int), s);
    hipLaunchKernelGGL(verify_idepth_kernel, dim3(256), dim3(256), 0, s, scale, flag);
    int h = 1;
    const bool ok = hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(flag);
    return ok && h == 0;
}
// Wave-aggregated integer counters: usable points per (pair, level), published for the LM kernel's statistics.
__device__ __forceinline__ void count_add(int* counter, int n) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0 && n != 0) atomicAdd(counter, n);
}
// level 1 straight from the depth map (from_depth, inverse_depth.rs:24-29, fused with the first halve). Also counts the usable
// pixels of level 0 (non-zero depth; the odd trailing row / column of level 0 has no level-1 parent and is counted by the
// threads of the last level-1 row / column) and of level 1.
template <bool FAST>
__global__ __launch_bounds__(256) void dense_idepth_level1_kernel(Geom g, const uint16_t* __restrict__ depth, Records rec) {
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = g.lv[1].rows, cols = g.lv[1].cols;
    int n0 = 0, n1 = 0;
    if (t < rows * cols) {
        const int y = t / cols, x = t - y * cols;
        const int fr = g.lv[0].rows, fc = g.lv[0].cols;
        const uint16_t* p = depth + (size_t)pair * g.S0 + (size_t)(2 * y) * fc + 2 * x;
        // children a=(2i,2j) b=(2i+1,2j) c=(2i,2j+1) d=(2i+1,2j+1) with i=row, j=col   (multires.rs:80-83)
        const uint32_t dz[4] = {p[0], p[fc], p[1], p[fc + 1]};
        float od, ov;
        fuse_depths<FAST>(g.depth_scale, g.idepth_variance, dz, &od, &ov, n0);
        const bool last_x = x == cols - 1 && (fc & 1), last_y = y == rows - 1 && (fr & 1);
        if (last_x) n0 += (p[2] != 0) + (p[fc + 2] != 0);
        if (last_y) n0 += (p[2 * fc] != 0) + (p[2 * fc + 1] != 0);
        if (last_x && last_y) n0 += p[2 * fc + 2] != 0;
        const size_t slot = (size_t)pair * g.slots_total + g.lv[1].slot_off + t;
        rec.IZ[slot] = od;
        rec.V[slot] = ov;
        n1 = ov >= 0.f;
    }
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 0, n0);
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 1, n1);
}
// Same, 4 level-1 pixels per thread when cols(level 0) % 8 == 0 and the depth rows are 16-byte aligned: two 16-byte loads, two
// 16-byte stores.
template <bool FAST>
__global__ __launch_bounds__(256) void dense_idepth_level1_wide_kernel(Geom g, const uint16_t* __restrict__ depth, Records rec) {
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int t0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int rows = g.lv[1].rows, cols = g.lv[1].cols;
    int n0 = 0, n1 = 0;
    if (t0 < rows * cols) {
        const int y = t0 / cols, x0 = t0 - y * cols;
        const int fr = g.lv[0].rows, fc = g.lv[0].cols;
        const uint16_t* p = depth + (size_t)pair * g.S0 + (size_t)(2 * y) * fc + 2 * x0;
        const uint4 r0 = *reinterpret_cast<const uint4*>(p), r1 = *reinterpret_cast<const uint4*>(p + fc);
        const uint32_t w0[4] = {r0.x, r0.y, r0.z, r0.w}, w1[4] = {r1.x, r1.y, r1.z, r1.w};
        float od[4], ov[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t dz[4] = {w0[k] & 0xffffu, w1[k] & 0xffffu, w0[k] >> 16, w1[k] >> 16};  // a, b, c, d
            fuse_depths<FAST>(g.depth_scale, g.idepth_variance, dz, &od[k], &ov[k], n0);
            n1 += ov[k] >= 0.f;
        }
        if (y == rows - 1 && (fr & 1)) {  // odd trailing row of level 0 (no odd column: fc % 8 == 0)
            const uint4 r2 = *reinterpret_cast<const uint4*>(p + 2 * fc);
            const uint32_t w2[4] = {r2.x, r2.y, r2.z, r2.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) n0 += ((w2[k] & 0xffffu) != 0) + ((w2[k] >> 16) != 0);
        }
        const size_t slot = (size_t)pair * g.slots_total + g.lv[1].slot_off + t0;
        *reinterpret_cast<float4*>(rec.IZ + slot) = make_float4(od[0], od[1], od[2], od[3]);
        *reinterpret_cast<float4*>(rec.V + slot) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 0, n0);
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 1, n1);
}
// Levels 1 AND 2 in one pass when rows(level 0) % 4 == 0 and cols(level 0) % 16 == 0 (both halvings exact, 16-byte aligned depth
// rows): one thread per level-2 pixel reads its 4x4 depth block (four 8-byte loads), fuses the four level-1 pixels (stored:
// inverse depth only — the level-1 weights are consumed right here and never written) and then the level-2 pixel. Halves the
// traffic of the two largest inverse-depth passes.
template <bool FAST>
__global__ __launch_bounds__(256) void dense_idepth_level12_kernel(Geom g, const uint16_t* __restrict__ depth, Records rec) {
    // TWO horizontally adjacent level-2 pixels per thread: 16-byte loads of the four depth rows, 16-byte stores of the two level-1 rows
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int rows2 = g.lv[2].rows, cols2 = g.lv[2].cols;  // cols2 is a multiple of 4 here
    int n0 = 0, n1 = 0, n2 = 0;
    if (t < rows2 * cols2) {
        const int y2 = t / cols2, x2 = t - y2 * cols2;
        const int fc = g.lv[0].cols, c1 = g.lv[1].cols;
        const uint16_t* p = depth + (size_t)pair * g.S0 + (size_t)(4 * y2) * fc + 4 * x2;
        uint32_t w[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + (size_t)r * fc);
            w[r][0] = v.x;
            w[r][1] = v.y;
            w[r][2] = v.z;
            w[r][3] = v.w;
        }
        float od1[2][4], ov1[2][4];  // [row][col] of the 2x4 level-1 block
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // children a=(2i,2j) b=(2i+1,2j) c=(2i,2j+1) d=(2i+1,2j+1)   (multires.rs:80-83)
                const uint32_t top = w[2 * i][j], bot = w[2 * i + 1][j];
                const uint32_t dz[4] = {top & 0xffffu, bot & 0xffffu, top >> 16, bot >> 16};
                fuse_depths<FAST>(g.depth_scale, g.idepth_variance, dz, &od1[i][j], &ov1[i][j], n0);
                n1 += ov1[i][j] >= 0.f;
            }
        const size_t s1 = (size_t)pair * g.slots_total + g.lv[1].slot_off + (size_t)(2 * y2) * c1 + 2 * x2;
        *reinterpret_cast<float4*>(rec.IZ + s1) = make_float4(od1[0][0], od1[0][1], od1[0][2], od1[0][3]);
        *reinterpret_cast<float4*>(rec.IZ + s1 + c1) = make_float4(od1[1][0], od1[1][1], od1[1][2], od1[1][3]);
        float od[2], ov[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float dv2[4] = {od1[0][2 * k], od1[1][2 * k], od1[0][2 * k + 1], od1[1][2 * k + 1]};
            const float vv2[4] = {ov1[0][2 * k], ov1[1][2 * k], ov1[0][2 * k + 1], ov1[1][2 * k + 1]};
            fuse_dso_mean(dv2, vv2, &od[k], &ov[k]);
            n2 += ov[k] >= 0.f;
        }
        const size_t s2 = (size_t)pair * g.slots_total + g.lv[2].slot_off + t;
        *reinterpret_cast<float2*>(rec.IZ + s2) = make_float2(od[0], od[1]);
        *reinterpret_cast<float2*>(rec.V + s2) = make_float2(ov[0], ov[1]);
    }
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 0, n0);
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 1, n1);
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 2, n2);
}
// Levels 1, 2 AND 3 in one pass when rows(level 0) % 8 == 0 and cols(level 0) % 8 == 0: one thread per level-3 pixel takes its 8x8
// depth block as two 4x8 halves (the body of the kernel above) and fuses their four level-2 pixels — level 2 is written once and not
// read back by a halving launch (0.63 GB per 4096 pairs at 640x480).
template <bool FAST>
__global__ __launch_bounds__(256) void dense_idepth_level123_kernel(Geom g, const uint16_t* __restrict__ depth, Records rec) {
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int cols3 = g.lv[3].cols;
    int n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    if (t < g.lv[3].rows * cols3) {
        const int y3 = t / cols3, x3 = t - y3 * cols3;
        const int fc = g.lv[0].cols, c1 = g.lv[1].cols, c2 = g.lv[2].cols;
        const size_t base = (size_t)pair * g.slots_total;
        float od2[2][2], ov2[2][2];  // [row][col] of the 2x2 level-2 block
        const uint16_t* p = depth + (size_t)pair * g.S0 + (size_t)(8 * y3) * fc + 8 * x3;
        uint4 v8[8];  // all eight rows requested before any is used
#pragma unroll
        for (int r = 0; r < 8; ++r) v8[r] = *reinterpret_cast<const uint4*>(p + (size_t)r * fc);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t w[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint4 v = v8[4 * h + r];
                w[r][0] = v.x;
                w[r][1] = v.y;
                w[r][2] = v.z;
                w[r][3] = v.w;
            }
            float od1[2][4], ov1[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // children a=(2i,2j) b=(2i+1,2j) c=(2i,2j+1) d=(2i+1,2j+1)   (multires.rs:80-83)
                    const uint32_t top = w[2 * i][j], bot = w[2 * i + 1][j];
                    const uint32_t dz[4] = {top & 0xffffu, bot & 0xffffu, top >> 16, bot >> 16};
                    fuse_depths<FAST>(g.depth_scale, g.idepth_variance, dz, &od1[i][j], &ov1[i][j], n0);
                    n1 += ov1[i][j] >= 0.f;
                }
            const size_t s1 = base + g.lv[1].slot_off + (size_t)(4 * y3 + 2 * h) * c1 + 4 * x3;
            *reinterpret_cast<float4*>(rec.IZ + s1) = make_float4(od1[0][0], od1[0][1], od1[0][2], od1[0][3]);
            *reinterpret_cast<float4*>(rec.IZ + s1 + c1) = make_float4(od1[1][0], od1[1][1], od1[1][2], od1[1][3]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float dv[4] = {od1[0][2 * k], od1[1][2 * k], od1[0][2 * k + 1], od1[1][2 * k + 1]};
                const float vv[4] = {ov1[0][2 * k], ov1[1][2 * k], ov1[0][2 * k + 1], ov1[1][2 * k + 1]};
                fuse_dso_mean(dv, vv, &od2[h][k], &ov2[h][k]);
                n2 += ov2[h][k] >= 0.f;
            }
            const size_t s2 = base + g.lv[2].slot_off + (size_t)(2 * y3 + h) * c2 + 2 * x3;
            *reinterpret_cast<float2*>(rec.IZ + s2) = make_float2(od2[h][0], od2[h][1]);
            *reinterpret_cast<float2*>(rec.V + s2) = make_float2(ov2[h][0], ov2[h][1]);
        }
        const float dv[4] = {od2[0][0], od2[1][0], od2[0][1], od2[1][1]};
        const float vv[4] = {ov2[0][0], ov2[1][0], ov2[0][1], ov2[1][1]};
        float od, ov;
        fuse_dso_mean(dv, vv, &od, &ov);
        n3 = ov >= 0.f;
        const size_t s3 = base + g.lv[3].slot_off + t;
        rec.IZ[s3] = od;
        rec.V[s3] = ov;
    }
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 0, n0);
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 1, n1);
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 2, n2);
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + 3, n3);
}
__global__ __launch_bounds__(256) void dense_idepth_halve_kernel(Geom g, int l, Records rec) {
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols;
    int n = 0;
    if (t < rows * cols) {
        const int y = t / cols, x = t - y * cols;
        const int fc = g.lv[l - 1].cols;
        const size_t cb = (size_t)pair * g.slots_total + g.lv[l - 1].slot_off + (size_t)(2 * y) * fc + 2 * x;
        const size_t idx[4] = {cb, cb + fc, cb + 1, cb + fc + 1};
        float dv[4], vv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            dv[m] = rec.IZ[idx[m]];
            vv[m] = rec.V[idx[m]];
        }
        float od, ov;
        fuse_dso_mean(dv, vv, &od, &ov);
        const size_t slot = (size_t)pair * g.slots_total + g.lv[l].slot_off + t;
        rec.IZ[slot] = od;
        rec.V[slot] = ov;
        n = ov >= 0.f;
    }
    count_add(rec.n_used + (size_t)pair * VORS_MAX_LEVELS + l, n);
}
// Inspection only (vors_batch_get_points in dense mode): materialise the records of ONE level of ONE pair into `out`
// (planes of n_slots entries, index = y*cols + x), with exactly the arithmetic the LM kernel uses on the fly.
__global__ __launch_bounds__(256) void dense_materialize_kernel(Geom g, int l, int pair, const uint8_t* __restrict__ kf0,
                                                                 const uint8_t* __restrict__ kfu, const uint16_t* __restrict__ depth,
                                                                 Records rec, Records out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols;
    if (t >= rows * cols) return;
    const int y = t / cols, x = t - y * cols;
    float iz;
    bool valid;
    if (l == 0) {
        const uint16_t dz = depth[(size_t)pair * g.S0 + t];
        valid = dz != 0;
        iz = g.depth_scale / (float)dz;
    } else {
        iz = rec.IZ[(size_t)pair * g.slots_total + g.lv[l].slot_off + t];
        valid = !(iz != iz);
    }
    if (valid) {
        int gx, gy;
        grad_at(g, kf0, kfu, pair, l, x, y, &gx, &gy);
        write_record(out, t, g.lv[l].k, x, y, iz, gx, gy, level_ptr(g, kf0, kfu, pair, l)[t]);
    } else {
        write_empty(out, t);
    }
}
__global__ __launch_bounds__(256) void slim_materialize_kernel(Geom g, int l, int pair, Records rec, int n, Records out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SlimRec r = rec.S[(size_t)pair * g.slots_total + g.lv[l].slot_off + i];
    write_record(out, i, g.lv[l].k, (int)(r.xy & 0xffffu), (int)(r.xy >> 16), r.iz, slim_gx(r.tg), slim_gy(r.tg), (uint8_t)(r.tg & 0xff));
}
void launch_slim_materialize(const Geom& g, int l, int pair, Records rec, int n, Records out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(slim_materialize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, g, l, pair, rec, n, out);
}
void launch_dense_materialize(const Geom& g, int l, int pair, Pyramid kf, const uint16_t* depth, Records rec, Records out,
                              hipStream_t s) {
    hipLaunchKernelGGL(dense_materialize_kernel, dim3((g.lv[l].n_slots + 255) / 256), dim3(256), 0, s, g, l, pair, kf.level0, kf.upper,
                       depth, rec, out);
}

// Zero `stride` ints per pair of a per-pair counter array — all n_pairs pairs, or (masked launch, Geom::sel_list) the selected ones.
__global__ __launch_bounds__(64) void zero_ints_kernel(Geom g, int* __restrict__ base, int stride) {
    const int pair = select_pair(g, blockIdx.x);
    if (pair < 0) return;
    for (int i = threadIdx.x; i < stride; i += 64) base[(size_t)pair * stride + i] = 0;
}
void launch_zero_ints(const Geom& g, int* base, int stride, int n_pairs, hipStream_t s) {
    if (!g.sel_list) (void)hipMemsetAsync(base, 0, (size_t)n_pairs * stride * sizeof(int), s);
    else hipLaunchKernelGGL(zero_ints_kernel, dim3(n_pairs), dim3(64), 0, s, g, base, stride);
}

// Region geometry of the coarse-to-fine keyframe kernel (read by capi.cpp when it sizes the handle): roots per wavefront, and
// wavefront regions per pair.
static int keyframe_roots_per_wave(const Geom& g) {
    static int kf_r = getenv("VORS_KF_R") ? atoi(getenv("VORS_KF_R")) : 4;  // roots per wavefront (tuning knob)
    int r = kf_r >= 8 ? 8 : (kf_r >= 4 ? 4 : (kf_r >= 2 ? 2 : 1));
    while (r > 1 && (size_t)KF_WAVES * r * kf_nodes(g.L) * 16 > 64 * 1024) r >>= 1;  // stay inside the 64 KiB a workgroup may ask for
    return r;
}
void keyframe_region_geometry(const Geom& g, int* kf_r, int* n_regions) {
    const int r = keyframe_roots_per_wave(g), n_roots = g.root_rows * g.root_cols;
    *kf_r = r;
    *n_regions = ((n_roots + KF_WAVES * r - 1) / (KF_WAVES * r)) * KF_WAVES;
}
void launch_keyframe(const Geom& g, Pyramid kf, const uint16_t* depth, Records rec, int n_pairs, hipStream_t s) {
    if (g.mode == VORS_CANDIDATES_DENSE) {
        launch_zero_ints(g, rec.n_used, VORS_MAX_LEVELS, n_pairs, s);
        int next = 2;
        if (g.L >= 2) {
            // (slots_total and slot_off are multiples of 4, so the vector stores of the wide kernels are aligned)
            const bool aligned = reinterpret_cast<uintptr_t>(depth) % 16 == 0;
            if (g.L >= 4 && aligned && g.lv[0].rows % 8 == 0 && g.lv[0].cols % 8 == 0 && !getenv("VORS_IDEPTH_LEVEL12")) {
                const dim3 grid((g.lv[3].n_slots + 255) / 256, n_pairs);
                if (g.fast_idepth) hipLaunchKernelGGL(dense_idepth_level123_kernel<true>, grid, dim3(256), 0, s, g, depth, rec);
                else hipLaunchKernelGGL(dense_idepth_level123_kernel<false>, grid, dim3(256), 0, s, g, depth, rec);
                next = 4;
            } else if (g.L >= 3 && aligned && g.lv[0].rows % 4 == 0 && g.lv[0].cols % 16 == 0) {
                if (g.fast_idepth) hipLaunchKernelGGL(dense_idepth_level12_kernel<true>, dim3((g.lv[2].n_slots / 2 + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, rec);
                else hipLaunchKernelGGL(dense_idepth_level12_kernel<false>, dim3((g.lv[2].n_slots / 2 + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, rec);
                next = 3;
            } else if (aligned && g.lv[0].cols % 8 == 0) {
                if (g.fast_idepth) hipLaunchKernelGGL(dense_idepth_level1_wide_kernel<true>, dim3((g.lv[1].n_slots / 4 + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, rec);
                else hipLaunchKernelGGL(dense_idepth_level1_wide_kernel<false>, dim3((g.lv[1].n_slots / 4 + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, rec);
            } else {
                if (g.fast_idepth) hipLaunchKernelGGL(dense_idepth_level1_kernel<true>, dim3((g.lv[1].n_slots + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, rec);
                else hipLaunchKernelGGL(dense_idepth_level1_kernel<false>, dim3((g.lv[1].n_slots + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, rec);
            }
        }
        for (int l = next; l < g.L; ++l)
            hipLaunchKernelGGL(dense_idepth_halve_kernel, dim3((g.lv[l].n_slots + 255) / 256, n_pairs), dim3(256), 0, s, g, l, rec);
    } else {
        // the region geometry the handle was sized for (keyframe_region_geometry at create(), capi.cpp) is the single source of truth
        const int r = rec.kf_r;
        dim3 grid(rec.n_regions / KF_WAVES, n_pairs);
        const size_t lds = (size_t)KF_WAVES * r * kf_nodes(g.L) * 16;
#ifndef VORS_KF_CONST_L
#define VORS_KF_CONST_L 1  // (0: every launch reads the level count from the geometry — A/B)
#endif
        if (r == 8 && g.L == 6 && VORS_KF_CONST_L) hipLaunchKernelGGL((keyframe_sparse_kernel<8, 6>), grid, dim3(64 * KF_WAVES), lds, s, g, kf.level0, kf.upper, depth, rec);
        else if (r == 2 && g.L == 6 && VORS_KF_CONST_L) hipLaunchKernelGGL((keyframe_sparse_kernel<2, 6>), grid, dim3(64 * KF_WAVES), lds, s, g, kf.level0, kf.upper, depth, rec);
        else if (r == 8) hipLaunchKernelGGL((keyframe_sparse_kernel<8, 0>), grid, dim3(64 * KF_WAVES), lds, s, g, kf.level0, kf.upper, depth, rec);
        else if (r == 4 && g.L == 6 && VORS_KF_CONST_L) hipLaunchKernelGGL((keyframe_sparse_kernel<4, 6>), grid, dim3(64 * KF_WAVES), lds, s, g, kf.level0, kf.upper, depth, rec);
        else if (r == 4 && g.L == 7 && VORS_KF_CONST_L) hipLaunchKernelGGL((keyframe_sparse_kernel<4, 7>), grid, dim3(64 * KF_WAVES), lds, s, g, kf.level0, kf.upper, depth, rec);
        else if (r == 4) hipLaunchKernelGGL((keyframe_sparse_kernel<4, 0>), grid, dim3(64 * KF_WAVES), lds, s, g, kf.level0, kf.upper, depth, rec);
        else if (r == 2) hipLaunchKernelGGL((keyframe_sparse_kernel<2, 0>), grid, dim3(64 * KF_WAVES), lds, s, g, kf.level0, kf.upper, depth, rec);
        else hipLaunchKernelGGL((keyframe_sparse_kernel<1, 0>), grid, dim3(64 * KF_WAVES), lds, s, g, kf.level0, kf.upper, depth, rec);
        // (REFERENCE arithmetic: launch_sort_colmajor, which every caller runs next, packs AND orders the regions in one pass — lm_reference.hip)
        if (!ref_rank_from_regions(g, rec))
            hipLaunchKernelGGL(compact_regions_kernel, dim3(n_pairs <= 1024 ? g.L : 1, n_pairs), dim3(256), 0, s, g, rec);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Exactness proof-by-enumeration for div_uniform (lie.h) and the depth lookup table of the dense LM source.
// ------------------------------------------------------------------------------------------------------------
__global__ void verify_fastdiv_kernel(float d, float r, int* mismatch) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;  // 2^23 significands
    if (m >= (1u << 23)) return;
    const FastDiv f{d, r, 1};
    // one binade with a large and one with a small exponent (both scale exactly; two are checked for good measure)
    const float x1 = __int_as_float((int)(0x3f800000u | m)), x2 = __int_as_float((int)(0x4b000000u | m));
    const bool bad = (div_uniform<true>(x1, f) != x1 / d) || (div_uniform<true>(-x1, f) != (-x1) / d) || (div_uniform<true>(x2, f) != x2 / d);
    if (bad) atomicExch(mismatch, 1);
}
bool verify_fastdiv(float d, float r, hipStream_t s) {
    if (!(d == d) || d == 0.0f || !(r == r) || fabsf(d) > 1e30f || fabsf(d) < 1e-30f) return false;
    int* flag = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&flag), sizeof(int)) != hipSuccess) return false;
    (void)hipMemsetAsync(flag, 0, sizeof(int), s);
    hipLaunchKernelGGL(verify_fastdiv_kernel, dim3((1u << 23) / 256), dim3(256), 0, s, d, r, flag);
    int h = 1;
    const bool ok = hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(flag);
    return ok && h == 0;
}
__global__ void build_depth_lut_kernel(float depth_scale, float2* lut) {
    const int dz = blockIdx.x * blockDim.x + threadIdx.x;
    if (dz >= 65536) return;
    const float iz = depth_scale / (float)dz;  // inverse_depth.rs:24-29 (dz = 0 is never used: the pixel is Unknown)
    lut[dz] = make_float2(iz, 1.0f / iz);      // lm_optimizer.rs:215: back_project(.., 1.0 / _z)
}
void launch_build_depth_lut(float depth_scale, float2* lut, hipStream_t s) {
    hipLaunchKernelGGL(build_depth_lut_kernel, dim3(256), dim3(256), 0, s, depth_scale, lut);
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

// Frames of the same scene family at explicit twists (sequence tooling): frame f is rendered with seed[f], salt[f] at exp(xi[f]).
struct SynthFrame {
    uint64_t seed, salt;
    double xi[6];
};
__global__ void synth_frames_kernel(const SynthFrame* __restrict__ frames, int rows, int cols, vors_synth::CameraD cam, int invalid_percent,
                                    uint8_t* __restrict__ gray, uint16_t* __restrict__ depth) {
    const int f = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * cols) return;
    const SynthFrame fr = frames[f];
    const vors_synth::RigidD m = vors_synth::se3_exp_d(fr.xi);
    const int y = t / cols, x = t - y * cols;
    const size_t o = (size_t)f * rows * cols + t;
    uint8_t gq;
    uint16_t dq;
    vors_synth::render_pixel(fr.seed, fr.salt, cam, m, x, y, invalid_percent, &gq, &dq);
    gray[o] = gq;
    depth[o] = dq;
}
void launch_synth_frames(const void* d_frames, int n_frames, int rows, int cols, const double cam5[5], int invalid_percent, uint8_t* gray,
                         uint16_t* depth, hipStream_t s) {
    vors_synth::CameraD cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    dim3 grid((rows * cols + 255) / 256, n_frames);
    hipLaunchKernelGGL(synth_frames_kernel, grid, dim3(256), 0, s, static_cast<const SynthFrame*>(d_frames), rows, cols, cam, invalid_percent, gray,
                       depth);
}

// ------------------------------------------------------------------------------------------------------------
// N sequences in lock-step (vors_trackers_*): the tracker state machine of inverse_compositional.rs:203-239 on the device.
// ------------------------------------------------------------------------------------------------------------
// After the LM stage of a frame: current_frame_pose <- the tracked pose (the LM kernel already wrote the previous pose for a sequence
// whose optimizer failed, :206-208); a sequence whose optical flow reached the threshold (:224) takes the current frame as its keyframe:
// keyframe_pose <- current_frame_pose (:238), and it is appended to the promotion list that the masked keyframe launches read.
// One workgroup; the list comes out in sequence order (wave ballots + a running base), so every launch that follows is deterministic.
// The frame index lives on the device (`frame_counter`, incremented here) so that the launch carries no per-frame argument: the
// single-sequence tracker replays its per-frame launch sequence from a HIP graph.
__global__ __launch_bounds__(256) void trackers_advance_kernel(int n_seq, int* __restrict__ frame_counter, const float* __restrict__ out_poses7,
                                                               const vors_pair_stats* __restrict__ stats, float* __restrict__ cur_poses7,
                                                               float* __restrict__ kf_poses7, int32_t* __restrict__ kf_frame,
                                                               int* __restrict__ promo_list, int* __restrict__ promo_count) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int frame_index = *frame_counter + 1;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    if (threadIdx.x == 0) *frame_counter = frame_index;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s0 = 0; s0 < n_seq; s0 += 256) {
        const int s = s0 + (int)threadIdx.x;
        bool promote = false;
        if (s < n_seq) {
            promote = stats[s].change_keyframe != 0;
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const float v = out_poses7[7 * s + q];
                cur_poses7[7 * s + q] = v;
                if (promote) kf_poses7[7 * s + q] = v;
            }
            if (promote) kf_frame[s] = frame_index;
        }
        const unsigned long long m = __ballot(promote);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (promote) promo_list[before + __popcll(m & ((1ull << lane) - 1ull))] = s;
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) *promo_count = s_base;
}
void launch_trackers_advance(int n_seq, int* frame_counter, const float* out_poses7, const vors_pair_stats* stats, float* cur_poses7,
                             float* kf_poses7, int32_t* kf_frame, int* promo_list, int* promo_count, hipStream_t s) {
    hipLaunchKernelGGL(trackers_advance_kernel, dim3(1), dim3(256), 0, s, n_seq, frame_counter, out_poses7, stats, cur_poses7, kf_poses7, kf_frame,
                       promo_list, promo_count);
}
// Results of sequence 0 of a lock-step handle as ONE packed record (pose7, status, keyframe index, vors_pair_stats; capi.cpp TrackerOut),
// stored by the device straight into the caller's pinned host memory: the single-sequence tracker reads its frame back without a copy call.
__global__ __launch_bounds__(64) void tracker_pack_out_kernel(const float* __restrict__ pose7, const int32_t* __restrict__ status,
                                                              const int32_t* __restrict__ kf_frame, const vors_pair_stats* __restrict__ stats,
                                                              uint32_t* __restrict__ out) {
    const int t = threadIdx.x;
    constexpr int NS = (int)(sizeof(vors_pair_stats) / 4);
    if (t < 7) out[t] = __float_as_uint(pose7[t]);
    if (t == 7) out[7] = (uint32_t)status[0];
    if (t == 8) out[8] = (uint32_t)kf_frame[0];
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(stats);
    for (int i = t; i < NS; i += 64) out[9 + i] = sp[i];
}
void launch_tracker_pack_out(const float* pose7, const int32_t* status, const int32_t* kf_frame, const vors_pair_stats* stats, void* out,
                             hipStream_t s) {
    hipLaunchKernelGGL(tracker_pack_out_kernel, dim3(1), dim3(64), 0, s, pose7, status, kf_frame, stats, static_cast<uint32_t*>(out));
}
// Dense mode keeps its keyframes (level 0, depth map, upper pyramid levels) in the handle: the promoted sequences copy theirs from the
// current frame (the reference MOVES the current pyramid into the keyframe, inverse_compositional.rs:230-235). 16 bytes per thread.
__global__ __launch_bounds__(256) void promote_copy_kernel(Geom g, const uint8_t* __restrict__ src, size_t src_stride, uint8_t* __restrict__ dst,
                                                           size_t dst_stride, size_t bytes, int vec) {
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const uint8_t* sp = src + (size_t)pair * src_stride;
    uint8_t* dp = dst + (size_t)pair * dst_stride;
    const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (t >= bytes) return;
    if (vec && t + 16 <= bytes) {
        *reinterpret_cast<uint4*>(dp + t) = *reinterpret_cast<const uint4*>(sp + t);
    } else {
        for (size_t k = t; k < bytes && k < t + 16; ++k) dp[k] = sp[k];
    }
}
__global__ void identity_poses_kernel(float* __restrict__ a, float* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 7 * n) {
        const float v = (i % 7 == 6) ? 1.0f : 0.0f;  // translation 0, quaternion (0, 0, 0, 1)
        a[i] = v;
        b[i] = v;
    }
}
void launch_identity_poses(float* a, float* b, int n, hipStream_t s) {
    hipLaunchKernelGGL(identity_poses_kernel, dim3((7 * n + 255) / 256), dim3(256), 0, s, a, b, n);
}
void launch_promote_copy(const Geom& g, const void* src, size_t src_stride, void* dst, size_t dst_stride, size_t bytes, int n_pairs,
                         hipStream_t s) {
    const int vec = (((uintptr_t)src | (uintptr_t)dst | src_stride | dst_stride) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(promote_copy_kernel, dim3((unsigned)((bytes + 4095) / 4096), n_pairs), dim3(256), 0, s, g, static_cast<const uint8_t*>(src),
                       src_stride, static_cast<uint8_t*>(dst), dst_stride, bytes, vec);
}

}  // namespace vors
