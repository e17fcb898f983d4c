// DSO-style candidate selection on the device (candidates_mode = 2) and the GENERIC-MASK keyframe path that turns an
// arbitrary level-0 candidate mask into the record planes the LM kernel consumes.
//
// Reference: src/core/candidates/dso.rs (select :98-147, pick_all_block_candidates :156-189, init_max_gradients :192-222,
// max_of_four_gradients :225-241, pick_level_block_candidates :248-276, region_thresholds :284-303,
// region_median_gradients :307-325) with the parameters of examples/candidates_dso.rs:40-59 (gradient magnitude =
// sqrt(gradient::squared_norm_direct) as u16, region 32 / (1.0, 3), blocks 4 / 3 levels / 0.5, recursion 2 / 0.8 / 4.0 / 1.1,
// target 2000). The reference's Tracker never calls it (it hard-wires coarse_to_fine); here it is an alternative level-0 mask
// source for BASELINE config 3. Deviation: the random sub-sampling branch (dso.rs:140-143) uses an unseeded thread_rng in the
// reference; a counter-based hash of (seed, row, col) stands in (same rule in the oracle), so runs are repeatable.
// Everything per pair is decided on the device (no host round trip): the up-to-3 rounds are launched unconditionally and
// pairs that have finished skip them.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "engine.h"

namespace vors {

#define DSO_REGION 32
#define DSO_LEVELS 3
#define DSO_TARGET 2000
#define DSO_SEED 0x5EEDD50ull

__device__ __forceinline__ uint64_t dso_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// gradient magnitude: (sqrt(((gx^2 + gy^2) / 4) as u16 as f32)) as u16, border 0  (gradient.rs:49-65, candidates_dso.rs:42)
__global__ __launch_bounds__(256) void dso_gradmag_kernel(Geom g, const uint8_t* __restrict__ kf0, DsoWs ws) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    if (t >= rows * cols) return;
    const int y = t / cols, x = t - y * cols;
    uint8_t out = 0;
    if (x > 0 && y > 0 && x < cols - 1 && y < rows - 1) {
        const uint8_t* p = kf0 + (size_t)pair * g.S0 + t;
        const int gx = (int)p[1] - (int)p[-1], gy = (int)p[cols] - (int)p[-cols];
        const int g2 = (gx * gx + gy * gy) / 4;
        out = (uint8_t)(int)sqrtf((float)g2);  // <= 180
    }
    ws.gmag[(size_t)pair * g.S0 + t] = out;
}
// region medians: sorted[len / 2] through a 256-bin histogram (dso.rs:307-325). One workgroup per region.
__global__ __launch_bounds__(256) void dso_region_median_kernel(Geom g, DsoWs ws) {
    __shared__ int hist[256];
    const int pair = blockIdx.y;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    const int rc = (cols + DSO_REGION - 1) / DSO_REGION;
    const int ri = blockIdx.x / rc, rj = blockIdx.x - ri * rc;
    const int h = min(DSO_REGION, rows - ri * DSO_REGION), w = min(DSO_REGION, cols - rj * DSO_REGION);
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t* gm = ws.gmag + (size_t)pair * g.S0;
    for (int t = threadIdx.x; t < h * w; t += 256) {
        const int ly = t / w, lx = t - ly * w;
        atomicAdd(&hist[gm[(size_t)(ri * DSO_REGION + ly) * cols + rj * DSO_REGION + lx]], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int k = (h * w) / 2;
        int acc = 0, med = 0;
        for (int b = 0; b < 256; ++b) {
            acc += hist[b];
            if (acc > k) {
                med = b;
                break;
            }
        }
        ws.median[(size_t)pair * ws.n_regions + blockIdx.x] = (uint16_t)med;
    }
}
// region thresholds: a * (mean3x3(median) + b)^2 cast to u16 (dso.rs:284-303), a = 1.0, b = 3
__global__ void dso_region_threshold_kernel(Geom g, DsoWs ws) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ws.n_regions) return;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    const int rr = (rows + DSO_REGION - 1) / DSO_REGION, rc = (cols + DSO_REGION - 1) / DSO_REGION;
    const int i = t / rc, j = t - i * rc;
    const uint16_t* med = ws.median + (size_t)pair * ws.n_regions;
    uint16_t sum = 0;
    int n = 0;
    for (int jj = max(0, j - 1); jj < min(rc, j + 2); ++jj)
        for (int ii = max(0, i - 1); ii < min(rr, i + 2); ++ii) {
            sum = (uint16_t)(sum + med[ii * rc + jj]);
            ++n;
        }
    const float tt = (float)sum / (float)n + 3.0f;
    ws.thresh[(size_t)pair * ws.n_regions + t] = (uint16_t)(1.0f * tt * tt);
}
// per-pair round state
__global__ void dso_init_state_kernel(DsoWs ws, int n_pairs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    DsoState st;
    st.base_size = 4;
    st.iterations_left = 2;
    st.done = 0;
    st.random_keep = -1;
    st.count = 0;
    ws.state[p] = st;
}
// Level-0 block maxima at the pair's current base size (dso.rs:192-222): first maximum in column-major order.
__global__ __launch_bounds__(256) void dso_blockmax_kernel(Geom g, DsoWs ws) {
    const int pair = blockIdx.y;
    const DsoState st = ws.state[pair];
    if (st.done) return;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols, bs = st.base_size;
    const int br = (rows + bs - 1) / bs, bc = (cols + bs - 1) / bs;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // also reset the level masks / picked plane of this round (grid covers S0 threads)
    if (t < g.S0) {
        ws.picked[(size_t)pair * g.S0 + t] = 0;
        ws.mask1[(size_t)pair * ws.mask_stride + t] = 1;  // mask of the NEXT levels, all true (dso.rs:259)
    }
    if (t == 0) ws.state[pair].count = 0;
    if (t >= br * bc) return;
    const int bi = t / bc, bj = t - bi * bc;
    const int si = bi * bs, sj = bj * bs, ei = min(si + bs, rows), ej = min(sj + bs, cols);
    const uint8_t* gm = ws.gmag + (size_t)pair * g.S0;
    int mg = gm[(size_t)si * cols + sj], mi = si, mj = sj;
    for (int j = sj; j < ej; ++j)
        for (int i = si; i < ei; ++i) {
            const int v = gm[(size_t)i * cols + j];
            if (v > mg) {
                mg = v;
                mi = i;
                mj = j;
            }
        }
    ws.max_g[(size_t)pair * ws.max_stride + t] = (uint8_t)mg;
    ws.max_pos[(size_t)pair * ws.max_stride + t] = (uint32_t)(mi * cols + mj);
}
// Block maxima of level l from level l-1 (halve with max_of_four_gradients, dso.rs:225-241: ties keep the earlier of a,b,c,d)
__global__ __launch_bounds__(256) void dso_halve_max_kernel(Geom g, DsoWs ws, int l) {
    const int pair = blockIdx.y;
    const DsoState st = ws.state[pair];
    if (st.done) return;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols, bs = st.base_size;
    int r = (rows + bs - 1) / bs, c = (cols + bs - 1) / bs, off = 0;
    for (int k = 1; k < l; ++k) {
        off += r * c;
        r /= 2;
        c /= 2;
    }
    const int pr = r, pc = c, poff = off;  // level l-1
    off += r * c;
    r /= 2;
    c /= 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0 || c == 0 || t >= r * c) return;
    const int i = t / c, j = t - i * c;
    const uint8_t* pg = ws.max_g + (size_t)pair * ws.max_stride + poff;
    const uint32_t* pp = ws.max_pos + (size_t)pair * ws.max_stride + poff;
    const int idx[4] = {(2 * i) * pc + 2 * j, (2 * i + 1) * pc + 2 * j, (2 * i) * pc + 2 * j + 1, (2 * i + 1) * pc + 2 * j + 1};
    // g_max(g1, g_max(g2, g_max(g3, g4))) with `if a < b {b} else {a}`
    int best = idx[3];
    for (int m = 2; m >= 0; --m)
        if (!(pg[idx[m]] < pg[best])) best = idx[m];
    (void)pr;
    ws.max_g[(size_t)pair * ws.max_stride + off + t] = pg[best];
    ws.max_pos[(size_t)pair * ws.max_stride + off + t] = pp[best];
}
// Picking at level l (0-based) (dso.rs:248-276). mask of level l: level 0 = all true; levels >= 1 = mask1/mask2 planes.
__global__ __launch_bounds__(256) void dso_pick_kernel(Geom g, DsoWs ws, int l) {
    const int pair = blockIdx.y;
    const DsoState st = ws.state[pair];
    if (st.done) return;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols, bs = st.base_size;
    int r = (rows + bs - 1) / bs, c = (cols + bs - 1) / bs, off = 0, moff = 0;
    for (int k = 0; k < l; ++k) {
        off += r * c;
        if (k >= 1) moff += r * c;
        r /= 2;
        c /= 2;
    }
    // limited_sequence stops when a side reaches 0: that level (and coarser ones) does not exist
    if (r == 0 || c == 0) return;
    const int mh = r, mw = c;  // the mask of this level has the shape of this level's maxima
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= mh * mw) return;
    const int i = t / mw, j = t - i * mw;
    if (i >= mh / 2 * 2 || j >= mw / 2 * 2) return;  // remainder rows / columns are never visited (dso.rs:263)
    uint8_t* mask_cur = ws.mask1 + (size_t)pair * ws.mask_stride + moff;         // level l (l >= 1)
    uint8_t* mask_next = ws.mask1 + (size_t)pair * ws.mask_stride + (l >= 1 ? moff + mh * mw : 0);
    const bool m = (l == 0) ? true : (mask_cur[t] != 0);
    const int nw = mw / 2;
    if (m) {
        const int mg = ws.max_g[(size_t)pair * ws.max_stride + off + t];
        const uint32_t pos = ws.max_pos[(size_t)pair * ws.max_stride + off + t];
        const int ig = pos / cols, jg = pos - ig * cols;
        const int rc = (cols + DSO_REGION - 1) / DSO_REGION;
        const float threshold = (float)ws.thresh[(size_t)pair * ws.n_regions + (ig / DSO_REGION) * rc + jg / DSO_REGION];
        float coef = 1.0f;
        for (int k = 0; k < l; ++k) coef *= 0.5f;  // threshold_level_coef *= threshold_factor
        if ((float)mg >= coef * threshold) {
            mask_next[(i / 2) * nw + j / 2] = 0;
            ws.picked[(size_t)pair * g.S0 + pos] = (uint8_t)(l + 1);
            atomicAdd(&ws.state[pair].count, 1);
        }
    } else {
        mask_next[(i / 2) * nw + j / 2] = 0;
    }
}
// End of a round (dso.rs:115-146): recurse with an adapted block size, or fix the outcome.
__global__ void dso_decide_kernel(DsoWs ws, int n_pairs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    DsoState st = ws.state[p];
    if (st.done) return;
    const float ratio = (float)st.count / (float)DSO_TARGET;
    const float ts = sqrtf(ratio) * ((float)st.base_size + 1.0f) - 1.0f;
    const int target_size = max(1, (int)roundf(ts));
    if (ratio < 0.8f || ratio > 4.0f) {
        if (target_size != st.base_size && st.iterations_left > 0) {
            st.base_size = target_size;
            st.iterations_left -= 1;
        } else {
            st.done = 1;
        }
    } else if (ratio > 1.1f) {
        st.random_keep = (int)(uint8_t)(255.0f / ratio);
        st.done = 1;
    } else {
        st.done = 1;
    }
    ws.state[p] = st;
}
__global__ __launch_bounds__(256) void dso_finalize_kernel(Geom g, DsoWs ws, uint8_t* __restrict__ mask_out) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.S0) return;
    const DsoState st = ws.state[pair];
    bool m = ws.picked[(size_t)pair * g.S0 + t] > 0;
    if (m && st.random_keep >= 0) {
        const int cols = g.lv[0].cols;
        const int i = t / cols, j = t - i * cols;
        const uint8_t r = (uint8_t)(dso_splitmix64(DSO_SEED ^ dso_splitmix64(((uint64_t)(uint32_t)i << 32) | (uint32_t)j)) & 0xff);
        m = r <= (uint8_t)st.random_keep;
    }
    mask_out[(size_t)pair * g.S0 + t] = m ? 1 : 0;
}

void launch_dso_mask(const Geom& g, Pyramid kf, DsoWs ws, uint8_t* mask_out, int n_pairs, hipStream_t s) {
    const int S0 = g.S0;
    dim3 gpix((S0 + 255) / 256, n_pairs);
    hipLaunchKernelGGL(dso_gradmag_kernel, gpix, dim3(256), 0, s, g, kf.level0, ws);
    hipLaunchKernelGGL(dso_region_median_kernel, dim3(ws.n_regions, n_pairs), dim3(256), 0, s, g, ws);
    hipLaunchKernelGGL(dso_region_threshold_kernel, dim3((ws.n_regions + 63) / 64, n_pairs), dim3(64), 0, s, g, ws);
    hipLaunchKernelGGL(dso_init_state_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, s, ws, n_pairs);
    for (int round = 0; round < 3; ++round) {  // first call + at most nb_iterations_left = 2 recursive calls
        hipLaunchKernelGGL(dso_blockmax_kernel, gpix, dim3(256), 0, s, g, ws);
        for (int l = 1; l < DSO_LEVELS; ++l) hipLaunchKernelGGL(dso_halve_max_kernel, gpix, dim3(256), 0, s, g, ws, l);
        for (int l = 0; l < DSO_LEVELS; ++l) hipLaunchKernelGGL(dso_pick_kernel, gpix, dim3(256), 0, s, g, ws, l);
        hipLaunchKernelGGL(dso_decide_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, s, ws, n_pairs);
    }
    hipLaunchKernelGGL(dso_finalize_kernel, gpix, dim3(256), 0, s, g, ws, mask_out);
}

// ------------------------------------------------------------------------------------------------------------
// Generic-mask keyframe path: level-0 mask -> inverse-depth pyramid (per-pixel planes, like the dense mode) -> per level, the
// usable pixels compacted in raster order into the record planes (deterministic: block-wide prefix sums, one workgroup per
// (pair, level)). The LM kernel then runs its record path over `capacity` slots per level and skips the empty tail.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_idepth_level0_kernel(Geom g, const uint16_t* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                                  PixelPlanes pp) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.S0) return;
    const uint16_t dz = depth[(size_t)pair * g.S0 + t];
    const bool known = mask[(size_t)pair * g.S0 + t] != 0 && dz != 0;  // zip_mask_map + from_depth (helper.rs:40-47, inverse_depth.rs:24-29)
    const size_t o = (size_t)pair * pp.stride + pp.off[0] + t;
    pp.iz[o] = known ? g.depth_scale / (float)dz : __builtin_nanf("");
    pp.v[o] = known ? g.idepth_variance : -1.0f;
}
__global__ __launch_bounds__(256) void mask_idepth_halve_kernel(Geom g, int l, PixelPlanes pp) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols;
    if (t >= rows * cols) return;
    const int y = t / cols, x = t - y * cols;
    const int fc = g.lv[l - 1].cols;
    const size_t cb = (size_t)pair * pp.stride + pp.off[l - 1] + (size_t)(2 * y) * fc + 2 * x;
    const size_t idx[4] = {cb, cb + fc, cb + 1, cb + fc + 1};  // a, b, c, d
    float dv[4], vv[4];
    int n = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float v = pp.v[idx[m]];
        if (v >= 0.f) {
            dv[n] = pp.iz[idx[m]];
            vv[n] = v;
            ++n;
        }
    }
    float od = __builtin_nanf(""), ov = -1.0f;
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
    const size_t o = (size_t)pair * pp.stride + pp.off[l] + t;
    pp.iz[o] = od;
    pp.v[o] = ov;
}
__device__ __forceinline__ void generic_write_record(const Records& rec, size_t slot, const Intr& k, int x, int y, float iz, int gx, int gy,
                                                     uint8_t tmpl) {
    const V3 P = back_project(k, (float)x, (float)y, 1.0f / iz);
    float J[6];
    warp_jacobian_at((float)gx, (float)gy, (float)x, (float)y, iz, k, J);
    rec.A[slot] = make_float4(P.x, P.y, P.z, (float)tmpl);
    rec.B[slot] = make_float4(J[0], J[1], J[2], J[3]);
    rec.C[slot] = make_float2(J[4], J[5]);
    rec.XY[slot] = (uint32_t)x | ((uint32_t)y << 16);
    rec.IZ[slot] = iz;
}
__global__ __launch_bounds__(256) void generic_records_kernel(Geom g, const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                               PixelPlanes pp, Records rec) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int pair = blockIdx.y, l = blockIdx.x;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols, n = rows * cols, cap = g.lv[l].n_slots;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* iz = pp.iz + (size_t)pair * pp.stride + pp.off[l];
    const uint8_t* img = level_ptr(g, kf0, kfu, pair, l);
    const size_t slot0 = (size_t)pair * g.slots_total + g.lv[l].slot_off;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int t = base + threadIdx.x;
        const float z = t < n ? iz[t] : __builtin_nanf("");
        const bool valid = !(z != z);
        const unsigned long long m = __ballot(valid);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_wave[w];
        const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        const int k = s_base + woff + before;
        if (valid && k < cap) {
            const int y = t / cols, x = t - y * cols;
            int gx, gy;
            grad_at(g, kf0, kfu, pair, l, x, y, &gx, &gy);
            generic_write_record(rec, slot0 + k, g.lv[l].k, x, y, z, gx, gy, img[t]);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += total;
        __syncthreads();
    }
    for (int k = min(s_base, cap) + threadIdx.x; k < cap; k += 256) {
        rec.A[slot0 + k] = make_float4(0.f, 0.f, 0.f, -1.0f);
        rec.XY[slot0 + k] = VORS_INVALID_XY;
    }
}

void launch_keyframe_generic(const Geom& g, Pyramid kf, const uint16_t* depth, const uint8_t* mask, PixelPlanes pp, Records rec,
                             int n_pairs, hipStream_t s) {
    hipLaunchKernelGGL(mask_idepth_level0_kernel, dim3((g.S0 + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, mask, pp);
    for (int l = 1; l < g.L; ++l)
        hipLaunchKernelGGL(mask_idepth_halve_kernel, dim3((g.lv[l].rows * g.lv[l].cols + 255) / 256, n_pairs), dim3(256), 0, s, g, l, pp);
    hipLaunchKernelGGL(generic_records_kernel, dim3(g.L, n_pairs), dim3(256), 0, s, g, kf.level0, kf.upper, pp, rec);
}

}  // namespace vors
