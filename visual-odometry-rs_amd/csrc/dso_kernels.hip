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
// Everything per pair is decided on the device (no host round trip): one workgroup per pair runs the up-to-3 rounds.
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

// Per 32x32 region (one workgroup): gradient magnitude (sqrt(((gx^2 + gy^2) / 4) as u16 as f32)) as u16, border 0
// (gradient.rs:49-65, candidates_dso.rs:42), written to the gmag plane, and the region median sorted[len / 2] through a
// 256-bin histogram (dso.rs:307-325). Also clears the pick stamps of the previous keyframe.
__global__ __launch_bounds__(256) void dso_gradmag_median_kernel(Geom g, const uint8_t* __restrict__ kf0, DsoWs ws) {
    __shared__ int hist[256];
    __shared__ int s_tot[4];
    const int pair = blockIdx.y;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    const int rc = (cols + DSO_REGION - 1) / DSO_REGION;
    const int ri = blockIdx.x / rc, rj = blockIdx.x - ri * rc;
    const int h = min(DSO_REGION, rows - ri * DSO_REGION), w = min(DSO_REGION, cols - rj * DSO_REGION);
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t* img = kf0 + (size_t)pair * g.S0;
    uint8_t* gm = ws.gmag + (size_t)pair * g.S0;
    uint8_t* pk = ws.picked + (size_t)pair * g.S0;
    // thread -> 4 consecutive pixels of one region row (8 threads per row, 32 rows per pass)
    const int ly = threadIdx.x >> 3, lx0 = (threadIdx.x & 7) * 4;
    const int y = ri * DSO_REGION + ly;
    if (ly < h) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lx = lx0 + k, x = rj * DSO_REGION + lx;
            if (lx < w) {
                int out = 0;
                if (x > 0 && y > 0 && x < cols - 1 && y < rows - 1) {
                    const uint8_t* p = img + (size_t)y * cols + x;
                    const int gx = (int)p[1] - (int)p[-1], gy = (int)p[cols] - (int)p[-cols];
                    out = (int)sqrtf((float)((gx * gx + gy * gy) / 4));  // <= 180
                }
                gm[(size_t)y * cols + x] = (uint8_t)out;
                pk[(size_t)y * cols + x] = 0;
                atomicAdd(&hist[out], 1);
            }
        }
    }
    __syncthreads();
    // first bin whose inclusive prefix count exceeds len / 2
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mine = hist[threadIdx.x];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_tot[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wave; ++q) woff += s_tot[q];
    incl += woff;
    const int k = (h * w) / 2;
    if (incl > k && incl - mine <= k) ws.median[(size_t)pair * ws.n_regions + blockIdx.x] = (uint16_t)threadIdx.x;
}

// All rounds of one pair in one workgroup (dso.rs:98-147): region thresholds, then up to three rounds of
//   block maxima at the current base size -> two halvings -> picking at the three levels -> decision,
// with workgroup barriers between the phases (the planes live in global memory; a barrier orders them within the workgroup).
// Picks are stamped with (round << 2 | level + 1) so that later rounds need not clear the plane.
template <typename F>
__device__ __forceinline__ void dso_for_each(int n, F f) {
    for (int t = threadIdx.x; t < n; t += blockDim.x) f(t);
}
__global__ __launch_bounds__(1024) void dso_rounds_kernel(Geom g, DsoWs ws) {
    __shared__ DsoState st;
    __shared__ int s_count;
    const int pair = blockIdx.x;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    const int rr = (rows + DSO_REGION - 1) / DSO_REGION, rc = (cols + DSO_REGION - 1) / DSO_REGION;
    const uint8_t* gm = ws.gmag + (size_t)pair * g.S0;
    uint8_t* max_g = ws.max_g + (size_t)pair * ws.max_stride;
    uint32_t* max_pos = ws.max_pos + (size_t)pair * ws.max_stride;
    uint8_t* mask1 = ws.mask1 + (size_t)pair * ws.mask_stride;
    uint8_t* picked = ws.picked + (size_t)pair * g.S0;
    const uint16_t* med = ws.median + (size_t)pair * ws.n_regions;
    uint16_t* thresh = ws.thresh + (size_t)pair * ws.n_regions;
    // region thresholds: a * (mean3x3(median) + b)^2 cast to u16 (dso.rs:284-303), a = 1.0, b = 3
    dso_for_each(ws.n_regions, [&](int t) {
        const int i = t / rc, j = t - i * rc;
        uint16_t sum = 0;
        int n = 0;
        for (int jj = max(0, j - 1); jj < min(rc, j + 2); ++jj)
            for (int ii = max(0, i - 1); ii < min(rr, i + 2); ++ii) {
                sum = (uint16_t)(sum + med[ii * rc + jj]);
                ++n;
            }
        const float tt = (float)sum / (float)n + 3.0f;
        thresh[t] = (uint16_t)(1.0f * tt * tt);
    });
    if (threadIdx.x == 0) {
        st.base_size = 4;
        st.iterations_left = 2;
        st.done = 0;
        st.random_keep = -1;
        st.count = 0;
        st.final_round = 0;
    }
    __syncthreads();
    for (int round = 0; round < 3; ++round) {  // first call + at most nb_iterations_left = 2 recursive calls
        if (st.done) break;
        const int bs = st.base_size;
        // block-pyramid geometry: level k has r[k] x c[k] blocks at offset off[k]; a level with an empty side (and the coarser
        // ones) does not exist (limited_sequence stops there)
        int r[DSO_LEVELS + 1], c[DSO_LEVELS + 1], off[DSO_LEVELS + 1], moff[DSO_LEVELS + 2];
        r[0] = (rows + bs - 1) / bs;
        c[0] = (cols + bs - 1) / bs;
        off[0] = 0;
        for (int k = 1; k <= DSO_LEVELS; ++k) {
            r[k] = r[k - 1] / 2;
            c[k] = c[k - 1] / 2;
            off[k] = off[k - 1] + r[k - 1] * c[k - 1];
        }
        moff[1] = 0;  // mask of level k >= 1 at moff[k] (level 0 is all true); moff[DSO_LEVELS] receives the discarded last mask
        for (int k = 2; k <= DSO_LEVELS + 1; ++k) moff[k] = moff[k - 1] + r[k - 1] * c[k - 1];
        if (threadIdx.x == 0) s_count = 0;
        // level-0 block maxima (dso.rs:192-222): first maximum in column-major order; masks of the next levels all true (dso.rs:259)
        dso_for_each(moff[DSO_LEVELS + 1], [&](int t) { mask1[t] = 1; });
        dso_for_each(r[0] * c[0], [&](int t) {
            const int bi = t / c[0], bj = t - bi * c[0];
            const int si = bi * bs, sj = bj * bs, ei = min(si + bs, rows), ej = min(sj + bs, cols);
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
            max_g[t] = (uint8_t)mg;
            max_pos[t] = (uint32_t)(mi * cols + mj);
        });
        __syncthreads();
        // block maxima of level l from level l-1 (max_of_four_gradients, dso.rs:225-241: ties keep the earlier of a,b,c,d)
        for (int l = 1; l < DSO_LEVELS; ++l) {
            const int pc = c[l - 1];
            const uint8_t* pg = max_g + off[l - 1];
            const uint32_t* pp = max_pos + off[l - 1];
            dso_for_each(r[l] * c[l], [&](int t) {
                const int i = t / c[l], j = t - i * c[l];
                const int idx[4] = {(2 * i) * pc + 2 * j, (2 * i + 1) * pc + 2 * j, (2 * i) * pc + 2 * j + 1, (2 * i + 1) * pc + 2 * j + 1};
                int best = idx[3];  // g_max(g1, g_max(g2, g_max(g3, g4))) with `if a < b {b} else {a}`
                for (int m = 2; m >= 0; --m)
                    if (!(pg[idx[m]] < pg[best])) best = idx[m];
                max_g[off[l] + t] = pg[best];
                max_pos[off[l] + t] = pp[best];
            });
            __syncthreads();
        }
        // picking, finest level first (dso.rs:248-276)
        float coef = 1.0f;  // threshold_level_coef *= threshold_factor
        for (int l = 0; l < DSO_LEVELS; ++l) {
            const int mh = r[l], mw = c[l], nw = mw / 2;
            const uint8_t* mask_cur = mask1 + (l >= 1 ? moff[l] : 0);
            uint8_t* mask_next = mask1 + moff[l + 1];
            int local = 0;
            dso_for_each(mh * mw, [&](int t) {
                const int i = t / mw, j = t - i * mw;
                if (i >= mh / 2 * 2 || j >= mw / 2 * 2) return;  // remainder rows / columns are never visited (dso.rs:263)
                const bool m = (l == 0) ? true : (mask_cur[t] != 0);
                if (m) {
                    const int mg = max_g[off[l] + t];
                    const uint32_t pos = max_pos[off[l] + t];
                    const int ig = pos / cols, jg = pos - ig * cols;
                    const float threshold = (float)thresh[(ig / DSO_REGION) * rc + jg / DSO_REGION];
                    if ((float)mg >= coef * threshold) {
                        mask_next[(i / 2) * nw + j / 2] = 0;
                        picked[pos] = (uint8_t)((round << 2) | (l + 1));
                        ++local;
                    }
                } else {
                    mask_next[(i / 2) * nw + j / 2] = 0;
                }
            });
            if (local) atomicAdd(&s_count, local);
            coef *= 0.5f;
            __syncthreads();
        }
        // end of the round (dso.rs:115-146): recurse with an adapted block size, or fix the outcome
        if (threadIdx.x == 0) {
            DsoState n = st;
            n.count = s_count;
            n.final_round = round;
            const float ratio = (float)n.count / (float)DSO_TARGET;
            const float ts = sqrtf(ratio) * ((float)n.base_size + 1.0f) - 1.0f;
            const int target_size = max(1, (int)roundf(ts));
            if (ratio < 0.8f || ratio > 4.0f) {
                if (target_size != n.base_size && n.iterations_left > 0) {
                    n.base_size = target_size;
                    n.iterations_left -= 1;
                } else {
                    n.done = 1;
                }
            } else if (ratio > 1.1f) {
                n.random_keep = (int)(uint8_t)(255.0f / ratio);
                n.done = 1;
            } else {
                n.done = 1;
            }
            st = n;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) ws.state[pair] = st;
}
__global__ __launch_bounds__(256) void dso_finalize_kernel(Geom g, DsoWs ws, uint8_t* __restrict__ mask_out) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.S0) return;
    const DsoState st = ws.state[pair];
    const int stamp = ws.picked[(size_t)pair * g.S0 + t];
    bool m = (stamp & 3) != 0 && (stamp >> 2) == st.final_round;
    if (m && st.random_keep >= 0) {
        const int cols = g.lv[0].cols;
        const int i = t / cols, j = t - i * cols;
        const uint8_t r = (uint8_t)(dso_splitmix64(DSO_SEED ^ dso_splitmix64(((uint64_t)(uint32_t)i << 32) | (uint32_t)j)) & 0xff);
        m = r <= (uint8_t)st.random_keep;
    }
    mask_out[(size_t)pair * g.S0 + t] = m ? 1 : 0;
}

void launch_dso_mask(const Geom& g, Pyramid kf, DsoWs ws, uint8_t* mask_out, int n_pairs, hipStream_t s) {
    hipLaunchKernelGGL(dso_gradmag_median_kernel, dim3(ws.n_regions, n_pairs), dim3(256), 0, s, g, kf.level0, ws);
    hipLaunchKernelGGL(dso_rounds_kernel, dim3(n_pairs), dim3(1024), 0, s, g, ws);
    hipLaunchKernelGGL(dso_finalize_kernel, dim3((g.S0 + 255) / 256, n_pairs), dim3(256), 0, s, g, ws, mask_out);
}

// ------------------------------------------------------------------------------------------------------------
// Generic-mask keyframe path: level-0 mask -> inverse-depth pyramid (per-pixel planes, like the dense mode) -> per level, the
// usable pixels compacted in raster order into the record planes (deterministic: per-chunk counts, then prefix sums). The LM
// kernel then runs its record path over `capacity` slots per level and skips the empty tail.
// ------------------------------------------------------------------------------------------------------------
// Level-0 inverse depth is never stored: zip_mask_map + from_depth (helper.rs:40-47, inverse_depth.rs:24-29) evaluated where needed.
__device__ __forceinline__ float level0_idepth(const Geom& g, const uint16_t* __restrict__ depth, const uint8_t* __restrict__ mask, size_t i) {
    if (mask[i] == 0) return __builtin_nanf("");
    const uint16_t dz = depth[i];
    return dz != 0 ? g.depth_scale / (float)dz : __builtin_nanf("");
}
// Level l >= 1 of the inverse-depth pyramid from level l - 1 (strategy_dso_mean, inverse_depth.rs:81-98; children in order a,b,c,d).
__global__ __launch_bounds__(256) void mask_idepth_halve_kernel(Geom g, int l, const uint16_t* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                                 PixelPlanes pp) {
    const int pair = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = g.lv[l].rows, cols = g.lv[l].cols;
    if (t >= rows * cols) return;
    const int y = t / cols, x = t - y * cols;
    const int fc = g.lv[l - 1].cols;
    float dv[4], vv[4];
    int n = 0;
    if (l == 1) {
        const size_t cb = (size_t)pair * g.S0 + (size_t)(2 * y) * fc + 2 * x;
        const size_t idx[4] = {cb, cb + fc, cb + 1, cb + fc + 1};  // a, b, c, d
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float z = level0_idepth(g, depth, mask, idx[m]);
            if (z == z) {
                dv[n] = z;
                vv[n] = g.idepth_variance;
                ++n;
            }
        }
    } else {
        const size_t cb = (size_t)pair * pp.stride + pp.off[l - 1] + (size_t)(2 * y) * fc + 2 * x;
        const size_t idx[4] = {cb, cb + fc, cb + 1, cb + fc + 1};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float v = pp.v[idx[m]];
            if (v >= 0.f) {
                dv[n] = pp.iz[idx[m]];
                vv[n] = v;
                ++n;
            }
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
__device__ __forceinline__ float generic_idepth(const Geom& g, const PixelPlanes& pp, const uint16_t* __restrict__ depth,
                                                const uint8_t* __restrict__ mask, int pair, int l, int t) {
    if (l == 0) return level0_idepth(g, depth, mask, (size_t)pair * g.S0 + t);
    return pp.iz[(size_t)pair * pp.stride + pp.off[l] + t];
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
// Chunk c of a pair covers VORS_CHUNK_PX consecutive pixels (raster order) of one level: count pass, then a record pass in which every
// workgroup sums the counts of the chunks before it in its level (<= 300 values at 640x480) to get its first slot.
__device__ __forceinline__ int chunk_level(const PixelPlanes& pp, int L, int c) {
    int l = 0;
    while (l + 1 < L && c >= pp.chunk_off[l + 1]) ++l;
    return l;
}
__global__ __launch_bounds__(256) void generic_count_kernel(Geom g, const uint16_t* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                             PixelPlanes pp) {
    __shared__ int s_wave[4];
    const int pair = blockIdx.y, c = blockIdx.x;
    const int l = chunk_level(pp, g.L, c);
    const int n = g.lv[l].rows * g.lv[l].cols;
    const int base = (c - pp.chunk_off[l]) * VORS_CHUNK_PX;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < VORS_CHUNK_PX / 256; ++k) {
        const int t = base + k * 256 + threadIdx.x;
        const float z = t < n ? generic_idepth(g, pp, depth, mask, pair, l, t) : __builtin_nanf("");
        cnt += __popcll(__ballot(z == z));
    }
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) pp.counts[(size_t)pair * pp.chunks_total + c] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}
__global__ __launch_bounds__(256) void generic_records_kernel(Geom g, const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                               const uint16_t* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                               PixelPlanes pp, Records rec) {
    __shared__ int s_wave[4];
    __shared__ int s_sub[VORS_CHUNK_PX / 256][4];
    const int pair = blockIdx.y, c = blockIdx.x;
    const int l = chunk_level(pp, g.L, c);
    const int rows = g.lv[l].rows, cols = g.lv[l].cols, n = rows * cols, cap = g.lv[l].n_slots;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* img = level_ptr(g, kf0, kfu, pair, l);
    const size_t slot0 = (size_t)pair * g.slots_total + g.lv[l].slot_off;
    const int* counts = pp.counts + (size_t)pair * pp.chunks_total;
    // first slot of this chunk = usable pixels in the earlier chunks of the level
    int before_chunks = 0;
    for (int q = pp.chunk_off[l] + threadIdx.x; q < c; q += 256) before_chunks += counts[q];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) before_chunks += __shfl_xor(before_chunks, o);
    if (lane == 0) s_wave[wave] = before_chunks;
    const int base = (c - pp.chunk_off[l]) * VORS_CHUNK_PX;
    float z[VORS_CHUNK_PX / 256];
    int before[VORS_CHUNK_PX / 256];
#pragma unroll
    for (int k = 0; k < VORS_CHUNK_PX / 256; ++k) {
        const int t = base + k * 256 + threadIdx.x;
        z[k] = t < n ? generic_idepth(g, pp, depth, mask, pair, l, t) : __builtin_nanf("");
        const unsigned long long m = __ballot(z[k] == z[k]);
        before[k] = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (lane == 0) s_sub[k][wave] = __popcll(m);
    }
    __syncthreads();
    int run = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
#pragma unroll
    for (int k = 0; k < VORS_CHUNK_PX / 256; ++k) {
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_sub[k][w];
        const int slot = run + woff + before[k];
        const int t = base + k * 256 + threadIdx.x;
        if (z[k] == z[k] && slot < cap) {
            const int y = t / cols, x = t - y * cols;
            int gx, gy;
            grad_at(g, kf0, kfu, pair, l, x, y, &gx, &gy);
            generic_write_record(rec, slot0 + slot, g.lv[l].k, x, y, z[k], gx, gy, img[t]);
        }
        run += s_sub[k][0] + s_sub[k][1] + s_sub[k][2] + s_sub[k][3];
    }
    // the last chunk of the level marks the unused tail of the level's slots as empty
    if (c + 1 == pp.chunk_off[l + 1])
        for (int k = min(run, cap) + threadIdx.x; k < cap; k += 256) {
            rec.A[slot0 + k] = make_float4(0.f, 0.f, 0.f, -1.0f);
            rec.XY[slot0 + k] = VORS_INVALID_XY;
        }
}

void launch_keyframe_generic(const Geom& g, Pyramid kf, const uint16_t* depth, const uint8_t* mask, PixelPlanes pp, Records rec,
                             int n_pairs, hipStream_t s) {
    for (int l = 1; l < g.L; ++l)
        hipLaunchKernelGGL(mask_idepth_halve_kernel, dim3((g.lv[l].rows * g.lv[l].cols + 255) / 256, n_pairs), dim3(256), 0, s, g, l, depth, mask, pp);
    hipLaunchKernelGGL(generic_count_kernel, dim3(pp.chunks_total, n_pairs), dim3(256), 0, s, g, depth, mask, pp);
    hipLaunchKernelGGL(generic_records_kernel, dim3(pp.chunks_total, n_pairs), dim3(256), 0, s, g, kf.level0, kf.upper, depth, mask, pp, rec);
}

}  // namespace vors
