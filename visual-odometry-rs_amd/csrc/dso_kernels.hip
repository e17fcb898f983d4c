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

// Per 32x32 region (one WAVEFRONT, four regions per workgroup, no workgroup barrier): gradient magnitude
// (sqrt(((gx^2 + gy^2) / 4) as u16 as f32)) as u16, border 0 (gradient.rs:49-65, candidates_dso.rs:42), written to the gmag plane, and
// the region median sorted[len / 2] through a wave-private 256-bin histogram in LDS (dso.rs:307-325). Also clears the pick stamps when the
// pair's stamp epoch wraps (dso_next_epoch below). A lane owns 4 consecutive pixels in each of 4 rows (row = lane / 8 + 8 * pass).
// floor(sqrt(n)) for 0 <= n < 2^22, exactly: the hardware square root (1 ulp) truncated, then corrected by at most one. Equals
// (sqrtf((float)n) as u16) of gradient.rs:49-65 — the correctly rounded root of an integer below 2^22 never rounds up to the next integer
// (k - sqrt(k^2 - 1) > 1 / (2k) >> half an ulp of k).
__device__ __forceinline__ int isqrt_floor(int n) {
    int s = (int)__builtin_amdgcn_sqrtf((float)n);
    s -= (s * s > n) ? 1 : 0;
    s += ((s + 1) * (s + 1) <= n) ? 1 : 0;
    return s;
}
// Pick stamps carry the EPOCH of their selection (1 .. 15, DsoState::epoch, per pair) in the high nibble: a stamp of an earlier keyframe is
// recognisably stale, so the 307 k-byte stamp plane is cleared once per 15 selections (when the epoch wraps) instead of per keyframe
// (1.26 GB of stores per 4096 pairs in the first pass).
__device__ __forceinline__ int dso_next_epoch(int prev) { return (int)((unsigned)prev % 15u) + 1; }
// The same for 0 <= n < 2^16 in four instructions: sqrt(n) of a non-square n is at least 1 / (2 sqrt(n)) > 0.0019 below the next integer,
// the hardware root is within 1 ulp (3e-5 at 256), so adding 0.001 lifts an exact root that came out one ulp low over its integer and
// never lifts anything else (checked for every n by vors_debug_isqrt_mismatches / tests/test_gpu_parity.py).
__device__ __forceinline__ int isqrt_floor_u16(int n) { return (int)(__builtin_amdgcn_sqrtf((float)n) + 0.001f); }
__global__ void isqrt_check_kernel(int* mismatches) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < (1 << 16) && isqrt_floor_u16(n) != isqrt_floor(n)) atomicAdd(mismatches, 1);
}
int count_isqrt_u16_mismatches(hipStream_t s) {  // every argument of the four-instruction root against the corrected one (must be 0)
    int* flag = nullptr;
    int host = -1;
    if (hipMalloc(reinterpret_cast<void**>(&flag), sizeof(int)) != hipSuccess) return -1;
    if (hipMemsetAsync(flag, 0, sizeof(int), s) == hipSuccess) {
        hipLaunchKernelGGL(isqrt_check_kernel, dim3((1 << 16) / 256), dim3(256), 0, s, flag);
        if (hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) host = -1;
    }
    (void)hipFree(flag);
    return host;
}
__device__ __forceinline__ void dso_hist_add(int* hist, int v, bool active) {
    // most of a region usually shares one value (flat image areas): the first active lane's value is added once for the whole
    // wavefront, the remaining lanes fall back to LDS atomics
    const unsigned long long act = __ballot(active);
    if (act == 0) return;
    const int lead = __builtin_amdgcn_readlane(v, __builtin_ctzll(act));
    const unsigned long long same = __ballot(active && v == lead);
    if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(act)) atomicAdd(&hist[lead], __popcll(same));
    if (active && v != lead) atomicAdd(&hist[v], 1);
}
__global__ __launch_bounds__(256) void dso_gradmag_median_kernel(Geom g, const uint8_t* __restrict__ kf0, DsoWs ws, bool wide) {
    __shared__ int s_hist[4][256];
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int region = blockIdx.x * 4 + wave;
    if (region >= ws.n_regions) return;
    int* hist = s_hist[wave];
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    const int rc = (cols + DSO_REGION - 1) / DSO_REGION;
    const int ri = region / rc, rj = region - ri * rc;
    const int h = min(DSO_REGION, rows - ri * DSO_REGION), w = min(DSO_REGION, cols - rj * DSO_REGION);
    *reinterpret_cast<int4*>(&hist[4 * lane]) = make_int4(0, 0, 0, 0);
    const uint8_t* img = kf0 + (size_t)pair * g.S0;
    uint8_t* gm = ws.gmag + (size_t)pair * g.S0;
    uint8_t* pk = ws.picked + (size_t)pair * g.S0;
    const bool clear = dso_next_epoch(ws.state[pair].epoch) == 1;  // (the rounds kernel behind this one advances the epoch)
    const int lx0 = (lane & 7) * 4, x0 = rj * DSO_REGION + lx0;
    int out[4][4];
    if (wide) {  // cols % 4 == 0 and 4-byte aligned planes: a lane's 4 pixels are one dword in every plane
        uint32_t up[4], dn[4], mid[4];
        int left[4], right[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int ly = (lane >> 3) + 8 * ps, y = ri * DSO_REGION + ly;
            const bool in = ly < h && lx0 < w && y > 0 && y < rows - 1;
            const size_t o = (size_t)y * cols + x0;
            up[ps] = in ? *reinterpret_cast<const uint32_t*>(img + o - cols) : 0u;
            dn[ps] = in ? *reinterpret_cast<const uint32_t*>(img + o + cols) : 0u;
            mid[ps] = in ? *reinterpret_cast<const uint32_t*>(img + o) : 0u;
            left[ps] = (in && x0 > 0) ? img[o - 1] : 0;
            right[ps] = (in && x0 + 4 < cols) ? img[o + 4] : 0;
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int ly = (lane >> 3) + 8 * ps, y = ri * DSO_REGION + ly;
            const bool own = ly < h && lx0 < w, in = own && y > 0 && y < rows - 1;
            const int row[6] = {left[ps], (int)(mid[ps] & 0xff), (int)((mid[ps] >> 8) & 0xff), (int)((mid[ps] >> 16) & 0xff), (int)(mid[ps] >> 24), right[ps]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int gx = row[k + 2] - row[k], gy = (int)((dn[ps] >> (8 * k)) & 0xff) - (int)((up[ps] >> (8 * k)) & 0xff);
                const int x = x0 + k;
                out[ps][k] = (in && x > 0 && x < cols - 1) ? isqrt_floor_u16((gx * gx + gy * gy) / 4) : 0;  // <= 180
            }
            if (own) {
                const size_t o = (size_t)y * cols + x0;
                *reinterpret_cast<uint32_t*>(gm + o) =
                    (uint32_t)out[ps][0] | ((uint32_t)out[ps][1] << 8) | ((uint32_t)out[ps][2] << 16) | ((uint32_t)out[ps][3] << 24);
                if (clear) *reinterpret_cast<uint32_t*>(pk + o) = 0u;
            }
        }
    } else {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int ly = (lane >> 3) + 8 * ps, y = ri * DSO_REGION + ly;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lx = lx0 + k, x = x0 + k;
                int v = 0;
                if (ly < h && lx < w) {
                    const size_t o = (size_t)y * cols + x;
                    if (x > 0 && y > 0 && x < cols - 1 && y < rows - 1) {
                        const uint8_t* p = img + o;
                        const int gx = (int)p[1] - (int)p[-1], gy = (int)p[cols] - (int)p[-cols];
                        v = isqrt_floor_u16((gx * gx + gy * gy) / 4);
                    }
                    gm[o] = (uint8_t)v;
                    if (clear) pk[o] = 0;
                }
                out[ps][k] = v;
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
#pragma unroll
        for (int k = 0; k < 4; ++k) dso_hist_add(hist, out[ps][k], (lane >> 3) + 8 * ps < h && lx0 + k < w);
    // first bin whose inclusive prefix count exceeds len / 2 (LDS operations of one wavefront complete in order)
    const int4 b = *reinterpret_cast<const int4*>(&hist[4 * lane]);
    const int mine = b.x + b.y + b.z + b.w;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const int kmed = (h * w) / 2;
    if (incl > kmed && incl - mine <= kmed) {
        int acc = incl - mine, med = 4 * lane;
        const int bins[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (acc <= kmed && acc + bins[q] > kmed) med = 4 * lane + q;
            acc += bins[q];
        }
        ws.median[(size_t)pair * ws.n_regions + region] = (uint16_t)med;
    }
}

// The same for a STRIP of four horizontally adjacent regions per workgroup (128 pixels x 32 rows), when cols % 16 == 0 and the planes are
// 16-byte aligned: a thread owns 16 consecutive pixels of ONE row (wavefront w: rows 8 w .. 8 w + 7 of the strip, 8 lanes per row), so every
// load / store instruction of a wavefront moves eight full 128-byte lines instead of eight quarter lines (the per-region form above spends
// 1.2 of its 1.57 ms per 4096 pairs on its 20 narrow loads per lane). The four region histograms are shared by the workgroup: a thread
// adds its 16 values run by run (flat areas: one LDS atomic per thread); then wavefront w extracts the median of region w.
template <bool MAXIMA>
__global__ __launch_bounds__(256) void dso_gradmag_median_strip_kernel(Geom g, const uint8_t* __restrict__ kf0, DsoWs ws) {
    __shared__ int s_hist[4][256];
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    const int rc = (cols + DSO_REGION - 1) / DSO_REGION, strips = (cols + 127) / 128;
    const int ri = blockIdx.x / strips, sj = blockIdx.x - ri * strips;
    *reinterpret_cast<int4*>(&s_hist[0][0] + 4 * tid) = make_int4(0, 0, 0, 0);
    const uint8_t* img = kf0 + (size_t)pair * g.S0;
    uint8_t* gm = ws.gmag + (size_t)pair * g.S0;
    uint8_t* pk = ws.picked + (size_t)pair * g.S0;
    const bool clear = dso_next_epoch(ws.state[pair].epoch) == 1;
    const int x0 = sj * 128 + (lane & 7) * 16, y = ri * DSO_REGION + wave * 8 + (lane >> 3);
    const bool own = x0 < cols && y < rows;  // (cols % 16 == 0: a thread's 16 pixels are all inside or all outside)
    const int reg = (lane & 7) >> 1;         // region of the strip this thread's pixels belong to
    int out[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) out[k] = 0;
    if (own) {
        const size_t o = (size_t)y * cols + x0;
        const bool yin = y > 0 && y < rows - 1;
        if (yin) {
            const uint4 up = *reinterpret_cast<const uint4*>(img + o - cols), md = *reinterpret_cast<const uint4*>(img + o),
                        dn = *reinterpret_cast<const uint4*>(img + o + cols);
            const int left = x0 > 0 ? img[o - 1] : 0, right = x0 + 16 < cols ? img[o + 16] : 0;
            const uint32_t u[4] = {up.x, up.y, up.z, up.w}, m[4] = {md.x, md.y, md.z, md.w}, d[4] = {dn.x, dn.y, dn.z, dn.w};
            int row[18];
            row[0] = left;
            row[17] = right;
#pragma unroll
            for (int k = 0; k < 16; ++k) row[k + 1] = (int)((m[k >> 2] >> (8 * (k & 3))) & 0xff);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int gx = row[k + 2] - row[k];
                const int gy = (int)((d[k >> 2] >> (8 * (k & 3))) & 0xff) - (int)((u[k >> 2] >> (8 * (k & 3))) & 0xff);
                const int x = x0 + k;
                out[k] = (x > 0 && x < cols - 1) ? isqrt_floor_u16((gx * gx + gy * gy) / 4) : 0;  // <= 180
            }
        }
        uint32_t w4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            w4[q] = (uint32_t)out[4 * q] | ((uint32_t)out[4 * q + 1] << 8) | ((uint32_t)out[4 * q + 2] << 16) | ((uint32_t)out[4 * q + 3] << 24);
        *reinterpret_cast<uint4*>(gm + o) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        if (clear) *reinterpret_cast<uint4*>(pk + o) = make_uint4(0u, 0u, 0u, 0u);
    }
    if constexpr (MAXIMA) {  // The first selection round's block maxima (dso.rs:192-222 at the initial block size 4; dso_rounds_kernel then skips its pass over the
        // gmag plane in round 0): a 4 x 4 block = 4 pixels of each of 4 threads (rows: lanes 8 and 16 apart, exchanged through the swizzle
        // crossbar). First maximum in column-major order = the largest key (value << 4 | 15 - (4 j + i)). The thread of a block's first row
        // stores its four blocks at once (4 bytes + 16 bytes) where the pair's planes are aligned for that.
        const int i = (lane >> 3) & 3;
        int key[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int k = -1;
            if (own) {
#pragma unroll
                for (int j = 0; j < 4; ++j) k = max(k, (out[4 * q + j] << 4) | (15 - (4 * j + i)));
            }
            k = max(k, __builtin_amdgcn_ds_swizzle(k, 0x201f));  // lane ^ 8
            k = max(k, __builtin_amdgcn_ds_swizzle(k, 0x401f));  // lane ^ 16
            key[q] = k;
        }
        if (own && i == 0) {
            uint8_t* max_g = ws.max_g + (size_t)pair * ws.max_stride;
            uint32_t* max_pos = ws.max_pos + (size_t)pair * ws.max_stride;
            const int t = (y >> 2) * (cols >> 2) + (x0 >> 2);
            uint32_t pos[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = 15 - (key[q] & 15);
                pos[q] = (uint32_t)((y + (idx & 3)) * cols + x0 + 4 * q + (idx >> 2));
            }
            if ((reinterpret_cast<uintptr_t>(max_g + t) & 3) == 0 && (reinterpret_cast<uintptr_t>(max_pos + t) & 15) == 0) {
                *reinterpret_cast<uint32_t*>(max_g + t) = (uint32_t)(key[0] >> 4) | ((uint32_t)(key[1] >> 4) << 8) | ((uint32_t)(key[2] >> 4) << 16) | ((uint32_t)(key[3] >> 4) << 24);
                *reinterpret_cast<uint4*>(max_pos + t) = make_uint4(pos[0], pos[1], pos[2], pos[3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    max_g[t + q] = (uint8_t)(key[q] >> 4);
                    max_pos[t + q] = pos[q];
                }
            }
        }
    }
    __syncthreads();
    if (own) {  // the thread's 16 values into its region's histogram, run by run
        int* hist = s_hist[reg];
        int cur = out[0], cnt = 1;
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            if (out[k] == cur) {
                ++cnt;
            } else {
                atomicAdd(&hist[cur], cnt);
                cur = out[k];
                cnt = 1;
            }
        }
        atomicAdd(&hist[cur], cnt);
    }
    __syncthreads();
    // wavefront w: the median of region w of the strip = first bin whose inclusive prefix count exceeds len / 2
    const int rj = sj * 4 + wave;
    if (rj >= rc) return;
    const int h = min(DSO_REGION, rows - ri * DSO_REGION), w = min(DSO_REGION, cols - rj * DSO_REGION);
    const int4 b = *reinterpret_cast<const int4*>(&s_hist[wave][4 * lane]);
    const int mine = b.x + b.y + b.z + b.w;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const int kmed = (h * w) / 2;
    if (incl > kmed && incl - mine <= kmed) {
        int acc = incl - mine, med = 4 * lane;
        const int bins[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (acc <= kmed && acc + bins[q] > kmed) med = 4 * lane + q;
            acc += bins[q];
        }
        ws.median[(size_t)pair * ws.n_regions + ri * rc + rj] = (uint16_t)med;
    }
}

__device__ __forceinline__ void kf_wave_sync_dso() {  // (LDS operations of one wavefront execute in order; this keeps the compiler from moving them)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// The same with FOUR ROWS per thread (16 x 4 pixels = four whole 4 x 4 blocks): a WAVEFRONT owns a strip (8 lane columns x 8 row groups =
// 128 x 32 pixels, four regions, its own four histograms), a workgroup four strips. Six row loads per four output rows instead of twelve,
// the block maxima without any lane exchange, no workgroup barrier, a quarter of the wavefronts.
template <bool MAXIMA>
__global__ __launch_bounds__(256) void dso_gradmag_median_strip4_kernel(Geom g, const uint8_t* __restrict__ kf0, DsoWs ws) {
    __shared__ int s_hist[4][4][256];  // [wavefront][region of its strip][bin]
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = g.lv[0].rows, cols = g.lv[0].cols;
    const int rr = (rows + DSO_REGION - 1) / DSO_REGION, rc = (cols + DSO_REGION - 1) / DSO_REGION, strips = (cols + 127) / 128;
    const int strip = blockIdx.x * 4 + wave;
    if (strip >= rr * strips) return;  // (no workgroup barrier below)
    const int ri = strip / strips, sj = strip - ri * strips;
    int(*hist4)[256] = s_hist[wave];
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<int4*>(&hist4[0][0] + 4 * (lane + 64 * q)) = make_int4(0, 0, 0, 0);
    const uint8_t* img = kf0 + (size_t)pair * g.S0;
    uint8_t* gm = ws.gmag + (size_t)pair * g.S0;
    uint8_t* pk = ws.picked + (size_t)pair * g.S0;
    const bool clear = dso_next_epoch(ws.state[pair].epoch) == 1;
    const int x0 = sj * 128 + (lane & 7) * 16, y0 = ri * DSO_REGION + 4 * (lane >> 3);
    const bool xin = x0 < cols;  // (cols % 16 == 0: a thread's 16 pixels of a row are all inside or all outside)
    int* hist = hist4[(lane & 7) >> 1];
    // the six image rows y0 - 1 .. y0 + 4 (clamped: a clamped row is only ever the neighbour of a border row, whose values are 0 anyway)
    uint4 rw[6];
    int lf[4], rt[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int y = min(max(y0 - 1 + k, 0), rows - 1);
        rw[k] = xin ? *reinterpret_cast<const uint4*>(img + (size_t)y * cols + x0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = min(y0 + k, rows - 1);
        const size_t o = (size_t)y * cols + x0;
        lf[k] = (xin && x0 > 0) ? img[o - 1] : 0;
        rt[k] = (xin && x0 + 16 < cols) ? img[o + 16] : 0;
    }
    int key[4] = {-1, -1, -1, -1};
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // row y0 + i
        const int y = y0 + i;
        const bool own = xin && y < rows;
        int out[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) out[k] = 0;
        if (own && y > 0 && y < rows - 1) {
            const uint32_t u[4] = {rw[i].x, rw[i].y, rw[i].z, rw[i].w}, m[4] = {rw[i + 1].x, rw[i + 1].y, rw[i + 1].z, rw[i + 1].w},
                           d[4] = {rw[i + 2].x, rw[i + 2].y, rw[i + 2].z, rw[i + 2].w};
            int row[18];
            row[0] = lf[i];
            row[17] = rt[i];
#pragma unroll
            for (int k = 0; k < 16; ++k) row[k + 1] = (int)((m[k >> 2] >> (8 * (k & 3))) & 0xff);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int gx = row[k + 2] - row[k];
                const int gy = (int)((d[k >> 2] >> (8 * (k & 3))) & 0xff) - (int)((u[k >> 2] >> (8 * (k & 3))) & 0xff);
                const int x = x0 + k;
                out[k] = (x > 0 && x < cols - 1) ? isqrt_floor_u16((gx * gx + gy * gy) / 4) : 0;  // <= 180
            }
        }
        if (own) {
            const size_t o = (size_t)y * cols + x0;
            uint32_t w4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w4[q] = (uint32_t)out[4 * q] | ((uint32_t)out[4 * q + 1] << 8) | ((uint32_t)out[4 * q + 2] << 16) | ((uint32_t)out[4 * q + 3] << 24);
            *reinterpret_cast<uint4*>(gm + o) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            if (clear) *reinterpret_cast<uint4*>(pk + o) = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (MAXIMA) {  // first maximum in column-major order = the largest key (value << 4 | 15 - (4 j + i))
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) key[q] = max(key[q], (out[4 * q + j] << 4) | (15 - (4 * j + i)));
            }
            int cur = out[0], cnt = 1;  // the row's 16 values into the region's histogram, run by run
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                if (out[k] == cur) {
                    ++cnt;
                } else {
                    atomicAdd(&hist[cur], cnt);
                    cur = out[k];
                    cnt = 1;
                }
            }
            atomicAdd(&hist[cur], cnt);
        }
    }
    if constexpr (MAXIMA) {
        if (xin && y0 < rows) {
            uint8_t* max_g = ws.max_g + (size_t)pair * ws.max_stride;
            uint32_t* max_pos = ws.max_pos + (size_t)pair * ws.max_stride;
            const int t = (y0 >> 2) * (cols >> 2) + (x0 >> 2);
            uint32_t pos[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = 15 - (key[q] & 15);
                pos[q] = (uint32_t)((y0 + (idx & 3)) * cols + x0 + 4 * q + (idx >> 2));
            }
            if ((reinterpret_cast<uintptr_t>(max_g + t) & 3) == 0 && (reinterpret_cast<uintptr_t>(max_pos + t) & 15) == 0) {
                *reinterpret_cast<uint32_t*>(max_g + t) = (uint32_t)(key[0] >> 4) | ((uint32_t)(key[1] >> 4) << 8) | ((uint32_t)(key[2] >> 4) << 16) | ((uint32_t)(key[3] >> 4) << 24);
                *reinterpret_cast<uint4*>(max_pos + t) = make_uint4(pos[0], pos[1], pos[2], pos[3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    max_g[t + q] = (uint8_t)(key[q] >> 4);
                    max_pos[t + q] = pos[q];
                }
            }
        }
    }
    kf_wave_sync_dso();
    // the medians of the strip's four regions = first bin whose inclusive prefix count exceeds len / 2
    const int h = min(DSO_REGION, rows - ri * DSO_REGION);
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int rj = sj * 4 + q;
        if (rj >= rc) break;
        const int w = min(DSO_REGION, cols - rj * DSO_REGION);
        const int4 b = *reinterpret_cast<const int4*>(&hist4[q][4 * lane]);
        const int mine = b.x + b.y + b.z + b.w;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const int kmed = (h * w) / 2;
        if (incl > kmed && incl - mine <= kmed) {
            int acc = incl - mine, med = 4 * lane;
            const int bins[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                if (acc <= kmed && acc + bins[qq] > kmed) med = 4 * lane + qq;
                acc += bins[qq];
            }
            ws.median[(size_t)pair * ws.n_regions + ri * rc + rj] = (uint16_t)med;
        }
    }
}

// All rounds of one pair in one workgroup (dso.rs:98-147): region thresholds, then up to three rounds of
//   block maxima at the current base size -> two halvings -> picking at the three levels -> decision,
// with workgroup barriers between the phases (the planes live in global memory; a barrier orders them within the workgroup).
// Picks are stamped with (epoch << 4 | round << 2 | level + 1) so that neither later rounds nor later keyframes need to clear the plane.
template <typename F>
__device__ __forceinline__ void dso_for_each(int n, F f) {
    for (int t = threadIdx.x; t < n; t += blockDim.x) f(t);
}
// The same with DSO_BATCH items per thread and trip in two steps — every item's loads are issued before the first item's stores: the phases
// of the rounds kernel are chains of dependent global round trips (25 trips of 2 at the finest level with one item at a time: 50 of the
// ~90 us a workgroup lasts), and byte stores in between keep the compiler from overlapping the items itself.
#define DSO_BATCH 4
template <typename T, typename L, typename S>
__device__ __forceinline__ void dso_for_each_batched(int n, L load, S store) {
    for (int t0 = threadIdx.x; t0 < n; t0 += (int)blockDim.x * DSO_BATCH) {
        T in[DSO_BATCH];
#pragma unroll
        for (int u = 0; u < DSO_BATCH; ++u) {
            const int t = t0 + u * (int)blockDim.x;
            if (t < n) in[u] = load(t);
        }
#pragma unroll
        for (int u = 0; u < DSO_BATCH; ++u) {
            const int t = t0 + u * (int)blockDim.x;
            if (t < n) store(t, in[u]);
        }
    }
}
#define DSO_LDS_REGIONS 2048  // region thresholds in LDS up to this many regions (640x480: 300, 1920x1080: 2040)
// `out` (when out.gsort is set — the sparse keyframe form): the picks of the final round are also kept as a LIST, and the workgroup turns
// it straight into what mask_sparse_scan_kernel would extract from the stamp plane — the usable picks (final mask && depth != 0) as
// (Morton code << 16 | depth) words in out.gsort, their number in *out.count — so that the 307 k-pixel plane is never scanned for ~2000
// picks (0.97 ms per 4096 pairs). A list that overflows publishes cap_n + 1: the pair is then scanned band by band from the stamps.
struct DsoListOut {
    const uint16_t* depth;  // [pairs][S0]
    uint64_t* gsort;        // per pair at pair * gsort_stride (in 64-bit words)
    size_t gsort_stride;
    int* count;             // per pair at pair * count_stride
    int count_stride, cap_n;
};
__device__ __forceinline__ uint32_t morton_part(uint32_t v);
__global__ __launch_bounds__(1024) void dso_rounds_kernel(Geom g, DsoWs ws, DsoListOut out, int have_first_maxima, int stamps_always) {
    __shared__ DsoState st;
    __shared__ int s_count, s_list_n, s_out_n;
    __shared__ uint16_t s_thresh[DSO_LDS_REGIONS];
    const int pair = select_pair(g, blockIdx.x);
    if (pair < 0) return;
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
        if (t < DSO_LDS_REGIONS) s_thresh[t] = (uint16_t)(1.0f * tt * tt);
    });
    const bool thresh_in_lds = ws.n_regions <= DSO_LDS_REGIONS;
    if (threadIdx.x == 0) {
        st.base_size = 4;
        st.iterations_left = 2;
        st.done = 0;
        st.random_keep = -1;
        st.count = 0;
        st.final_round = 0;
        st.epoch = dso_next_epoch(ws.state[pair].epoch);
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
        if (threadIdx.x == 0) {
            s_count = 0;
            s_list_n = 0;
        }
        // ROUND 0 ON THE FIRST PASS'S BLOCK MAXIMA, UPPER LEVELS IN LDS (have_first_maxima = 2: the host found room for them). The generic
        // form below walks level after level through planes in global memory, a workgroup barrier and a global round trip or two per step;
        // here a thread owns a level-1 block: it reads its four level-0 children (8 independent loads), takes the level-0 picks (level 0's
        // mask is all true), and leaves the level-1 maximum and mask in LDS; the same one level up out of LDS; then the level-2 picks. Same
        // comparisons, same tie rules, same stamps; only the order of the pick list differs (it is sorted afterwards).
        const bool fast_round = round == 0 && have_first_maxima == 2;
        if (fast_round) {
            const bool stamps_now = !out.gsort || stamps_always;
            extern __shared__ __attribute__((aligned(16))) uint32_t dso_lds[];
            const int n1 = r[1] * c[1], n2 = r[2] * c[2];
            uint32_t* p1 = dso_lds;                                 // [n1] position of the level-1 maximum
            uint32_t* p2 = p1 + n1;                                 // [n2]
            uint8_t* g1 = reinterpret_cast<uint8_t*>(p2 + n2);      // [n1] its magnitude
            uint8_t* m1 = g1 + n1;                                  // [n1] level-1 mask
            uint8_t* g2 = m1 + n1;                                  // [n2]
            uint8_t* m2 = g2 + n2;                                  // [n2]
            __syncthreads();  // (s_count, s_list_n reset above)
            auto thr_at = [&](uint32_t pos) {
                const int ig = pos / cols, jg = pos - ig * cols;
                const int reg = (ig / DSO_REGION) * rc + jg / DSO_REGION;
                return (float)(thresh_in_lds ? s_thresh[reg] : thresh[reg]);
            };
            // (list form: the stamps are the fallback of two overflows that this kernel sees itself — it writes them at its end, then only:
            // 12 M scattered byte stores per 4096 pairs, 0.1 ms)
            auto pick = [&](uint32_t pos, int l) {
                if (stamps_now) picked[pos] = (uint8_t)((st.epoch << 4) | (round << 2) | (l + 1));
                if (out.gsort) {
                    const int idx = atomicAdd(&s_list_n, 1);
                    if (idx < ws.list_cap) ws.pick_list[(size_t)pair * ws.list_cap + idx] = pos;
                }
            };
            int local = 0;
            {   // level 0 picks + level 1
                const int pc = c[0], cl = c[1];
                for (int t0 = threadIdx.x; t0 < n1; t0 += (int)blockDim.x * 2) {
                    uint8_t gv[2][4];
                    uint32_t pv[2][4];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int t = t0 + u * (int)blockDim.x;
                        if (t < n1) {
                            const int i = t / cl, j = t - i * cl;
                            const int idx[4] = {(2 * i) * pc + 2 * j, (2 * i + 1) * pc + 2 * j, (2 * i) * pc + 2 * j + 1, (2 * i + 1) * pc + 2 * j + 1};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                gv[u][k] = max_g[idx[k]];
                                pv[u][k] = max_pos[idx[k]];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int t = t0 + u * (int)blockDim.x;
                        if (t < n1) {
                            bool any = false;
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if ((float)gv[u][k] >= 1.0f * thr_at(pv[u][k])) {
                                    pick(pv[u][k], 0);
                                    ++local;
                                    any = true;
                                }
                            int b = 3;  // g_max(g1, g_max(g2, g_max(g3, g4))) with `if a < b {b} else {a}` (dso.rs:225-241)
#pragma unroll
                            for (int m = 2; m >= 0; --m)
                                if (!(gv[u][m] < gv[u][b])) b = m;
                            g1[t] = b == 0 ? gv[u][0] : (b == 1 ? gv[u][1] : (b == 2 ? gv[u][2] : gv[u][3]));
                            p1[t] = b == 0 ? pv[u][0] : (b == 1 ? pv[u][1] : (b == 2 ? pv[u][2] : pv[u][3]));
                            m1[t] = any ? 0 : 1;
                        }
                    }
                }
            }
            __syncthreads();
            {   // level 1 picks + level 2
                const int pc = c[1], cl = c[2];
                for (int t = threadIdx.x; t < n2; t += (int)blockDim.x) {
                    const int i = t / cl, j = t - i * cl;
                    const int idx[4] = {(2 * i) * pc + 2 * j, (2 * i + 1) * pc + 2 * j, (2 * i) * pc + 2 * j + 1, (2 * i + 1) * pc + 2 * j + 1};
                    uint8_t gv[4];
                    bool keep = true;  // the level-2 mask: every child unmasked and none of them picked
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gv[k] = g1[idx[k]];
                        if (m1[idx[k]] != 0) {
                            if ((float)gv[k] >= 0.5f * thr_at(p1[idx[k]])) {
                                pick(p1[idx[k]], 1);
                                ++local;
                                keep = false;
                            }
                        } else {
                            keep = false;
                        }
                    }
                    int b = 3;
#pragma unroll
                    for (int m = 2; m >= 0; --m)
                        if (!(gv[m] < gv[b])) b = m;
                    g2[t] = gv[b];
                    p2[t] = p1[idx[b]];
                    m2[t] = keep ? 1 : 0;
                }
            }
            __syncthreads();
            {   // level 2 picks (remainder rows / columns are never visited, dso.rs:263)
                const int mh = r[2], mw = c[2];
                for (int t = threadIdx.x; t < n2; t += (int)blockDim.x) {
                    const int i = t / mw, j = t - i * mw;
                    if (i >= mh / 2 * 2 || j >= mw / 2 * 2) continue;
                    if (m2[t] != 0 && (float)g2[t] >= 0.25f * thr_at(p2[t])) {
                        pick(p2[t], 2);
                        ++local;
                    }
                }
            }
            if (local) atomicAdd(&s_count, local);
            __syncthreads();
        } else {
        // level-0 block maxima (dso.rs:192-222): first maximum in column-major order; masks of the next levels all true (dso.rs:259)
        dso_for_each(moff[DSO_LEVELS + 1], [&](int t) { mask1[t] = 1; });
        const bool dword_blocks = bs == 4 && cols % 4 == 0 && reinterpret_cast<uintptr_t>(gm) % 4 == 0;
        // (round 0 after the strip kernel: that kernel left the 4 x 4 block maxima here already)
        if (!(round == 0 && have_first_maxima)) dso_for_each(r[0] * c[0], [&](int t) {
            const int bi = t / c[0], bj = t - bi * c[0];
            const int si = bi * bs, sj = bj * bs, ei = min(si + bs, rows), ej = min(sj + bs, cols);
            if (dword_blocks && ei - si == 4) {  // the first round's 4x4 blocks: four dword loads instead of sixteen byte loads
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const uint32_t*>(gm + (size_t)(si + i) * cols + sj);
                int mg = (int)(w[0] & 0xff), mi = 0, mj = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int v = (int)((w[i] >> (8 * j)) & 0xff);
                        if (v > mg) {
                            mg = v;
                            mi = i;
                            mj = j;
                        }
                    }
                max_g[t] = (uint8_t)mg;
                max_pos[t] = (uint32_t)((si + mi) * cols + sj + mj);
                return;
            }
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
            struct HalveIn {
                int best;
                uint8_t g;
            };
            const int cl = c[l], offl = off[l];
            for (int t0 = threadIdx.x; t0 < r[l] * cl; t0 += (int)blockDim.x * DSO_BATCH) {
                HalveIn in[DSO_BATCH];
                uint32_t posv[DSO_BATCH];
#pragma unroll
                for (int u = 0; u < DSO_BATCH; ++u) {
                    const int t = t0 + u * (int)blockDim.x;
                    if (t < r[l] * cl) {
                        const int i = t / cl, j = t - i * cl;
                        const int idx[4] = {(2 * i) * pc + 2 * j, (2 * i + 1) * pc + 2 * j, (2 * i) * pc + 2 * j + 1, (2 * i + 1) * pc + 2 * j + 1};
                        const uint8_t gv[4] = {pg[idx[0]], pg[idx[1]], pg[idx[2]], pg[idx[3]]};
                        int b = 3;  // g_max(g1, g_max(g2, g_max(g3, g4))) with `if a < b {b} else {a}`
#pragma unroll
                        for (int m = 2; m >= 0; --m)
                            if (!(gv[m] < gv[b])) b = m;
                        in[u].best = b == 0 ? idx[0] : (b == 1 ? idx[1] : (b == 2 ? idx[2] : idx[3]));
                        in[u].g = b == 0 ? gv[0] : (b == 1 ? gv[1] : (b == 2 ? gv[2] : gv[3]));
                    }
                }
#pragma unroll
                for (int u = 0; u < DSO_BATCH; ++u)
                    if (t0 + u * (int)blockDim.x < r[l] * cl) posv[u] = pp[in[u].best];
#pragma unroll
                for (int u = 0; u < DSO_BATCH; ++u) {
                    const int t = t0 + u * (int)blockDim.x;
                    if (t < r[l] * cl) {
                        max_g[offl + t] = in[u].g;
                        max_pos[offl + t] = posv[u];
                    }
                }
            }
            __syncthreads();
        }
        // picking, finest level first (dso.rs:248-276)
        float coef = 1.0f;  // threshold_level_coef *= threshold_factor
        for (int l = 0; l < DSO_LEVELS; ++l) {
            const int mh = r[l], mw = c[l], nw = mw / 2;
            const uint8_t* mask_cur = mask1 + (l >= 1 ? moff[l] : 0);
            uint8_t* mask_next = mask1 + moff[l + 1];
            int local = 0;
            struct PickIn {
                int mg;
                uint32_t pos;
                int kind;  // 0: a remainder row / column, never visited (dso.rs:263); 1: masked off; 2: a candidate block
            };
            const int offl = off[l];
            dso_for_each_batched<PickIn>(
                mh * mw,
                [&](int t) {
                    const int i = t / mw, j = t - i * mw;
                    PickIn in;
                    in.kind = (i >= mh / 2 * 2 || j >= mw / 2 * 2) ? 0 : (((l == 0) ? true : (mask_cur[t] != 0)) ? 2 : 1);
                    in.mg = max_g[offl + t];
                    in.pos = max_pos[offl + t];
                    return in;
                },
                [&](int t, const PickIn& in) {
                    if (in.kind == 0) return;
                    const int i = t / mw, j = t - i * mw;
                    if (in.kind == 2) {
                        const uint32_t pos = in.pos;
                        const int ig = pos / cols, jg = pos - ig * cols;
                        const int reg = (ig / DSO_REGION) * rc + jg / DSO_REGION;
                        const float threshold = (float)(thresh_in_lds ? s_thresh[reg] : thresh[reg]);
                        if ((float)in.mg >= coef * threshold) {
                            mask_next[(i / 2) * nw + j / 2] = 0;
                            picked[pos] = (uint8_t)((st.epoch << 4) | (round << 2) | (l + 1));
                            ++local;
                            if (out.gsort) {
                                const int idx = atomicAdd(&s_list_n, 1);
                                if (idx < ws.list_cap) ws.pick_list[(size_t)pair * ws.list_cap + idx] = pos;
                            }
                        }
                    } else {
                        mask_next[(i / 2) * nw + j / 2] = 0;
                    }
                });
            if (local) atomicAdd(&s_count, local);
            coef *= 0.5f;
            __syncthreads();
        }
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
    if (!out.gsort) return;
    // The stamps of a final round 0 that ran in the LDS form without them (its level-1 / level-2 maxima and masks are still in LDS): the same
    // comparisons again, stamps only.
    const bool stamps_missing = have_first_maxima == 2 && !stamps_always && st.final_round == 0;
    auto stamp_round0 = [&]() {
        extern __shared__ __attribute__((aligned(16))) uint32_t dso_lds[];
        const int r0 = (rows + 3) / 4, c0 = (cols + 3) / 4, r1 = r0 / 2, c1 = c0 / 2, r2 = r1 / 2, c2 = c1 / 2, n1 = r1 * c1, n2 = r2 * c2;
        (void)r0;
        const uint32_t* p1 = dso_lds;
        const uint32_t* p2 = p1 + n1;
        const uint8_t* g1 = reinterpret_cast<const uint8_t*>(p2 + n2);
        const uint8_t* m1 = g1 + n1;
        const uint8_t* g2 = m1 + n1;
        const uint8_t* m2 = g2 + n2;
        const bool thresh_lds = ws.n_regions <= DSO_LDS_REGIONS;
        auto thr_at = [&](uint32_t pos) {
            const int ig = pos / cols, jg = pos - ig * cols;
            const int reg = (ig / DSO_REGION) * rc + jg / DSO_REGION;
            return (float)(thresh_lds ? s_thresh[reg] : thresh[reg]);
        };
        const uint8_t tag = (uint8_t)(st.epoch << 4);
        for (int t = threadIdx.x; t < n1; t += (int)blockDim.x) {
            const int i = t / c1, j = t - i * c1;
            const int idx[4] = {(2 * i) * c0 + 2 * j, (2 * i + 1) * c0 + 2 * j, (2 * i) * c0 + 2 * j + 1, (2 * i + 1) * c0 + 2 * j + 1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t pos = max_pos[idx[k]];
                if ((float)max_g[idx[k]] >= 1.0f * thr_at(pos)) picked[pos] = (uint8_t)(tag | 1);
            }
        }
        for (int t = threadIdx.x; t < n2; t += (int)blockDim.x) {
            const int i = t / c2, j = t - i * c2;
            const int idx[4] = {(2 * i) * c1 + 2 * j, (2 * i + 1) * c1 + 2 * j, (2 * i) * c1 + 2 * j + 1, (2 * i + 1) * c1 + 2 * j + 1};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (m1[idx[k]] != 0 && (float)g1[idx[k]] >= 0.5f * thr_at(p1[idx[k]])) picked[p1[idx[k]]] = (uint8_t)(tag | 2);
            if (i < r2 / 2 * 2 && j < c2 / 2 * 2 && m2[t] != 0 && (float)g2[t] >= 0.25f * thr_at(p2[t])) picked[p2[t]] = (uint8_t)(tag | 3);
        }
    };
    // the final round's picks -> the usable ones as sort words (any order: mask_sparse_records_kernel sorts them)
    if (threadIdx.x == 0) s_out_n = 0;
    __syncthreads();
    const int n_list = s_list_n;
    if (n_list > ws.list_cap) {  // (uniform) overflow: leave it to the scan of the stamp plane
        if (threadIdx.x == 0) out.count[(size_t)pair * out.count_stride] = out.cap_n + 1;
        if (stamps_missing) stamp_round0();
        return;
    }
    const uint32_t* list = ws.pick_list + (size_t)pair * ws.list_cap;
    const uint16_t* dp = out.depth + (size_t)pair * g.S0;
    uint64_t* gsort = out.gsort + (size_t)pair * out.gsort_stride;
    const DsoState fin = st;
    for (int i0 = threadIdx.x; i0 < n_list; i0 += (int)blockDim.x * DSO_BATCH) {  // (positions, then depths: two round trips per DSO_BATCH picks)
        int tv[DSO_BATCH];
        uint32_t dzv[DSO_BATCH];
#pragma unroll
        for (int u = 0; u < DSO_BATCH; ++u) tv[u] = (i0 + u * (int)blockDim.x < n_list) ? (int)list[i0 + u * (int)blockDim.x] : -1;
#pragma unroll
        for (int u = 0; u < DSO_BATCH; ++u) dzv[u] = tv[u] >= 0 ? dp[tv[u]] : 0u;
#pragma unroll
        for (int u = 0; u < DSO_BATCH; ++u) {
            if (tv[u] < 0) continue;
            const int t = tv[u], y = t / cols, x = t - y * cols;
            bool m = true;
            if (fin.random_keep >= 0) {  // random sub-sampling branch (dso.rs:140-143), the same counter hash as dso_final_mask
                const uint8_t rr8 = (uint8_t)(dso_splitmix64(DSO_SEED ^ dso_splitmix64(((uint64_t)(uint32_t)y << 32) | (uint32_t)x)) & 0xff);
                m = rr8 <= (uint8_t)fin.random_keep;
            }
            const uint32_t dz = dzv[u];
            if (m && dz != 0) {
                const int o = atomicAdd(&s_out_n, 1);
                if (o < out.cap_n) gsort[o] = ((uint64_t)(morton_part((uint32_t)y) | (morton_part((uint32_t)x) << 1)) << 16) | dz;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out.count[(size_t)pair * out.count_stride] = s_out_n;  // (> cap_n: band mode, like the scan kernel's overflow)
    if (stamps_missing && s_out_n > out.cap_n) stamp_round0();  // (uniform: band mode reads the stamps)
}
__device__ __forceinline__ uint8_t dso_final_mask(const DsoState& st, int stamp, int t, int cols) {
    bool m = (stamp & 3) != 0 && ((stamp >> 2) & 3) == st.final_round && (stamp >> 4) == st.epoch;
    if (m && st.random_keep >= 0) {  // random sub-sampling branch (dso.rs:140-143), counter-hash instead of thread_rng
        const int i = t / cols, j = t - i * cols;
        const uint8_t r = (uint8_t)(dso_splitmix64(DSO_SEED ^ dso_splitmix64(((uint64_t)(uint32_t)i << 32) | (uint32_t)j)) & 0xff);
        m = r <= (uint8_t)st.random_keep;
    }
    return m ? 1 : 0;
}
// Final-mask bits of the 16 stamps in w[0..3] (pixels t0 .. t0 + 15): the cheap test for all 16, then the sub-sampling hash only for
// the stamps that passed it, one per trip (a wavefront then pays the hash max-picks-per-lane times, not 16 times).
__device__ __forceinline__ uint32_t dso_final_bits16(const DsoState& st, const uint32_t w[4], int t0, int cols) {
    uint32_t cand = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int stamp = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
        if ((stamp & 3) != 0 && ((stamp >> 2) & 3) == st.final_round && (stamp >> 4) == st.epoch) cand |= 1u << k;
    }
    if (st.random_keep >= 0) {
        uint32_t keep = 0;
        for (uint32_t bb = cand; bb; bb &= bb - 1) {
            const int k = __ffs(bb) - 1, t = t0 + k, i = t / cols, j = t - i * cols;
            const uint8_t r = (uint8_t)(dso_splitmix64(DSO_SEED ^ dso_splitmix64(((uint64_t)(uint32_t)i << 32) | (uint32_t)j)) & 0xff);
            if (r <= (uint8_t)st.random_keep) keep |= 1u << k;
        }
        cand = keep;
    }
    return cand;
}
// 16 consecutive pixels per thread (one 16-byte load and store when the planes allow it).
__global__ __launch_bounds__(256) void dso_finalize_kernel(Geom g, DsoWs ws, uint8_t* __restrict__ mask_out) {
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int t0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (t0 >= g.S0) return;
    const DsoState st = ws.state[pair];
    const int cols = g.lv[0].cols;
    const size_t o = (size_t)pair * g.S0 + t0;
    if (g.S0 % 16 == 0) {
        const uint4 pk = *reinterpret_cast<const uint4*>(ws.picked + o);
        uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w}, out[4] = {0, 0, 0, 0};
        if ((pk.x | pk.y | pk.z | pk.w) != 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int stamp = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
                if (stamp) out[k >> 2] |= (uint32_t)dso_final_mask(st, stamp, t0 + k, cols) << (8 * (k & 3));
            }
        }
        *reinterpret_cast<uint4*>(mask_out + o) = make_uint4(out[0], out[1], out[2], out[3]);
    } else {
        for (int k = 0; k < 16 && t0 + k < g.S0; ++k) mask_out[o + k] = dso_final_mask(st, ws.picked[o + k], t0 + k, cols);
    }
}

// Selection up to the pick stamps (gradient magnitude + region medians, then all rounds of one pair in one workgroup).
static void launch_dso_selection(const Geom& g, Pyramid kf, DsoWs ws, int n_pairs, hipStream_t s, DsoListOut out = DsoListOut{}) {
    const bool wide_img = g.lv[0].cols % 4 == 0 && reinterpret_cast<uintptr_t>(kf.level0) % 4 == 0;
    int have_first_maxima = 0;
    if (g.lv[0].cols % 16 == 0 && g.S0 % 16 == 0 && reinterpret_cast<uintptr_t>(kf.level0) % 16 == 0) {
        const int rr = (g.lv[0].rows + DSO_REGION - 1) / DSO_REGION, strips = (g.lv[0].cols + 127) / 128;
        // the strip kernel also leaves round 0's block maxima (4096 pairs: +0.155 ms there, -0.205 ms in the rounds kernel);
        // VORS_DSO_FIRST_MAXIMA=0: the rounds kernel computes them from the gmag plane (A/B and tests)
        const char* fm = getenv("VORS_DSO_FIRST_MAXIMA");
        have_first_maxima = (fm && atoi(fm) == 0) ? 0 : 1;
        const char* e4 = getenv("VORS_DSO_STRIP4");  // four rows per thread, a wavefront per strip (4096 pairs: 0.88 -> 0.74 ms); 0: a row per thread (A/B and tests)
        if (!(e4 && atoi(e4) == 0)) {
            if (have_first_maxima) hipLaunchKernelGGL(dso_gradmag_median_strip4_kernel<true>, dim3((rr * strips + 3) / 4, n_pairs), dim3(256), 0, s, g, kf.level0, ws);
            else hipLaunchKernelGGL(dso_gradmag_median_strip4_kernel<false>, dim3((rr * strips + 3) / 4, n_pairs), dim3(256), 0, s, g, kf.level0, ws);
        } else if (have_first_maxima) hipLaunchKernelGGL(dso_gradmag_median_strip_kernel<true>, dim3(rr * strips, n_pairs), dim3(256), 0, s, g, kf.level0, ws);
        else hipLaunchKernelGGL(dso_gradmag_median_strip_kernel<false>, dim3(rr * strips, n_pairs), dim3(256), 0, s, g, kf.level0, ws);
    } else {
        hipLaunchKernelGGL(dso_gradmag_median_kernel, dim3((ws.n_regions + 3) / 4, n_pairs), dim3(256), 0, s, g, kf.level0, ws, wide_img);
    }
    // Threads per pair: the kernel is a chain of short phases over planes in global memory with a barrier between them; a large batch
    // runs faster with more, smaller workgroups per CU to interleave (round 3, 4096 pairs: 1.17 -> 1.02 ms with 512 threads), a small one
    // with the shortest chain per pair. Results do not depend on it.
    const char* e = getenv("VORS_DSO_ROUNDS_THREADS");
    const int forced = e ? atoi(e) : 0;
    // Round 4, sweep over 64 .. 1024 threads in steps of 64 at 64 / 512 / 2048 / 4096 pairs (profiles/r04_dso_rounds_threads.log): 768 threads
    // — twelve wavefronts, three per SIMD, two workgroups per CU — are fastest from 512 pairs on (4096 pairs: step 5.00 ms with 512 threads,
    // 5.06 with 1024, 4.84 with 768), 1024 for the small batches (the shortest chain per pair).
    const int rounds_threads = (forced >= 64 && forced <= 1024 && forced % 64 == 0) ? forced : (n_pairs >= 512 ? 768 : 1024);
    // round 0 with the upper block levels in LDS (10 bytes per level-1 block and per level-2 block of the 4 x 4 grid) where they fit next to a
    // second workgroup of the CU; VORS_DSO_ROUNDS_LDS=0: the generic form on the planes in global memory (A/B and tests)
    size_t lds = 0;
    if (have_first_maxima) {
        const int r1 = ((g.lv[0].rows + 3) / 4) / 2, c1 = ((g.lv[0].cols + 3) / 4) / 2, r2 = r1 / 2, c2 = c1 / 2;
        const size_t need = ((size_t)r1 * c1 + (size_t)r2 * c2) * 6 + 16;
        const char* el = getenv("VORS_DSO_ROUNDS_LDS");
        if (r2 > 0 && c2 > 0 && need <= 56 * 1024 && !(el && atoi(el) == 0)) {
            lds = need;
            have_first_maxima = 2;
        }
    }
    const char* es = getenv("VORS_DSO_STAMPS");  // 1: the LDS form writes its pick stamps always (A/B and tests), not only when something will read them
    hipLaunchKernelGGL(dso_rounds_kernel, dim3(n_pairs), dim3(rounds_threads), lds, s, g, ws, out, have_first_maxima, (es && atoi(es) != 0) ? 1 : 0);
}
// ------------------------------------------------------------------------------------------------------------
// Generic-mask keyframe path: level-0 mask -> inverse-depth pyramid (per-pixel planes, like the dense mode) -> per level, the
// usable pixels compacted in raster order into the record planes (deterministic: per-chunk counts, then prefix sums). The LM
// kernel then runs its record path over the slots in use (Records::n_used).
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
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
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
// Level 1 when cols(level 0) % 16 == 0: 8 consecutive level-1 pixels per thread, the 2 x 16 mask bytes below them in two loads;
// depth is only read under set mask bytes (candidates are sparse).
// FROM_DSO (rows(level 0) even as well): the mask bytes are decided here from the selector's pick stamps (what dso_finalize_kernel
// does) and written out, and the usable level-0 pixels are counted into their compaction chunks (integer atomics: deterministic)
// — one pass over the level-0 planes instead of three.
template <bool FROM_DSO>
__global__ __launch_bounds__(256) void mask_idepth_level1_wide_kernel(Geom g, const uint16_t* __restrict__ depth, uint8_t* __restrict__ mask,
                                                                       PixelPlanes pp, DsoWs ws) {
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int rows = g.lv[1].rows, cols = g.lv[1].cols, fc = g.lv[0].cols;
    const int t0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (t0 >= rows * cols) return;
    const int y = t0 / cols, x0 = t0 - y * cols;
    const size_t cb = (size_t)pair * g.S0 + (size_t)(2 * y) * fc + 2 * x0;
    uint4 m0, m1;
    if (FROM_DSO) {
        const DsoState st = ws.state[pair];
        const uint4 p0 = *reinterpret_cast<const uint4*>(ws.picked + cb), p1 = *reinterpret_cast<const uint4*>(ws.picked + cb + fc);
        uint32_t in[2][4] = {{p0.x, p0.y, p0.z, p0.w}, {p1.x, p1.y, p1.z, p1.w}}, out[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if ((p0.x | p0.y | p0.z | p0.w | p1.x | p1.y | p1.z | p1.w) != 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int stamp = (in[r][k >> 2] >> (8 * (k & 3))) & 0xff;
                    if (stamp) out[r][k >> 2] |= (uint32_t)dso_final_mask(st, stamp, (2 * y + r) * fc + 2 * x0 + k, fc) << (8 * (k & 3));
                }
        }
        m0 = make_uint4(out[0][0], out[0][1], out[0][2], out[0][3]);
        m1 = make_uint4(out[1][0], out[1][1], out[1][2], out[1][3]);
        *reinterpret_cast<uint4*>(mask + cb) = m0;
        *reinterpret_cast<uint4*>(mask + cb + fc) = m1;
    } else {
        m0 = *reinterpret_cast<const uint4*>(mask + cb);
        m1 = *reinterpret_cast<const uint4*>(mask + cb + fc);
    }
    float od[8], ov[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        od[k] = __builtin_nanf("");
        ov[k] = -1.0f;
    }
    int n_row[2] = {0, 0};  // usable level-0 pixels in the two rows this thread covers
    if ((m0.x | m0.y | m0.z | m0.w | m1.x | m1.y | m1.z | m1.w) != 0) {
        const uint32_t w0[4] = {m0.x, m0.y, m0.z, m0.w}, w1[4] = {m1.x, m1.y, m1.z, m1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // children a = (2y, 2x), b = (2y+1, 2x), c = (2y, 2x+1), d = (2y+1, 2x+1): bytes 2k, 2k+1 of the two rows
            const uint32_t top = (w0[k >> 1] >> (16 * (k & 1))) & 0xffffu, bot = (w1[k >> 1] >> (16 * (k & 1))) & 0xffffu;
            if ((top | bot) == 0) continue;
            const bool mk[4] = {(top & 0xff) != 0, (bot & 0xff) != 0, (top >> 8) != 0, (bot >> 8) != 0};
            const size_t idx[4] = {cb + 2 * k, cb + fc + 2 * k, cb + 2 * k + 1, cb + fc + 2 * k + 1};
            float dv[4];
            int n = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (!mk[m]) continue;
                const uint16_t dz = depth[idx[m]];
                if (dz != 0) {
                    dv[n++] = g.depth_scale / (float)dz;
                    n_row[m & 1] += 1;
                }
            }
            const float v = g.idepth_variance;
            if (n == 1) {
                od[k] = dv[0];
                ov[k] = v;
            } else if (n == 2) {
                ov[k] = v + v;
                od[k] = (dv[0] * v + dv[1] * v) / ov[k];
            } else if (n == 3) {
                ov[k] = v + v + v;
                od[k] = (dv[0] * v + dv[1] * v + dv[2] * v) / ov[k];
            } else if (n == 4) {
                ov[k] = v + v + v + v;
                od[k] = (dv[0] * v + dv[1] * v + dv[2] * v + dv[3] * v) / ov[k];
            }
        }
    }
    if (FROM_DSO) {  // candidates are sparse: few threads get here with a non-zero count
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (n_row[r]) {
                const int px = (2 * y + r) * fc + 2 * x0;  // the 16 pixels of a row segment never straddle a 4096-pixel chunk (px % 16 == 0)
                atomicAdd(&pp.counts[(size_t)pair * pp.chunks_total + pp.chunk_off[0] + px / VORS_CHUNK_PX], n_row[r]);
            }
    }
    const size_t o = (size_t)pair * pp.stride + pp.off[1] + t0;
    reinterpret_cast<float4*>(pp.iz + o)[0] = make_float4(od[0], od[1], od[2], od[3]);
    reinterpret_cast<float4*>(pp.iz + o)[1] = make_float4(od[4], od[5], od[6], od[7]);
    reinterpret_cast<float4*>(pp.v + o)[0] = make_float4(ov[0], ov[1], ov[2], ov[3]);
    reinterpret_cast<float4*>(pp.v + o)[1] = make_float4(ov[4], ov[5], ov[6], ov[7]);
}
__device__ __forceinline__ float generic_idepth(const Geom& g, const PixelPlanes& pp, const uint16_t* __restrict__ depth,
                                                const uint8_t* __restrict__ mask, int pair, int l, int t) {
    if (l == 0) return level0_idepth(g, depth, mask, (size_t)pair * g.S0 + t);
    return pp.iz[(size_t)pair * pp.stride + pp.off[l] + t];
}
// Chunk c of a pair covers VORS_CHUNK_PX consecutive pixels (raster order) of one level, 16 consecutive pixels per thread: count pass,
// then a record pass in which every workgroup sums the counts of the chunks before it in its level to get its first slot.
__device__ __forceinline__ int chunk_level(const PixelPlanes& pp, int L, int c) {
    int l = 0;
    while (l + 1 < L && c >= pp.chunk_off[l + 1]) ++l;
    return l;
}
// bit k set <=> pixel t0 + k of level l has a known inverse depth
__device__ __forceinline__ unsigned usable16(const Geom& g, const PixelPlanes& pp, const uint16_t* __restrict__ depth,
                                             const uint8_t* __restrict__ mask, int pair, int l, int t0, int n) {
    unsigned vb = 0;
    if (t0 >= n) return 0;
    if (l == 0) {
        const size_t o = (size_t)pair * g.S0 + t0;
        if (g.S0 % 16 == 0) {
            const uint4 m = *reinterpret_cast<const uint4*>(mask + o);
            if ((m.x | m.y | m.z | m.w) == 0) return 0;
            const uint32_t w[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (((w[k >> 2] >> (8 * (k & 3))) & 0xff) != 0 && depth[o + k] != 0) vb |= 1u << k;
        } else {
            for (int k = 0; k < 16 && t0 + k < n; ++k)
                if (mask[o + k] != 0 && depth[o + k] != 0) vb |= 1u << k;
        }
    } else {
        const float* iz = pp.iz + (size_t)pair * pp.stride + pp.off[l] + t0;  // 16-byte aligned: stride, off and t0 are multiples of 4
        if (t0 + 16 <= n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 z = reinterpret_cast<const float4*>(iz)[q];
                vb |= (z.x == z.x ? 1u : 0u) << (4 * q) | (z.y == z.y ? 2u : 0u) << (4 * q) | (z.z == z.z ? 4u : 0u) << (4 * q) |
                      (z.w == z.w ? 8u : 0u) << (4 * q);
            }
        } else {
            for (int k = 0; t0 + k < n; ++k)
                if (iz[k] == iz[k]) vb |= 1u << k;
        }
    }
    return vb;
}
__global__ __launch_bounds__(256) void generic_count_kernel(Geom g, const uint16_t* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                             PixelPlanes pp, int first_chunk) {
    __shared__ int s_wave[4];
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int c = first_chunk + blockIdx.x;
    const int l = chunk_level(pp, g.L, c);
    const int n = g.lv[l].rows * g.lv[l].cols;
    const int t0 = (c - pp.chunk_off[l]) * VORS_CHUNK_PX + threadIdx.x * 16;
    int cnt = __popc(usable16(g, pp, depth, mask, pair, l, t0, n));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) pp.counts[(size_t)pair * pp.chunks_total + c] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}
// Compaction: the usable pixels of a level, in raster order, get consecutive slots; only their coordinates are written here (the
// candidates are sparse: a handful of lanes per wavefront do this part), the records follow in a dense pass over the slots.
__global__ __launch_bounds__(256) void generic_compact_kernel(Geom g, const uint16_t* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                               PixelPlanes pp, Records rec) {
    __shared__ int s_before[4], s_wave[4];
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int c = blockIdx.x;
    const int l = chunk_level(pp, g.L, c);
    const int rows = g.lv[l].rows, cols = g.lv[l].cols, n = rows * cols, cap = g.lv[l].n_slots;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t slot0 = (size_t)pair * g.slots_total + g.lv[l].slot_off;
    const int* counts = pp.counts + (size_t)pair * pp.chunks_total;
    // first slot of this chunk = usable pixels in the earlier chunks of the level
    int before_chunks = 0;
    for (int q = pp.chunk_off[l] + threadIdx.x; q < c; q += 256) before_chunks += counts[q];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) before_chunks += __shfl_xor(before_chunks, o);
    const int t0 = (c - pp.chunk_off[l]) * VORS_CHUNK_PX + threadIdx.x * 16;
    unsigned vb = usable16(g, pp, depth, mask, pair, l, t0, n);
    // exclusive prefix of the per-thread counts over the workgroup
    const int mine = __popc(vb);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    if (lane == 0) s_before[wave] = before_chunks;
    __syncthreads();
    int slot = s_before[0] + s_before[1] + s_before[2] + s_before[3] + incl - mine;
    for (int w = 0; w < wave; ++w) slot += s_wave[w];
    while (vb) {
        const int k = __ffs(vb) - 1;
        vb &= vb - 1;
        if (slot < cap) {
            const int t = t0 + k;
            const int y = t / cols, x = t - y * cols;
            rec.S[slot0 + slot].xy = (uint32_t)x | ((uint32_t)y << 16);
        }
        ++slot;
    }
    // the last thread of the level's last chunk publishes how many of the level's slots are in use
    if (c + 1 == pp.chunk_off[l + 1] && threadIdx.x == 255) rec.n_used[(size_t)pair * VORS_MAX_LEVELS + l] = min(slot, cap);
}
// One thread per slot in use (grid-stride over a level's slots with GENERIC_BUILD_WGS workgroups): the record of the pixel.
#define GENERIC_BUILD_WGS 4
__global__ __launch_bounds__(256) void generic_build_records_kernel(Geom g, const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                                     const uint16_t* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                                     PixelPlanes pp, Records rec) {
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int l = blockIdx.x / GENERIC_BUILD_WGS, w = blockIdx.x - l * GENERIC_BUILD_WGS;
    const int n = rec.n_used[(size_t)pair * VORS_MAX_LEVELS + l], cols = g.lv[l].cols;
    const uint8_t* img = level_ptr(g, kf0, kfu, pair, l);
    const size_t slot0 = (size_t)pair * g.slots_total + g.lv[l].slot_off;
    for (int slot = w * 256 + threadIdx.x; slot < n; slot += GENERIC_BUILD_WGS * 256) {
        const uint32_t xy = rec.S[slot0 + slot].xy;
        const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16), t = y * cols + x;
        int gx, gy;
        grad_at(g, kf0, kfu, pair, l, x, y, &gx, &gy);
        rec.S[slot0 + slot].iz = generic_idepth(g, pp, depth, mask, pair, l, t);
        rec.S[slot0 + slot].tg = slim_pack_tg(img[t], gx, gy);
    }
}

// `dso` non-null: the mask is still in the selector's pick stamps; the fused level-1 pass finalizes it into `mask` on the way.
static void keyframe_from_mask(const Geom& g, Pyramid kf, const uint16_t* depth, uint8_t* mask, PixelPlanes pp, Records rec, const DsoWs* dso,
                               int n_pairs, hipStream_t s) {
    int first_chunk = 0;
    for (int l = 1; l < g.L; ++l) {
        const int n = g.lv[l].rows * g.lv[l].cols;
        if (l == 1 && dso) {  // (the caller checked the shape: cols % 16 == 0, rows even)
            launch_zero_ints(g, pp.counts, pp.chunks_total, n_pairs, s);
            hipLaunchKernelGGL(mask_idepth_level1_wide_kernel<true>, dim3((n / 8 + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, mask, pp, *dso);
            first_chunk = pp.chunk_off[1];  // the level-0 chunks have been counted
        } else if (l == 1 && g.lv[0].cols % 16 == 0) {
            hipLaunchKernelGGL(mask_idepth_level1_wide_kernel<false>, dim3((n / 8 + 255) / 256, n_pairs), dim3(256), 0, s, g, depth, mask, pp, DsoWs{});
        } else {
            hipLaunchKernelGGL(mask_idepth_halve_kernel, dim3((n + 255) / 256, n_pairs), dim3(256), 0, s, g, l, depth, mask, pp);
        }
    }
    hipLaunchKernelGGL(generic_count_kernel, dim3(pp.chunks_total - first_chunk, n_pairs), dim3(256), 0, s, g, depth, mask, pp, first_chunk);
    hipLaunchKernelGGL(generic_compact_kernel, dim3(pp.chunks_total, n_pairs), dim3(256), 0, s, g, depth, mask, pp, rec);
    hipLaunchKernelGGL(generic_build_records_kernel, dim3(GENERIC_BUILD_WGS * g.L, n_pairs), dim3(256), 0, s, g, kf.level0, kf.upper, depth, mask, pp,
                       rec);
}
// ------------------------------------------------------------------------------------------------------------
// Sparse form of the generic-mask path (the DSO selector keeps ~0.7 % of the pixels; going through dense per-level planes cost
// 4.9 ms per 4096 pairs against 1.1 ms for the whole coarse-to-fine keyframe stage). One workgroup per pair:
//   A  one pass over the level-0 plane (pick stamps -> final mask bytes, written out; or the mask itself), the usable pixels
//      (mask && depth != 0) appended as (Morton code of (x, y) << 16 | depth) — bit 0 of the code is y's bit 0, bit 1 x's bit 0, ...
//   B  bitonic sort of those words (LDS up to 4096 of them, global scratch beyond). In Morton order the children of EVERY node of
//      EVERY level are contiguous and appear as a = (2i, 2j), b = (2i+1, 2j), c = (2i, 2j+1), d = (2i+1, 2j+1): the order
//      inverse_depth.rs:81-98 fuses them in.
//   C  level l from level l - 1: a segmented fusion over runs of equal code >> 2 (strategy_dso_mean on the known children; a parent
//      outside the halved shape — odd trailing row / column, multires.rs:67-88 — does not exist), compacted in order.
//   D  per level the 12-byte records (coordinates, inverse depth, template + integer gradient gathered from the pyramid).
// Same values as the plane path, bit for bit; the lists come out in Morton order instead of raster order (the LM sums differ in
// their last bits only). Scratch: the (otherwise unused) pixel planes of the pair.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t morton_part(uint32_t v) {
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
__device__ __forceinline__ uint32_t morton_compact(uint32_t k) {
    k &= 0x55555555u;
    k = (k | (k >> 1)) & 0x33333333u;
    k = (k | (k >> 2)) & 0x0F0F0F0Fu;
    k = (k | (k >> 4)) & 0x00FF00FFu;
    k = (k | (k >> 8)) & 0x0000FFFFu;
    return k;
}
constexpr int SPARSE_LDS_N = 4096;  // entries of the LDS form: 48 KB (sort words, then key / inverse depth / weight in their place): three workgroups per CU
// Exclusive position of a flagged thread among the flagged threads of the workgroup (T threads), plus their number.
// Two barriers; `s_wave` is workgroup scratch.
template <int T>
__device__ __forceinline__ int block_rank(bool flag, int* s_wave, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    __syncthreads();  // (s_wave may still be read from the previous call)
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int pos = before, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) {
        pos += w < wave ? s_wave[w] : 0;
        tot += s_wave[w];
    }
    *total = tot;
    return pos;
}
// Phase A over the pixels [p0, p1) of the level-0 plane of one pair (whole rows): final mask bytes out, usable pixels appended to `gsort`.
struct SparseScan {
    const uint8_t* src;
    uint8_t* mout;
    const uint16_t* dp;
    uint64_t* gsort;
    DsoState st;
    int from_stamps, cols0, cap_n;
    bool vec;
};
template <int T>
__device__ __forceinline__ void sparse_scan(const SparseScan& q, int p0, int p1, int* s_n) {
    for (int t0 = p0 + (int)threadIdx.x * 16; t0 < p1; t0 += T * 16) {
        uint32_t bits = 0;
        if (q.vec) {
            const uint4 pk = *reinterpret_cast<const uint4*>(q.src + t0);
            if ((pk.x | pk.y | pk.z | pk.w) != 0) {
                const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int bq = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
                    if (bq && (!q.from_stamps || dso_final_mask(q.st, bq, t0 + k, q.cols0))) bits |= 1u << k;
                }
            }
            if (q.from_stamps && q.mout) {
                uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 16; ++k) o[k >> 2] |= ((bits >> k) & 1u) << (8 * (k & 3));
                *reinterpret_cast<uint4*>(q.mout + t0) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        } else {
            for (int k = 0; k < 16 && t0 + k < p1; ++k) {
                const int bq = q.src[t0 + k];
                const bool m = bq && (!q.from_stamps || dso_final_mask(q.st, bq, t0 + k, q.cols0));
                if (m) bits |= 1u << k;
                if (q.from_stamps && q.mout) q.mout[t0 + k] = m ? 1 : 0;
            }
        }
        uint32_t usable = 0;
        for (uint32_t bb = bits; bb; bb &= bb - 1) {
            const int k = __ffs(bb) - 1;
            if (q.dp[t0 + k] != 0) usable |= 1u << k;
        }
        if (usable) {
            int pos = atomicAdd(s_n, __popc(usable));
            for (uint32_t bb = usable; bb; bb &= bb - 1, ++pos) {
                const int k = __ffs(bb) - 1, t = t0 + k, y = t / q.cols0, x = t - y * q.cols0;
                if (pos < q.cap_n) q.gsort[pos] = ((uint64_t)(morton_part((uint32_t)y) | (morton_part((uint32_t)x) << 1)) << 16) | q.dp[t];
            }
        }
    }
}
// Phases B-D on n0 appended words. `a` (sort words) and `key` (three arrays of `cap_set`) are EITHER the LDS buffers OR the pair's global
// scratch; the function is inlined once per case so that the LDS case compiles to ds_ instructions (not flat ones).
template <int T>
__device__ __forceinline__ void sparse_flush(const Geom& g, const uint8_t* kf0, const uint8_t* kfu, int pair, const uint64_t* gsort, int n0,
                                             uint64_t* a, bool copy_in, uint32_t* key, int cap_set, const Records& rec, int* s_wave, int* s_out,
                                             uint32_t* s_prev, bool bitonic) {
    const int tid = threadIdx.x;
    // One set of (key, inverse depth, weight) arrays, rewritten in place level by level: a pass reads its inputs into registers, meets
    // at a barrier, then writes (the outputs of a level land at or below inputs already consumed).
    float* dd = reinterpret_cast<float*>(key + cap_set);
    float* vv = reinterpret_cast<float*>(key + 2 * cap_set);
    if (copy_in && !bitonic) {
        // LDS form, round 4: a two-step BUCKET sort instead of the bitonic network (66 dependent compare-exchange stages with an LDS
        // round trip and a barrier each: 0.45 of the kernel's 0.60 ms per 4096 pairs). The keys are distinct pixel positions, so:
        //  1. bucket = the high bits of the Morton code (at most 1024 buckets = square tiles of the image in Morton order): a counter
        //     per bucket, an element's slot in its bucket from the atomic's return value (any order);
        //  2. exclusive scan of the counters, elements scattered to bucket start + slot;
        //  3. an element's final place = bucket start + the number of bucket mates with a smaller word (a DSO pick list holds at most one
        //     pick per 4x4 block and a few from the recursive rounds: a handful of mates; any mask works, a dense one just loops longer).
        // Five barriers; the (key, inverse depth, weight) arrays are written straight from the registers.
        constexpr int Q = SPARSE_LDS_N / T, BINS = 1024, PER = BINS / T > 0 ? BINS / T : 1;
        static_assert(BINS % T == 0 || T > BINS, "bins per thread");
        uint32_t* hist = key + 2 * cap_set;  // (the weights' place: free until the arrays are written)
        const uint32_t code_max = morton_part((uint32_t)(g.lv[0].rows - 1)) | (morton_part((uint32_t)(g.lv[0].cols - 1)) << 1);
        const int nbits = 32 - __builtin_clz(code_max | 1u), shift = 16 + max(0, nbits - 10);
        for (int i = tid; i < BINS; i += T) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (tid + T * q < n0) atomicAdd(&hist[(uint32_t)(gsort[tid + T * q] >> shift)], 1u);
        __syncthreads();
        {
            uint32_t c[PER], sum = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                c[j] = tid * PER + j < BINS ? hist[tid * PER + j] : 0;
                sum += c[j];
            }
            const int lane = tid & 63, wave = tid >> 6;
            uint32_t incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) s_wave[wave] = (int)incl;
            __syncthreads();
            uint32_t before = incl - sum;
#pragma unroll
            for (int k = 0; k < T / 64; ++k) before += k < wave ? (uint32_t)s_wave[k] : 0u;
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (tid * PER + j < BINS) {
                    hist[tid * PER + j] = before;  // start of the bucket; the scatter below moves it to the bucket's end
                    before += c[j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (tid + T * q < n0) {  // (the words again, from L2: cheaper than 16 registers held across the scan)
                const uint64_t w = gsort[tid + T * q];
                a[atomicAdd(&hist[(uint32_t)(w >> shift)], 1u)] = w;
            }
        __syncthreads();
        uint64_t wp[Q];  // word | final place << 52 (a word has at most 48 bits: 32 of Morton code, 16 of depth)
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (tid + T * q < n0) {
                const uint64_t w = a[tid + T * q];
                const uint32_t bin = (uint32_t)(w >> shift);
                const int b0 = bin ? (int)hist[bin - 1] : 0, b1 = (int)hist[bin];
                int r = b0;
                for (int j = b0; j < b1; ++j) r += a[j] < w ? 1 : 0;
                wp[q] = w | ((uint64_t)r << 52);
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (tid + T * q < n0) {
                const int place = (int)(wp[q] >> 52);
                key[place] = (uint32_t)(wp[q] >> 16);
                dd[place] = g.depth_scale / (float)(uint32_t)(wp[q] & 0xffffu);  // from_depth, inverse_depth.rs:24-29
                vv[place] = g.idepth_variance;
            }
    } else {
        int P = 2;
        while (P < n0) P <<= 1;
        if (copy_in)
            for (int i = tid; i < P; i += T) a[i] = i < n0 ? gsort[i] : ~0ull;
        else
            for (int i = n0 + tid; i < P; i += T) a[i] = ~0ull;
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P; i += T) {
                    const int q = i ^ j;
                    if (q > i) {
                        const uint64_t x = a[i], y = a[q];
                        if ((x > y) == ((i & k) == 0)) {
                            a[i] = y;
                            a[q] = x;
                        }
                    }
                }
                __syncthreads();
            }
        if (copy_in) {  // LDS form: the arrays take the place of the sorted words (n0 <= 4096: four words per thread through registers)
            uint64_t c[SPARSE_LDS_N / T];
#pragma unroll
            for (int q = 0; q < SPARSE_LDS_N / T; ++q) c[q] = tid + T * q < n0 ? a[tid + T * q] : 0ull;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < SPARSE_LDS_N / T; ++q) {
                const int i = tid + T * q;
                if (i < n0) {
                    key[i] = (uint32_t)(c[q] >> 16);
                    dd[i] = g.depth_scale / (float)(uint32_t)(c[q] & 0xffffu);  // from_depth, inverse_depth.rs:24-29
                    vv[i] = g.idepth_variance;
                }
            }
        } else {
            for (int i = tid; i < n0; i += T) {
                const uint64_t c = a[i];
                key[i] = (uint32_t)(c >> 16);
                dd[i] = g.depth_scale / (float)(uint32_t)(c & 0xffffu);
                vv[i] = g.idepth_variance;
            }
        }
    }
    __syncthreads();
    int n = n0;
    for (int l = 0; l < g.L; ++l) {
        if (l > 0) {
            const int rows = g.lv[l].rows, cols = g.lv[l].cols;
            int n_new = 0;
            if (tid == 0) *s_prev = 0xffffffffu;
            __syncthreads();
            for (int base = 0; base < n; base += T) {
                const int i = base + tid;
                bool head = false;
                uint32_t pk = 0;
                float fd = 0.f, fv = 0.f;
                const uint32_t prev_chunk_last = *s_prev;  // key >> 2 of the entry before this chunk (overwritten by now, perhaps)
                if (i < n) {
                    pk = key[i] >> 2;
                    const uint32_t before = tid == 0 ? prev_chunk_last : (key[max(i - 1, 0)] >> 2);  // (clamped: the load may be issued for i == 0 too)
                    head = before != pk;
                    if (head) head = (int)morton_compact(pk >> 1) < cols && (int)morton_compact(pk) < rows;
                    if (head) {
                        // strategy_dso_mean over the known children, in [a, b, c, d] order (inverse_depth.rs:81-98)
                        int m = 1;
                        while (m < 4 && i + m < n && (key[i + m] >> 2) == pk) ++m;
                        if (m == 1) {
                            fd = dd[i];
                            fv = vv[i];
                        } else {
                            float sv = vv[i] + vv[i + 1], sd = dd[i] * vv[i] + dd[i + 1] * vv[i + 1];
                            if (m > 2) {
                                sv += vv[i + 2];
                                sd += dd[i + 2] * vv[i + 2];
                            }
                            if (m > 3) {
                                sv += vv[i + 3];
                                sd += dd[i + 3] * vv[i + 3];
                            }
                            fv = sv;
                            fd = sd / sv;
                        }
                    }
                }
                int total;
                const int pos = n_new + block_rank<T>(head, s_wave, &total);  // (its barriers separate this chunk's reads from its writes)
                if (i < n && (tid == T - 1 || i == n - 1)) *s_prev = pk;
                if (head) {
                    key[pos] = pk;
                    dd[pos] = fd;
                    vv[pos] = fv;
                }
                n_new += total;
                __syncthreads();
            }
            n = n_new;
        }
        // records of level l (engine.h SlimRec), appended behind those of the earlier groups of bands
        const int cap = g.lv[l].n_slots, o0 = s_out[l], n_out = max(0, min(n, cap - o0));
        SlimRec* out = rec.S + (size_t)pair * g.slots_total + g.lv[l].slot_off + o0;
        for (int i = tid; i < n_out; i += T) {
            const uint32_t k2 = key[i];
            const int x = (int)morton_compact(k2 >> 1), y = (int)morton_compact(k2);
            int gx, gy, tm;
            grad_tmpl_at(g, kf0, kfu, pair, l, x, y, &gx, &gy, &tm);
            out[i] = SlimRec{(uint32_t)x | ((uint32_t)y << 16), dd[i], slim_pack_tg(tm, gx, gy)};
        }
        __syncthreads();
        if (tid == 0) s_out[l] = o0 + n_out;
        __syncthreads();
    }
}
// Phase A for the whole batch at memory speed: many workgroups per pair, the usable pixels appended to the pair's scratch in any order
// (phase B sorts them), their number in `counts[pair]` (zeroed by the launcher). A pair with more than `cap_n` of them is scanned again,
// band by band, by its own workgroup below.
constexpr int SCAN_U = 4;  // 16-pixel groups per thread of the scan kernel (64 consecutive pixels)
__global__ __launch_bounds__(256) void mask_sparse_scan_kernel(Geom g, const uint16_t* __restrict__ depth, uint8_t* __restrict__ mask, DsoWs ws,
                                                                int from_stamps, PixelPlanes pp, int cap_n) {
    // four 16-pixel groups per thread (a wavefront that does less is bound by its own launch); the workgroup's usable pixels get ONE
    // reservation in the pair's list (a prefix sum over the workgroup, one global atomic); the 16 depths under a group with a pick come
    // in one round trip.
    __shared__ int s_wave[4], s_base;
    const int pair = select_pair(g, blockIdx.y);
    if (pair < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S0 = g.S0, cols0 = g.lv[0].cols;
    const uint8_t* src = (from_stamps ? ws.picked : mask) + (size_t)pair * S0;
    uint8_t* mout = mask ? mask + (size_t)pair * S0 : nullptr;
    const uint16_t* dp = depth + (size_t)pair * S0;
    uint64_t* gsort = reinterpret_cast<uint64_t*>(pp.v + (size_t)pair * pp.stride);
    // 16-byte loads need 16-byte aligned planes: the mask / stamp planes are the handle's own (hipMalloc), the depth map is the caller's
    const bool vec = cols0 % 16 == 0 && S0 % 16 == 0 && reinterpret_cast<uintptr_t>(depth) % 16 == 0;
    // group u of a thread: pixels [t0(u), t0(u) + 16), consecutive lanes on consecutive groups (every load / store instruction is contiguous)
#define SCAN_T0(u) (((blockIdx.x * SCAN_U + (u)) * 256 + tid) * 16)
    DsoState st{};
    if (from_stamps) st = ws.state[pair];
    uint32_t usable[SCAN_U];
    int mine = 0;
    if (vec) {
        uint4 pk[SCAN_U];
#pragma unroll
        for (int u = 0; u < SCAN_U; ++u) pk[u] = (SCAN_T0(u) < S0) ? *reinterpret_cast<const uint4*>(src + SCAN_T0(u)) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < SCAN_U; ++u) {
            const int t0 = SCAN_T0(u);
            const uint4 p4 = pk[u];
            uint32_t bits = 0;
            if ((p4.x | p4.y | p4.z | p4.w) != 0) {
                const uint32_t w[4] = {p4.x, p4.y, p4.z, p4.w};
                if (from_stamps) {
                    bits = dso_final_bits16(st, w, t0, cols0);
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if ((w[k >> 2] >> (8 * (k & 3))) & 0xff) bits |= 1u << k;
                }
            }
            if (from_stamps && mask && t0 < S0) {  // (the mask plane is only kept for the plane path: the sparse form passes nullptr)
                uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 16; ++k) o[k >> 2] |= ((bits >> k) & 1u) << (8 * (k & 3));
                *reinterpret_cast<uint4*>(mout + t0) = make_uint4(o[0], o[1], o[2], o[3]);
            }
            uint32_t us = 0;
            if (bits) {  // the 16 depths under the group in one round trip
                const uint4 d0 = *reinterpret_cast<const uint4*>(dp + t0), d1 = *reinterpret_cast<const uint4*>(dp + t0 + 8);
                const uint32_t dw[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) us |= (((dw[j] & 0xffffu) != 0 ? 1u : 0u) | ((dw[j] >> 16) != 0 ? 2u : 0u)) << (2 * j);
                us &= bits;
            }
            usable[u] = us;
            mine += __popc(us);
        }
    } else {
#pragma unroll
        for (int u = 0; u < SCAN_U; ++u) {
            const int t0 = SCAN_T0(u);
            uint32_t us = 0;
            for (int k = 0; k < 16 && t0 + k < S0; ++k) {
                const int bq = src[t0 + k];
                const bool m = bq && (!from_stamps || dso_final_mask(st, bq, t0 + k, cols0));
                if (m && dp[t0 + k] != 0) us |= 1u << k;
                if (from_stamps && mout) mout[t0 + k] = m ? 1 : 0;
            }
            usable[u] = us;
            mine += __popc(us);
        }
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        before += w < wave ? s_wave[w] : 0;
        total += s_wave[w];
    }
    if (tid == 0) s_base = total ? atomicAdd(pp.counts + (size_t)pair * pp.chunks_total, total) : 0;
    __syncthreads();
    if (mine == 0) return;
    int pos = s_base + before;
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) {
        if (usable[u] == 0) continue;
        const int t0 = SCAN_T0(u), y0 = t0 / cols0, x0 = t0 - y0 * cols0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (!((usable[u] >> k) & 1u)) continue;
            int x = x0 + k, y = y0;
            if (x >= cols0) {  // (only when a 16-pixel group straddles rows: cols % 16 != 0)
                y += x / cols0;
                x = x % cols0;
            }
            if (pos < cap_n) gsort[pos] = ((uint64_t)(morton_part((uint32_t)y) | (morton_part((uint32_t)x) << 1)) << 16) | dp[t0 + k];
            ++pos;
        }
    }
}
#undef SCAN_T0
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(T == 512 ? 6 : 4))) void mask_sparse_records_kernel(Geom g, const uint8_t* __restrict__ kf0, const uint8_t* __restrict__ kfu,
                                                                    const uint16_t* __restrict__ depth, uint8_t* __restrict__ mask, DsoWs ws,
                                                                    int from_stamps, PixelPlanes pp, Records rec, int cap_n, int bitonic) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_set[3 * SPARSE_LDS_N];  // 48 KB: the sort words first (32 KB), then the three arrays in their place
    uint64_t* lds_sort = reinterpret_cast<uint64_t*>(lds_set);
    __shared__ int s_n, s_wave[16], s_out[VORS_MAX_LEVELS];
    __shared__ uint32_t s_prev;
    const int pair = select_pair(g, blockIdx.x);
    if (pair < 0) return;
    const int tid = threadIdx.x;
    const int S0 = g.S0, rows0 = g.lv[0].rows, cols0 = g.lv[0].cols;
    uint64_t* gsort = reinterpret_cast<uint64_t*>(pp.v + (size_t)pair * pp.stride);
    uint32_t* gset = reinterpret_cast<uint32_t*>(pp.iz + (size_t)pair * pp.stride);
    SparseScan q;
    q.src = (from_stamps ? ws.picked : mask) + (size_t)pair * S0;
    q.mout = mask ? mask + (size_t)pair * S0 : nullptr;
    q.dp = depth + (size_t)pair * S0;
    q.gsort = gsort;
    q.st = DsoState{};
    if (from_stamps) q.st = ws.state[pair];
    q.from_stamps = from_stamps;
    q.cols0 = cols0;
    q.cap_n = cap_n;
    q.vec = cols0 % 16 == 0 && S0 % 16 == 0;  // (band starts are then multiples of 16 as well)
    if (tid < VORS_MAX_LEVELS) s_out[tid] = 0;
    // The whole plane in one go when its usable pixels fit the scratch (any selector's mask does). Otherwise in groups of BANDS of whole
    // tile rows (2^(L-1) image rows: no node of any level has children in two bands) — as many bands per group as fit; a band alone always
    // fits (launch rule).
    const int band_rows = 1 << (g.L - 1), n_bands = (rows0 + band_rows - 1) / band_rows;
    bool all_at_once = true;
    int band = 0;
    while (band < n_bands) {
        __syncthreads();
        if (tid == 0) s_n = 0;
        __syncthreads();
        int n0 = 0;
        if (all_at_once) {
            n0 = pp.counts[(size_t)pair * pp.chunks_total];  // mask_sparse_scan_kernel has been over the whole plane
            if (n0 > cap_n) {
                all_at_once = false;
                continue;
            }
            band = n_bands;
        } else {
            const int first = band;
            for (; band < n_bands; ++band) {
                sparse_scan<T>(q, band * band_rows * cols0, min(S0, (band + 1) * band_rows * cols0), &s_n);
                __syncthreads();
                const int n_now = s_n;
                __syncthreads();
                if (n_now > cap_n) break;  // this band does not fit behind the earlier ones: they go first, it is scanned again
                n0 = n_now;
            }
            if (band == first) {  // (a band alone overflowing: excluded by the launch rule; skip it rather than spin)
                ++band;
                continue;
            }
        }
        if (n0 == 0) continue;
        if (n0 <= SPARSE_LDS_N)
            sparse_flush<T>(g, kf0, kfu, pair, gsort, n0, lds_sort, true, lds_set, SPARSE_LDS_N, rec, s_wave, s_out, &s_prev, bitonic != 0);
        else
            sparse_flush<T>(g, kf0, kfu, pair, gsort, n0, gsort, false, gset, cap_n, rec, s_wave, s_out, &s_prev, true);
    }
    __syncthreads();
    if (tid < g.L) rec.n_used[(size_t)pair * VORS_MAX_LEVELS + tid] = s_out[tid];
}

// DSO-style selection + keyframe precompute. When the shape allows it (level 1 exists, cols % 16 == 0, rows even) the mask is
// finalized inside the level-1 inverse-depth pass (one pass over the level-0 planes instead of three).
void launch_keyframe_dso(const Geom& g, Pyramid kf, const uint16_t* depth, DsoWs ws, uint8_t* mask, PixelPlanes pp, Records rec, int n_pairs,
                         hipStream_t s) {
    const bool fused = g.L >= 2 && g.lv[0].cols % 16 == 0 && g.lv[0].rows % 2 == 0;
    // Sparse form whenever the pair's pixel planes (its scratch) hold every pixel of one band of tile rows (then any mask works: denser
    // ones go through in several groups of bands); VORS_DSO_PLANES=1 forces the plane path.
    int cap_n = 1;
    while (2 * cap_n <= pp.stride / 3) cap_n *= 2;
    const char* env_planes = getenv("VORS_DSO_PLANES");  // (read per launch: tests switch forms on one handle)
    const bool force_planes = env_planes && atoi(env_planes) != 0;
    if (!force_planes && cap_n >= (1 << (g.L - 1)) * g.lv[0].cols && g.lv[0].cols < 65536 && g.lv[0].rows < 65536) {  // a band alone fits
        // (from the pick stamps: nobody reads the final mask plane in this form, so it is not written)
        uint8_t* no_mask = nullptr;
        const char* env_scan = getenv("VORS_DSO_SCAN");
        const bool use_scan = env_scan && atoi(env_scan) != 0;
        if (use_scan) {  // the usable picks extracted by a pass over the stamp plane (round 2's form; same lists after the sort)
            launch_dso_selection(g, kf, ws, n_pairs, s);
            launch_zero_ints(g, pp.counts, pp.chunks_total, n_pairs, s);
            hipLaunchKernelGGL(mask_sparse_scan_kernel, dim3((g.S0 + 256 * 16 * SCAN_U - 1) / (256 * 16 * SCAN_U), n_pairs), dim3(256), 0, s, g, depth, no_mask, ws, 1, pp, cap_n);
        } else {         // the selection rounds hand their picks over as a list: the 307 k-pixel plane is not scanned for ~2000 picks
            launch_dso_selection(g, kf, ws, n_pairs, s,
                                 DsoListOut{depth, reinterpret_cast<uint64_t*>(pp.v), (size_t)pp.stride / 2, pp.counts, pp.chunks_total, cap_n});
        }
        // Threads per pair of the records kernel (48 KB of LDS per workgroup): 512 from 512 pairs on — three workgroups per CU instead of
        // two, 1.09 -> 0.60 ms per 4096 pairs; a small batch is faster with 1024 (the shortest chain per pair). Same lists either way.
        const char* e = getenv("VORS_DSO_RECORDS_THREADS");
        const int forced = e ? atoi(e) : 0;
        const int rt = (forced == 512 || forced == 1024) ? forced : (n_pairs >= 512 ? 512 : 1024);
        // VORS_DSO_SORT=bitonic: round 3's sort network in the LDS form instead of the bucket sort (same lists; A/B and tests; read per launch)
        const char* so = getenv("VORS_DSO_SORT");
        const int bitonic = (so && so[0] == 'b') ? 1 : 0;
        if (rt == 512)
            hipLaunchKernelGGL(mask_sparse_records_kernel<512>, dim3(n_pairs), dim3(512), 0, s, g, kf.level0, kf.upper, depth, no_mask, ws, 1, pp, rec, cap_n, bitonic);
        else
            hipLaunchKernelGGL(mask_sparse_records_kernel<1024>, dim3(n_pairs), dim3(1024), 0, s, g, kf.level0, kf.upper, depth, no_mask, ws, 1, pp, rec, cap_n, bitonic);
        return;
    }
    launch_dso_selection(g, kf, ws, n_pairs, s);
    if (!fused) hipLaunchKernelGGL(dso_finalize_kernel, dim3((g.S0 + 4095) / 4096, n_pairs), dim3(256), 0, s, g, ws, mask);
    keyframe_from_mask(g, kf, depth, mask, pp, rec, fused ? &ws : nullptr, n_pairs, s);
}

}  // namespace vors
