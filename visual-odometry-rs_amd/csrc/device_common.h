// Device helpers shared by kernels.hip and lm_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.h"

namespace vors {

#define VORS_INVALID_XY 0xFFFFFFFFu

// Pair addressed by index k of a kernel's pair dimension: k itself, or — masked launch (Geom::sel_list) — the k-th selected pair,
// -1 beyond the selection (the workgroup has nothing to do). Workgroup-uniform.
__device__ __forceinline__ int select_pair(const Geom& g, int k) {
    if (!g.sel_list) return k;
    return k < *g.sel_count ? g.sel_list[k] : -1;
}

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


// Same gradient + the pixel's own grey level from the same round trip: at l >= 1 the level-l pixel IS the floored mean of the 2x2 block
// the block gradient reads (multires.rs:21-31), at l == 0 the centre pixel is one more load issued with the four neighbours.
__device__ __forceinline__ void grad_tmpl_at(const Geom& g, const uint8_t* level0, const uint8_t* upper, int pair, int l, int x,
                                             int y, int* gx, int* gy, int* tm) {
    // (32-bit lane offsets from a wavefront-uniform base: the loads take the scalar-base + vector-offset form, no 64-bit lane arithmetic)
    if (l == 0) {
        const int rows = g.lv[0].rows, cols = g.lv[0].cols;
        const uint8_t* p = level0 + (size_t)pair * g.S0;
        const unsigned o = __umul24((unsigned)y, (unsigned)cols) + (unsigned)x;
        const bool interior = !(x == 0 || y == 0 || x == cols - 1 || y == rows - 1);
        const unsigned dy = interior ? (unsigned)cols : 0u;
#ifdef VORS_GRAD_BYTE_LOADS
        const unsigned dx = interior ? 1u : 0u;
        const int l0 = p[o - dx], r0 = p[o + dx], c0 = p[o];
#else
        // left, centre and right as ONE (unaligned) dword from x - 1 — three gathers per point instead of five. A border pixel (gradient 0:
        // only the centre is needed) reads the dword that holds it without leaving the image.
        const unsigned base = interior ? o - 1u : min(o, (unsigned)g.S0 - 4u);
        uint32_t w;
        __builtin_memcpy(&w, p + base, 4);
        const int c0 = (int)((w >> (8u * (interior ? 1u : o - base))) & 0xffu);
        const int l0 = interior ? (int)(w & 0xffu) : c0, r0 = interior ? (int)((w >> 16) & 0xffu) : c0;
#endif
        const int u0 = p[o - dy], d0 = p[o + dy];
        *tm = c0;
        *gx = (r0 - l0) / 2;  // borders: the taps alias the centre pixel -> 0, like gradient.rs:15-33
        *gy = (d0 - u0) / 2;
    } else {
        const int fc = g.lv[l - 1].cols;
        const uint8_t* pb = level_ptr(g, level0, upper, pair, l - 1);
        const unsigned o = __umul24((unsigned)(2 * y), (unsigned)fc) + (unsigned)(2 * x);  // (24-bit multiply: full rate; v_mul_lo_u32 is quarter rate)
        const uint8_t* p = pb + o;
        uint16_t r0, r1;  // the 2x2 block as two (possibly unaligned) 16-bit loads instead of four byte loads
        __builtin_memcpy(&r0, p, 2);
        __builtin_memcpy(&r1, pb + (o + (unsigned)fc), 2);
        const int a = r0 & 0xff, c = r0 >> 8, b = r1 & 0xff, d = r1 >> 8;
        *gx = (c + d - a - b) / 2;
        *gy = (b - a + d - c) / 2;
        *tm = (a + b + c + d) >> 2;
    }
}

}  // namespace vors
