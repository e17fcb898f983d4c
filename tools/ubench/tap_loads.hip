// Micro-benchmark (development aid): what do the bilinear TAP loads of a dense quad cost per wavefront instruction on gfx950, by form?
// A lane owns 4 horizontally adjacent pixels whose warped positions are (about) 4 adjacent pixels of the current image: today 8 16-bit
// gathers (two per point), alternatives: the two rows of the quad as ONE wide load each (unaligned 8 bytes; 12 / 16 bytes from the aligned
// dword below), and the same gathers with most lanes masked off (does a gather's cost follow the number of active lanes?).
// Image-like addressing: lane l of a wavefront reads around x = 4 l + jitter of row r(it, wave) of a 640-wide L2-resident image.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/tap_loads.hip -o /tmp/tap_loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
#define COLS 640
#define ROWS 480

template <int FORM>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ img, uint32_t* out, int active_mod) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    uint32_t s = 0;
    const bool active = (lane % active_mod) == 0;
    for (int it = 0; it < ITERS; ++it) {
        const int row = (it * 7 + wave * 3) % (ROWS - 2);
        const int x = ((lane * 4 + (it & 3) + ((wave & 1) ? 256 : 0)) % (COLS - 16));
        const unsigned base = (unsigned)(row * COLS + x);
        if (FORM == 0 || FORM == 5) {  // 8 x 16-bit gathers
            if (FORM == 0 || active) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint16_t a, b;
                    __builtin_memcpy(&a, img + base + g, 2);
                    __builtin_memcpy(&b, img + base + COLS + g, 2);
                    s += a + b;
                }
            }
        } else if (FORM == 1) {  // 2 x 8 unaligned bytes
            uint64_t a, b;
            __builtin_memcpy(&a, img + base, 8);
            __builtin_memcpy(&b, img + base + COLS, 8);
            s += (uint32_t)a + (uint32_t)(a >> 32) + (uint32_t)b + (uint32_t)(b >> 32);
        } else if (FORM == 2) {  // 6 aligned dwords
            const unsigned b0 = base & ~3u;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) s += *reinterpret_cast<const uint32_t*>(img + b0 + r * COLS + 4 * j);
        } else if (FORM == 3) {  // 2 x 12 bytes from the aligned dword below (dwordx3, 4-byte aligned)
            const unsigned b0 = base & ~3u;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                struct __attribute__((packed, aligned(4))) W3 { uint32_t a, b, c; };
                const W3 w = *reinterpret_cast<const W3*>(img + b0 + r * COLS);
                s += w.a + w.b + w.c;
            }
        } else if (FORM == 4) {  // 2 x 16 bytes, 4-byte aligned
            const unsigned b0 = base & ~3u;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                struct __attribute__((packed, aligned(4))) W4 { uint32_t a, b, c, d; };
                const W4 w = *reinterpret_cast<const W4*>(img + b0 + r * COLS);
                s += w.a + w.b + w.c + w.d;
            }
        } else if (FORM == 6) {  // 2 x 8 unaligned bytes + the same again for the lanes of `active` (a third / fourth row for some lanes)
            uint64_t a, b;
            __builtin_memcpy(&a, img + base, 8);
            __builtin_memcpy(&b, img + base + COLS, 8);
            s += (uint32_t)a + (uint32_t)(a >> 32) + (uint32_t)b + (uint32_t)(b >> 32);
            if (active) {
                uint64_t c;
                __builtin_memcpy(&c, img + base + 2 * COLS, 8);
                s += (uint32_t)c + (uint32_t)(c >> 32);
            }
        } else if (FORM == 7) {  // 4 x 32-bit unaligned (two points per load)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                uint32_t a, b;
                __builtin_memcpy(&a, img + base + g, 4);
                __builtin_memcpy(&b, img + base + COLS + g, 4);
                s += a + b;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FORM>
void run(const char* name, int n_loads, const uint8_t* img, uint32_t* out, int waves_per_simd, int active_mod = 1) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, img, out, active_mod);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, img, out, active_mod); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double quads_per_cu = (double)waves_per_simd * 4 * ITERS;  // wavefront-iterations per CU
    const double cyc = ms * 1e-3 * 2.4e9 / quads_per_cu;
    printf("%-46s waves/SIMD %d active 1/%d: %.3f ms  %.1f CU-cycles per wavefront-quad (%.1f per load instruction)\n", name, waves_per_simd, active_mod, ms, cyc, cyc / n_loads);
}

int main() {
    uint8_t* img; (void)hipMalloc(&img, COLS * ROWS + 64); (void)hipMemset(img, 7, COLS * ROWS + 64);
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {2, 8}) {
        run<0>("8 x 16-bit gather (today)", 8, img, out, w);
        run<7>("4 x 32-bit unaligned", 4, img, out, w);
        run<1>("2 x 8 unaligned bytes", 2, img, out, w);
        run<2>("6 x aligned dword", 6, img, out, w);
        run<3>("2 x 12 bytes, dword aligned", 2, img, out, w);
        run<4>("2 x 16 bytes, dword aligned", 2, img, out, w);
        run<5>("8 x 16-bit gather, masked", 8, img, out, w, 4);
        run<5>("8 x 16-bit gather, masked", 8, img, out, w, 16);
        run<5>("8 x 16-bit gather, masked", 8, img, out, w, 64);
        run<6>("2 x 8 unaligned + a third row for some lanes", 3, img, out, w, 16);
        run<6>("2 x 8 unaligned + a third row for some lanes", 3, img, out, w, 4);
    }
    return 0;
}
