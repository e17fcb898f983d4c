// Micro-benchmark (development aid): how many cycles of a CU's texture path does ONE 64-lane load instruction take, by ADDRESS PATTERN?
// The one-wavefront-per-pair LM kernel issues, per 128 points, 4 two-byte tap gathers (each lane its own image row: up to 64 distinct
// cache lines per instruction) and 6 coalesced record loads; 16 wavefronts share a CU's one texture path. Patterns, all on per-wavefront
// images (level 0: 640 x 480 = 300 KB each: beyond vL1D and L2 once 4096 of them are live; or a 1200-byte level-4 image):
//   0 same address in every lane              1 lanes 2 bytes apart (one line)         2 one column, consecutive rows (stride 640: c2f-like runs)
//   3 pseudo-random pixels (DSO-like)         4 stride 40 in a 1200-byte image (level 4) 5 16 contiguous bytes per lane as ONE dwordx4 load
//   6 pattern 3 as a dword load (does the element size matter?)   7-9 pattern 4 at even / odd / 3-mod-4 addresses (does alignment matter?)
// Every wavefront: ITERS iterations of 4 independent loads; reports cycles per load instruction per CU (16 wavefronts resident per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_patterns.hip -o tools/ubench/gather_patterns   (run through gpurun)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define ITERS 512
#define IMG (640 * 480)

template <int PAT>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ imgs, uint32_t* out, unsigned long long* cycles) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint8_t* img = imgs + (size_t)wave * IMG;
    uint32_t s = 0;
    uint32_t rng = wave * 2654435761u + lane * 40503u + 12345u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        uint32_t v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = it * 4 + g;
            unsigned off;
            if (PAT == 0) off = (unsigned)(q * 977) % (IMG - 64);
            else if (PAT == 1) off = (unsigned)(q * 977) % (IMG - 256) + 2 * lane;
            else if (PAT == 2) off = ((unsigned)(q * 13) % 600) + (unsigned)(((q * 7) % 400 + lane) * 640);
            else if (PAT == 3 || PAT == 6) {
                rng = rng * 1664525u + 1013904223u;
                off = (rng >> 8) % (IMG - 64);
            } else if (PAT == 4) off = ((unsigned)(lane * 40 + q * 3)) % 1198;
            else if (PAT == 7) off = (((unsigned)(lane * 40 + q * 3)) % 1196) & ~1u;        // pattern 4, even addresses
            else if (PAT == 8) off = ((((unsigned)(lane * 40 + q * 3)) % 1196) & ~1u) | 1u;  // odd (a 2-byte load across a 2-byte boundary)
            else if (PAT == 9) off = ((((unsigned)(lane * 40 + q * 3)) % 1192) & ~3u) | 3u;  // across a dword boundary
            else off = (unsigned)((q * 1024 + lane * 16) % (IMG - 64));
            if (PAT == 5) {
                uint4 w;
                __builtin_memcpy(&w, img + off, 16);
                v[g] = w.x + w.y + w.z + w.w;
            } else if (PAT == 6) {
                uint32_t w;
                __builtin_memcpy(&w, img + (off & ~3u), 4);
                v[g] = w;
            } else {
                uint16_t w;
                __builtin_memcpy(&w, img + off, 2);
                v[g] = w;
            }
        }
        s += v[0] + v[1] + v[2] + v[3];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (s == 0x7fffffffu) out[0] = s;
    if (lane == 0) atomicAdd(cycles, t1 - t0);
}

template <int PAT>
static void run(const char* name, const uint8_t* imgs, uint32_t* out, unsigned long long* cyc, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    unsigned long long c = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(cyc, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, imgs, out, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
    }
    const double waves = blocks * 4.0, instr_per_wave = ITERS * 4.0;
    const double cyc_per_wave = (double)c / waves;
    // 16 wavefronts per CU share the texture path: cycles of the path per instruction = wavefront cycles per instruction / 16
    printf("%-44s %8.3f ms  %7.1f cycles per load per wavefront = %6.1f per load per CU  (%.3g loads/s chip-wide)\n", name, best, cyc_per_wave / instr_per_wave,
           cyc_per_wave / instr_per_wave / 16.0, waves * instr_per_wave * 64 / best * 1e3);
}

int main() {
    const int blocks = 1024;  // 4096 wavefronts = 16 per CU, each its own 300 KB image (1.26 GB)
    uint8_t* imgs;
    uint32_t* out;
    unsigned long long* cyc;
    hipMalloc(&imgs, (size_t)blocks * 4 * IMG);
    hipMalloc(&out, 4);
    hipMalloc(&cyc, 8);
    hipMemset(imgs, 1, (size_t)blocks * 4 * IMG);
    run<0>("0 one address", imgs, out, cyc, blocks);
    run<1>("1 one line (lanes 2 B apart)", imgs, out, cyc, blocks);
    run<2>("2 one column, 64 rows (stride 640)", imgs, out, cyc, blocks);
    run<3>("3 pseudo-random pixels of a 300 KB image", imgs, out, cyc, blocks);
    run<4>("4 stride 40 in 1200 bytes", imgs, out, cyc, blocks);
    run<5>("5 dwordx4, lanes contiguous", imgs, out, cyc, blocks);
    run<6>("6 pattern 3 as dword loads", imgs, out, cyc, blocks);
    run<7>("7 pattern 4, even addresses", imgs, out, cyc, blocks);
    run<8>("8 pattern 4, odd addresses", imgs, out, cyc, blocks);
    run<9>("9 pattern 4, addresses = 3 mod 4", imgs, out, cyc, blocks);
    return 0;
}
