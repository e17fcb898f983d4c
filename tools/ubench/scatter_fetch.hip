// Micro-benchmark (development aid): what does ONE scattered 2-byte load cost in HBM traffic AS THE COUNTERS SEE IT, and how many of them per
// second does the chip sustain? The LM stage's bilinear taps are such loads (two per point and evaluation), and MI355X_MICROARCH.md calibrates
// FETCH_SIZE (x 2) for wide streaming reads only. Every lane reads 2 bytes at a pseudo-random address of a buffer far larger than L2 + MALL
// (no reuse by construction); rocprofv3 --pmc FETCH_SIZE over this binary / loads issued = raw counter bytes per load.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/scatter_fetch.hip -o tools/ubench/scatter_fetch
//   (gpurun)  tools/ubench/scatter_fetch ; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/sf -o b -- tools/ubench/scatter_fetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define LOADS_PER_LANE 256
__global__ __launch_bounds__(256) void scatter_kernel(const uint8_t* __restrict__ buf, size_t bytes, uint32_t* out, int inflight) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = gid * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    uint32_t s = 0;
    for (int it = 0; it < LOADS_PER_LANE; it += 8) {
        uint16_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // eight independent requests in flight per lane
            x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
            const size_t off = ((x * 0x2545F4914F6CDD1Dull) % (bytes - 2)) & ~(size_t)1;
            __builtin_memcpy(&v[k], buf + off, 2);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    if (s == 0x7fffffffu) out[0] = s;
    (void)inflight;
}
int main() {
    const size_t bytes = (size_t)8 << 30;  // 8 GiB
    uint8_t* buf;
    uint32_t* out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    const int blocks = 256 * 64;  // 4.2 M lanes x 256 loads = 1.07 G loads
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(256), 0, 0, buf, bytes, out, 8);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double loads = (double)blocks * 256 * LOADS_PER_LANE;
        printf("scattered 2-byte loads: %.3g loads in %.3f ms = %.3g loads/s (%.2f TB/s at 32 B per load, %.2f at 64 B)\n", loads, ms, loads / ms * 1e3,
               loads * 32 / ms * 1e3 / 1e12, loads * 64 / ms * 1e3 / 1e12);
    }
    return 0;
}
