// Developer probe: shader clock (s_memtime) against the constant 100 MHz counter (s_memrealtime) for a latency-chain kernel
// at different numbers of resident workgroups.  hipcc --offload-arch=gfx950 -O3 clk_probe.hip -o clk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(128) void k_chain(uint64_t* out, int iters, int mode)
{
    __shared__ uint32_t sh[2048];
    uint32_t x = threadIdx.x + blockIdx.x;
    sh[threadIdx.x] = x; sh[threadIdx.x + 128] = x * 3;
    __syncthreads();
    const uint64_t c0 = __builtin_readcyclecounter();
    uint64_t r0; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r0));
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        if (mode >= 1) { x += sh[(x >> 8) & 2047]; }
        if (mode >= 2) { __syncthreads(); }
        x ^= x >> 7;
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    uint64_t r1; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r1));
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = c1 - c0; out[3 * blockIdx.x + 1] = r1 - r0; out[3 * blockIdx.x + 2] = x; }
}
int main()
{
    uint64_t* d; hipMalloc(&d, 8 * 3 * 8192);
    std::vector<uint64_t> h(3 * 8192);
    for (int mode = 0; mode < 3; ++mode)
        for (int wgs : {30, 300, 1000, 3000, 6000}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipEventRecord(a);
                hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(128), 0, 0, d, 200000, mode);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                hipMemcpy(h.data(), d, 8 * 3 * wgs, hipMemcpyDeviceToHost);
                double cyc = 0, rt = 0;
                for (int i = 0; i < wgs; ++i) { cyc += h[3 * i]; rt += h[3 * i + 1]; }
                cyc /= wgs; rt /= wgs;
                if (rep == 2) printf("mode %d wgs %5d: %.3f ms  shader cycles %.0f  realtime ticks %.0f (%.3f ms @100MHz)  => %.2f GHz\n", mode, wgs, ms, cyc, rt, rt / 1e5, cyc / (rt * 10.0) / 1e0 / 1000 * 1000 / 1000);
            }
        }
    return 0;
}
