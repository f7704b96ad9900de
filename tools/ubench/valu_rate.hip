// How many cycles does a SIMD of the MI355X spend per wave64 integer VALU instruction when several waves compete for it?
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate
// One workgroup of 64 x 4 x W threads on one CU = W waves on each of the four SIMDs; every wave runs N iterations of eight
// INDEPENDENT v_add_u32 / v_xor_b32 / v_and_or_b32 chains (so a single wave is limited by issue, not by dependency latency) and
// reports s_memtime ticks.  cycles per instruction per SIMD = ticks * (clock ratio) / (W * instructions per wave).  DESIGN.md
// section 8 uses the result to say what bounds the exact brushfire when the chip is full (6 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND>
__global__ void k(uint32_t* out, uint64_t* ticks, int n)
{
    uint32_t a = threadIdx.x, b = a * 3u, c = a * 5u, d = a * 7u, e = a * 11u, f = a * 13u, g = a * 17u, h = a * 19u;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();      // s_memtime
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) { a += 0x9E37u; b += 0x79B9u; c += 0x7F4Au; d += 0x7C15u; e += 0xF39Cu; f += 0xC0CBu; g += 0xA4EDu; h += 0x1B87u; }
        else { a ^= a >> 7; b ^= b >> 5; c ^= c >> 3; d ^= d >> 9; e ^= e >> 11; f ^= f >> 13; g ^= g >> 6; h ^= h >> 4; }     // 2 VALU each (v_lshrrev + v_xor) unless fused
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}

int main()
{
    const int n = 20000;
    uint32_t* d_out; uint64_t* d_t;
    CK(hipMalloc(&d_out, 1024 * 4 * 4)); CK(hipMalloc(&d_t, 64 * 8));
    int clk = 0, wall = 0;
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    CK(hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0));
    std::printf("# shader clock %d kHz, s_memtime clock %d kHz; n = %d iterations of 8 independent chains\n", clk, wall, n);
    for (int kind = 0; kind < 2; ++kind)
        for (int W : {1, 2, 4}) {
            const int threads = 256 * W;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            if (kind == 0) { k<0><<<1, threads>>>(d_out, d_t, 1000); CK(hipEventRecord(e0)); k<0><<<1, threads>>>(d_out, d_t, n); CK(hipEventRecord(e1)); }
            else { k<1><<<1, threads>>>(d_out, d_t, 1000); CK(hipEventRecord(e0)); k<1><<<1, threads>>>(d_out, d_t, n); CK(hipEventRecord(e1)); }
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> t(4 * W);
            CK(hipMemcpy(t.data(), d_t, 8 * 4 * W, hipMemcpyDeviceToHost));
            uint64_t mx = 0; for (auto v : t) mx = v > mx ? v : mx;
            const double instr_per_wave = (double)n * (kind == 0 ? 8 : 16);
            const double cyc = (double)mx * ((double)clk / (double)wall);            // shader cycles of the slowest wave
            (void)cyc;      // (the counter's rate differs between boxes; the wall clock at the nominal shader clock is what is reported)
            std::printf("%-22s %d wave(s) per SIMD: kernel %.3f ms = %.2f cycles (at the nominal %d kHz) per VALU instruction per SIMD\n",
                        kind == 0 ? "v_add_u32 chains" : "v_lshrrev+v_xor chains", W, ms, (double)ms * 1e-3 * clk * 1e3 / (W * instr_per_wave), clk);
        }
    return 0;
}
