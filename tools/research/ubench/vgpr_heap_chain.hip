// research micro-benchmark: cost of a DEPENDENT chain of accesses to an array held in VGPRs (dynamic uniform register index +
// readlane / writelane), one wave -- the access pattern a register-resident binary heap would have -- against the same chain through LDS.
//   hipcc -O3 --offload-arch=gfx950 tools/research/ubench/vgpr_heap_chain.hip -o gpurun_out/vgpr_heap_chain && gpurun_out/vgpr_heap_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k_vgpr(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n)
{
    uint32_t h[16];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = in[r * 64 + lane];
    uint32_t acc = 0;
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)in[1024]);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        const uint32_t reg = s >> 6, ln = s & 63u;
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)h[reg], (int)ln);
        acc += v;
        const uint32_t s2 = (v * 2654435761u) >> 22;
        const uint32_t reg2 = s2 >> 6, ln2 = s2 & 63u;
        { uint32_t o = h[reg2]; asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(o) : "s"(v + 1u), "s"(ln2) : "m0"); h[reg2] = o; }
        s = s2;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r * 64 + lane] = h[r];
    if (lane == 0) { out[1024] = acc; out[1025] = (uint32_t)(t1 - t0); out[1026] = (uint32_t)((t1 - t0) >> 32); }
}
__global__ void k_lds(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n)
{
    __shared__ uint32_t h[1024];
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < 16; ++r) h[r * 64 + lane] = in[r * 64 + lane];
    __syncthreads();
    uint32_t acc = 0;
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)in[1024]);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)h[s]);
        acc += v;
        const uint32_t s2 = (v * 2654435761u) >> 22;
        if (lane == 0) h[s2] = v + 1u;
        s = s2;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    __syncthreads();
    for (int r = 0; r < 16; ++r) out[r * 64 + lane] = h[r * 64 + lane];
    if (lane == 0) { out[1024] = acc; out[1025] = (uint32_t)(t1 - t0); out[1026] = (uint32_t)((t1 - t0) >> 32); }
}
int main()
{
    const int n = 200000;
    std::vector<uint32_t> in(1025);
    for (int i = 0; i < 1024; ++i) in[i] = (uint32_t)i * 2246822519u + 12345u;
    in[1024] = 17;
    uint32_t *d_in, *d_out;
    hipMalloc(&d_in, 1025 * 4); hipMalloc(&d_out, 1027 * 4);
    hipMemcpy(d_in, in.data(), 1025 * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> o1(1027), o2(1027);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_vgpr, dim3(1), dim3(64), 0, 0, d_in, d_out, n); hipDeviceSynchronize();
        hipMemcpy(o1.data(), d_out, 1027 * 4, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, d_in, d_out, n); hipDeviceSynchronize();
        hipMemcpy(o2.data(), d_out, 1027 * 4, hipMemcpyDeviceToHost);
    }
    const double c1 = (double)(((uint64_t)o1[1026] << 32) | o1[1025]) / n, c2 = (double)(((uint64_t)o2[1026] << 32) | o2[1025]) / n;
    bool same = true; for (int i = 0; i <= 1024; ++i) same = same && o1[i] == o2[i];
    printf("dependent read+write chain, one wave: VGPR-resident array %.1f clock ticks / iteration, LDS array %.1f; results %s\n", c1, c2, same ? "equal" : "DIFFER");
    printf("(s_memtime / readcyclecounter ticks; compare the two numbers with each other)\n");
    return 0;
}
