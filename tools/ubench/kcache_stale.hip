// Which read paths can serve a stale copy of a device table that is rewritten between kernel launches?  (MI355X; round 6.)
//
//   hipcc -O3 --offload-arch=gfx950 -o kcache_stale tools/ubench/kcache_stale.hip -lpthread && ./kcache_stale [iterations]
//
// T host threads ("contexts"), each with its own table, streams and staging buffers, repeat:
//     rewrite the table with the iteration number   (hipMemcpyAsync from pinned / from pageable memory, or a writer KERNEL)
//     launch a reader kernel                         (on the stream that carried the write, or on another stream behind an event)
//     compare what the reader saw through three paths with what was written:
//        scalar  : uniform load through a `const __restrict__` kernel argument   -> s_load_dword, scalar data cache (K$) -> L2
//        vector  : plain global_load_dword                                        -> vector L1 (TCP) -> L2
//        agent   : __hip_atomic_load(relaxed, agent scope)                        -> coherent at the L2 (sc1)
// A path that ever returns the previous iteration's value is not safe for host-rewritten tables.  The product's rule after this
// experiment is in DESIGN.md section 8 ("scalar-cache hazard"); tools/check_scalar_loads.py enforces it on the ISA.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::abort(); } } while (0)

// The table must FIT the scalar data cache (16 KB, shared by a few CUs) or capacity evictions hide everything: the first version of
// this experiment read 3000 records (240 KB) per launch and never saw a stale line.  LAMA_KC_BLK=3000 restores that.
#ifndef LAMA_KC_BLK
#define LAMA_KC_BLK 96
#endif
constexpr int N_BLK = LAMA_KC_BLK;      // one block per "particle", like the product's per-particle launches
constexpr int N_ENT = N_BLK * 20;       // words: a PartRec is 80 B

__global__ __launch_bounds__(64) void k_reader(const uint32_t* __restrict__ tab, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 20u;                     // wave-uniform: the compiler emits s_load_dword for tab[i]
    const uint32_t s = tab[i];
    uint32_t iv = i + 1u;
    asm volatile("" : "+v"(iv));                             // a VGPR address: global_load_dword
    const uint32_t v = tab[iv];
    uint32_t ia = i + 2u;
    asm volatile("" : "+v"(ia));
    const uint32_t a = __hip_atomic_load(tab + ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = s; out[3 * blockIdx.x + 1] = v; out[3 * blockIdx.x + 2] = a; }
}
__global__ __launch_bounds__(256) void k_writer(uint32_t* __restrict__ tab, uint32_t value, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) tab[i] = value;
}
__global__ __launch_bounds__(256) void k_busy(uint32_t* __restrict__ x, int n, int rounds)        // something else that keeps the chip occupied
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t v = x[i];
    for (int r = 0; r < rounds; ++r) v = v * 1664525u + 1013904223u;
    x[i] = v;
}

enum Writer { W_PINNED, W_PAGEABLE, W_KERNEL };
struct Cfg { int threads; Writer writer; bool other_stream; bool busy; };
struct Res { uint64_t stale[3] = {0, 0, 0}; uint64_t reads = 0; };

static void worker(int tid, const Cfg cfg, int iters, Res* res)
{
    CK(hipSetDevice(0));
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    uint32_t *d_tab, *d_out, *d_busy, *h_pin, *h_out;
    CK(hipMalloc(&d_tab, N_ENT * 4)); CK(hipMalloc(&d_out, N_BLK * 12)); CK(hipMalloc(&d_busy, 1 << 20));
    CK(hipMemset(d_busy, 0, 1 << 20));
    CK(hipHostMalloc(&h_pin, N_ENT * 4)); CK(hipHostMalloc(&h_out, N_BLK * 12));
    std::vector<uint32_t> h_page(N_ENT);
    for (int k = 1; k <= iters; ++k) {
        const uint32_t value = ((uint32_t)tid << 24) | (uint32_t)k;
        if (cfg.writer == W_KERNEL) {
            k_writer<<<(N_ENT + 255) / 256, 256, 0, sa>>>(d_tab, value, N_ENT);
        } else {
            uint32_t* src = cfg.writer == W_PINNED ? h_pin : h_page.data();
            for (int i = 0; i < N_ENT; ++i) src[i] = value;
            CK(hipMemcpyAsync(d_tab, src, N_ENT * 4, hipMemcpyHostToDevice, sa));
        }
        hipStream_t sr = sa;
        if (cfg.other_stream) { CK(hipEventRecord(ev, sa)); CK(hipStreamWaitEvent(sb, ev, 0)); sr = sb; }
        if (cfg.busy) k_busy<<<1024, 256, 0, sr>>>(d_busy, 1 << 18, 64);
        k_reader<<<N_BLK, 64, 0, sr>>>(d_tab, d_out);
        CK(hipMemcpyAsync(h_out, d_out, N_BLK * 12, hipMemcpyDeviceToHost, sr));
        CK(hipStreamSynchronize(sr));
        if (cfg.other_stream) CK(hipStreamSynchronize(sa));
        for (int b = 0; b < N_BLK; ++b)
            for (int j = 0; j < 3; ++j) res->stale[j] += h_out[3 * b + j] != value;
        res->reads += N_BLK;
    }
    CK(hipFree(d_tab)); CK(hipFree(d_out)); CK(hipFree(d_busy)); CK(hipHostFree(h_pin)); CK(hipHostFree(h_out));
    CK(hipEventDestroy(ev)); CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 400;
    const char* wname[] = {"hipMemcpyAsync(pinned)", "hipMemcpyAsync(pageable)", "writer kernel"};
    const char* q = std::getenv("GPU_MAX_HW_QUEUES");
    std::printf("# iterations per thread %d, table %d B, reader blocks %d, GPU_MAX_HW_QUEUES=%s\n", iters, N_ENT * 4, N_BLK, q ? q : "(default)");
    std::printf("# %-26s %-12s %-5s %-7s | stale scalar / vector / agent-scope reads of %s\n", "table written by", "reader on", "busy", "threads", "total");
    uint64_t any_agent = 0;
    for (int threads : {1, 8})
        for (Writer w : {W_PINNED, W_PAGEABLE, W_KERNEL})
            for (bool other : {false, true})
                for (bool busy : {false, true}) {
                    if (busy && threads == 1) continue;
                    Cfg cfg{threads, w, other, busy};
                    std::vector<Res> res(threads);
                    std::vector<std::thread> th;
                    for (int t = 0; t < threads; ++t) th.emplace_back(worker, t, cfg, iters, &res[t]);
                    for (auto& t : th) t.join();
                    Res sum;
                    for (auto& r : res) { for (int j = 0; j < 3; ++j) sum.stale[j] += r.stale[j]; sum.reads += r.reads; }
                    any_agent += sum.stale[2];
                    std::printf("  %-26s %-12s %-5s %-7d | %llu / %llu / %llu of %llu\n", wname[w], other ? "other stream" : "same stream", busy ? "yes" : "no", threads,
                                (unsigned long long)sum.stale[0], (unsigned long long)sum.stale[1], (unsigned long long)sum.stale[2], (unsigned long long)sum.reads);
                    std::fflush(stdout);
                }
    return any_agent ? 1 : 0;
}
