// Developer micro-benchmark (round 4): what does ONE wave on a dependent chain pay per instruction kind on gfx950?
// The exact brushfire is a serial chain per particle; this table is the cost model its instruction stream is tuned
// against.  Every test is a single 64-lane wave (or a 2-wave workgroup for the hand-over tests) timed with s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/issue_costs.hip -o gpurun_out/issue_costs && gpurun_out/issue_costs
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>

#define CLK() __builtin_readcyclecounter()
#define NREP 64          // outer repetitions of every block
#define STR2(x) #x
#define STR(x) STR2(x)

struct Res { uint64_t cyc[64]; };

__global__ __launch_bounds__(64) void k_single(Res* out, const uint32_t* __restrict__ gro, uint32_t* grw, int n_chase)
{
    __shared__ uint32_t lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = (uint32_t)((i * 1103515245u + 12345u) >> 8) & 4095u;
    __syncthreads();
    int k = 0;
    uint64_t t0, t1;
    uint32_t x = (uint32_t)lane, y = 1, z = 2, w = 3;
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)gro[0]);
#define BEGIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t0 = CLK();
#define END() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t1 = CLK(); if (lane == 0) out->cyc[k] = t1 - t0; ++k;

    // 0: empty (timer overhead)
    BEGIN(); END();
    // 1: 256 dependent v_add_u32
    BEGIN(); asm volatile(".rept 256\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x)); END();
    // 2: 256 independent-ish v_add_u32 (4 chains)
    BEGIN(); asm volatile(".rept 64\n v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n .endr" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); END();
    // 3: 256 dependent s_add_u32
    BEGIN(); asm volatile(".rept 256\n s_add_u32 %0, %0, 1\n .endr" : "+s"(s) :: "scc"); END();
    // 4: 64 x (v_readfirstlane -> s_add -> v_mov)   VALU->SALU->VALU ping-pong
    BEGIN(); asm volatile(".rept 64\n v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 1\n v_mov_b32 %0, %1\n .endr" : "+v"(x), "+s"(s) :: "scc"); END();
    // 5: 64 x (v_cmp -> vcc ; s_cbranch_vccz NOT taken)
    BEGIN(); asm volatile(".rept 64\n v_cmp_eq_u32 vcc, %0, %0\n s_cbranch_vccz 9f\n .endr\n 9:" :: "v"(x) : "vcc"); END();
    // 6: 64 x (v_cmp -> vcc ; s_cbranch_vccnz TAKEN over one instruction)
    BEGIN(); asm volatile(".rept 64\n v_cmp_eq_u32 vcc, %0, %0\n s_cbranch_vccnz 1f\n s_nop 0\n 1:\n .endr" :: "v"(x) : "vcc"); END();
    // 7: 64 x (s_cmp ; s_cbranch_scc1 NOT taken)
    BEGIN(); asm volatile(".rept 64\n s_cmp_eq_u32 %0, 0x7fffffff\n s_cbranch_scc1 9f\n .endr\n 9:" :: "s"(s) : "scc"); END();
    // 8: 64 x (s_cmp ; s_cbranch_scc1 TAKEN over one instruction)
    BEGIN(); asm volatile(".rept 64\n s_cmp_lg_u32 %0, 0x7fffffff\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n .endr" :: "s"(s) : "scc"); END();
    // 9: 64 x exec-masked region that is skipped (saveexec ; cbranch_execz TAKEN ; restore)
    BEGIN(); asm volatile(".rept 64\n v_cmp_ne_u32 vcc, %0, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, 1\n 1:\n s_or_b64 exec, exec, s[20:21]\n .endr" : "+v"(x) :: "vcc", "s20", "s21"); END();
    // 10: 64 x exec-masked region that is entered by all lanes (branch NOT taken)
    BEGIN(); asm volatile(".rept 64\n v_cmp_eq_u32 vcc, %0, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, 1\n 1:\n s_or_b64 exec, exec, s[20:21]\n .endr" : "+v"(x) :: "vcc", "s20", "s21"); END();
    // 11: the same work branch-free: v_cmp + v_cndmask
    BEGIN(); asm volatile(".rept 64\n v_cmp_eq_u32 vcc, %0, %0\n v_add_u32 %1, %0, 1\n v_cndmask_b32 %0, %0, %1, vcc\n .endr" : "+v"(x), "+v"(y) :: "vcc"); END();
    // 12: 64 x s_branch (unconditional, taken) over one instruction
    BEGIN(); asm volatile(".rept 64\n s_branch 1f\n s_nop 0\n 1:\n .endr"); END();
    // 13: 64 x s_waitcnt with nothing outstanding
    BEGIN(); asm volatile(".rept 64\n s_waitcnt vmcnt(0) lgkmcnt(0)\n .endr"); END();
    // 14: LDS pointer chase, 64 dependent ds_read_b32 (all lanes the same address)
    { uint32_t p = 5; BEGIN(); for (int i = 0; i < 64; ++i) { p = lds[p]; asm volatile("" : "+v"(p)); } END(); x += p; }
    // 15: LDS chase through the scalar side: ds_read -> readfirstlane -> address
    { uint32_t p = 5; BEGIN(); for (int i = 0; i < 64; ++i) { p = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[p]); } END(); x += p; }
    // 16: ds_write then ds_read of the same address, 64 times (store -> load turn-around)
    { uint32_t p = 7; BEGIN(); for (int i = 0; i < 64; ++i) { lds[(p & 63u) + 64 * 8] = p + 1; asm volatile("" ::: "memory"); p = lds[(p & 63u) + 64 * 8]; asm volatile("" : "+v"(p)); } END(); x += p; }
    // 17: ds_bpermute chain (64)
    { uint32_t p = (uint32_t)lane; BEGIN(); for (int i = 0; i < 64; ++i) { p = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((p * 4u + 4u) & 255u), (int)p); } END(); x += p; }
    // 18: DPP row_shr chain (64)
    BEGIN(); asm volatile(".rept 64\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n .endr" : "+v"(x)); END();
    // 19: readlane with a dependent scalar lane index (64): idx = readlane(v, idx) & 63
    { uint32_t idx = 3; BEGIN(); for (int i = 0; i < 64; ++i) { idx = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)idx) & 63u; } END(); s += idx; }
    // 20: ballot -> ffs -> readlane chain (64)
    { uint32_t v = x; BEGIN(); for (int i = 0; i < 64; ++i) { const unsigned long long m = __ballot(v & 1u) | 0x8000000000000000ull; const int l = __ffsll((long long)m) - 1; v += (uint32_t)__builtin_amdgcn_readlane((int)v, l); asm volatile("" : "+v"(v)); } END(); x += v; }
    // 21: global pointer chase, small ring (L1/TCP resident), 64 dependent loads, uniform address in a VGPR
    { uint32_t p = 0; BEGIN(); for (int i = 0; i < 64; ++i) { p = gro[64 + (p & 63u)]; asm volatile("" : "+v"(p)); } END(); x += p; }
    // 22: global pointer chase over 8 MB (L2 / MALL), 64 dependent loads
    { uint32_t p = 1; BEGIN(); for (int i = 0; i < 64; ++i) { p = gro[4096 + (p % (uint32_t)n_chase)]; asm volatile("" : "+v"(p)); } END(); x += p; }
    // 23: scalar-load chase (s_load through the scalar cache), small ring
    { uint32_t p = 0; BEGIN(); for (int i = 0; i < 64; ++i) { p = (uint32_t)__builtin_amdgcn_readfirstlane((int)p); p = gro[64 + (p & 63u)]; } END(); x += p; }
    // 24: store then load back the same global address (wave-private), 32 times
    { uint32_t p = 3; BEGIN(); for (int i = 0; i < 32; ++i) { grw[64 + lane] = p + 1; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); p = __builtin_nontemporal_load(&grw[64 + lane]); asm volatile("" : "+v"(p)); } END(); x += p; }
    // 24b follows as 25: plain load after store (may hit TCP)
    { uint32_t p = 3; BEGIN(); for (int i = 0; i < 32; ++i) { ((volatile uint32_t*)grw)[128 + lane] = p + 1; p = ((volatile uint32_t*)grw)[128 + lane]; } END(); x += p; }
    // 26: 64 global stores issued back to back, then one wait
    BEGIN(); for (int i = 0; i < 64; ++i) ((volatile uint32_t*)grw)[256 + lane] = x + i; END();
    // 27: 64 global atomic-or (no return), then one wait
    BEGIN(); for (int i = 0; i < 64; ++i) atomicOr(&grw[512 + (lane & 3)], 1u << (i & 31)); END();
    // 28: 64 x s_memtime back to back (timer cost itself)
    { uint64_t a = 0; BEGIN(); for (int i = 0; i < 64; ++i) a += CLK(); END(); x += (uint32_t)a; }
    // 29: 64 x (v_readlane fixed lane -> s_cmp -> s_cselect -> v_mov): a typical "uniform decision" round trip
    BEGIN(); asm volatile(".rept 64\n v_readlane_b32 %1, %0, 4\n s_cmp_eq_u32 %1, 0\n s_cselect_b32 %1, 1, 2\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x), "+s"(s) :: "scc"); END();
    // 30: 64 x v_cmp -> s_and_b64 with exec -> s_cmp on the pair (ballot then scalar test) -> v
    BEGIN(); asm volatile(".rept 64\n v_cmp_gt_u32 vcc, %0, 0\n s_and_b64 s[20:21], vcc, exec\n s_cselect_b32 %1, 1, 2\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x), "+s"(s) :: "vcc", "scc", "s20", "s21"); END();
    // 31: 256 dependent v_mad_u64_u32-free integer multiply chain v_mul_lo_u32
    BEGIN(); asm volatile(".rept 64\n v_mul_lo_u32 %0, %0, %0\n .endr" : "+v"(x)); END();
    // 32: 64 dependent v_mul_i32_i24
    BEGIN(); asm volatile(".rept 64\n v_mul_i32_i24 %0, %0, %0\n .endr" : "+v"(x)); END();
    // 33: 64 dependent 64-bit shifts (v_lshlrev_b64)
    { uint64_t q = x; BEGIN(); asm volatile(".rept 64\n v_lshlrev_b64 %0, 1, %0\n .endr" : "+v"(q)); END(); x += (uint32_t)q; }
    // 34: ds_read2_b64 dependent chain (the heap's pair load)
    { uint32_t p = 8; uint64_t* l64 = (uint64_t*)lds; BEGIN(); for (int i = 0; i < 64; ++i) { const uint64_t a = l64[(p & 1022u)], b = l64[(p & 1022u) + 1]; p = (uint32_t)(a + b); asm volatile("" : "+v"(p)); } END(); x += p; }

    // ---- round 4 additions: the sequences the brushfire's straight-line pop is made of
    // 35: 64 x (v_readlane -> VALU reads that SGPR)
    BEGIN(); asm volatile(".rept 64\n v_readlane_b32 %1, %0, 4\n s_nop 0\n v_add_u32 %0, %1, %0\n .endr" : "+v"(x), "+s"(s)); END();
    // 36: 64 x (s_add -> VALU reads that SGPR)
    BEGIN(); asm volatile(".rept 64\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %1, %0\n .endr" : "+v"(x), "+s"(s) :: "scc"); END();
    // 37: 64 x (v_sub ; v_ashrrev 31 ; v_and)  mask arithmetic, dependent
    BEGIN(); asm volatile(".rept 64\n v_sub_u32 %1, %0, %2\n v_ashrrev_i32 %1, 31, %1\n v_and_b32 %0, %1, %0\n v_add_u32 %0, 3, %0\n .endr" : "+v"(x), "+v"(y) : "v"(z)); END();
    // 38: 64 x (v_cmp -> sgpr pair ; v_and with that SGPR as data) incl. the 2 wait states the compiler inserts
    BEGIN(); asm volatile(".rept 64\n v_cmp_ne_u32 s[20:21], 0, %0\n s_nop 1\n v_and_b32 %1, s20, %0\n v_add_u32 %0, %1, %0\n .endr" : "+v"(x), "+v"(y) :: "s20", "s21"); END();
    // 39: 64 x (v_cmp -> sgpr pair ; v_cndmask with that mask)
    BEGIN(); asm volatile(".rept 64\n v_cmp_ne_u32 s[20:21], 0, %0\n v_cndmask_b32 %1, 1, %0, s[20:21]\n v_add_u32 %0, %1, %0\n .endr" : "+v"(x), "+v"(y) :: "s20", "s21"); END();
    // 40: 16 x (4 buffer-less global stores to distinct lines, then 16 dependent VALU) : does store issue stall the VALU?
    { uint32_t* gp = grw + 1024 + lane; BEGIN(); for (int i = 0; i < 16; ++i) { gp[0] = x; gp[64] = x; gp[128] = x; gp[192] = x; asm volatile(".rept 16\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x)); } END(); }
    // 41: the same 16 x 16 VALU without stores
    BEGIN(); for (int i = 0; i < 16; ++i) { asm volatile(".rept 16\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x)); } END();
    // 42: 16 x (1 global atomic-or x2 to distinct words, 16 VALU)
    { unsigned long long* ap = (unsigned long long*)(grw + 2048) + lane; BEGIN(); for (int i = 0; i < 16; ++i) { atomicOr(ap + 64 * (i & 3), 1ull << i); asm volatile(".rept 16\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x)); } END(); }
    // 43: 16 x (ds_write_b64 ; 16 VALU)
    { uint64_t* l64 = (uint64_t*)lds; BEGIN(); for (int i = 0; i < 16; ++i) { l64[64 + lane] = x; asm volatile(".rept 16\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x) :: "memory"); } END(); }
    // 44: 16 x (global load L1-resident -> wait -> 3 readlane -> s_ logic -> v_) the head of a pop
    { uint32_t p = 0; BEGIN(); for (int i = 0; i < 16; ++i) { p = gro[64 + ((p + lane) & 63u)]; const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)p, 4), b = (uint32_t)__builtin_amdgcn_readlane((int)p, 5); p = (a & 7u) + (b & 3u) + (uint32_t)lane; asm volatile("" : "+v"(p)); } END(); x += p; }
    // 45: 64 x DPP quad_perm step with select (the own-best minimum): mov_dpp, s_nop, min, cmp, cndmask
    BEGIN(); asm volatile(".rept 64\n v_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x), "+v"(y) :: "vcc"); END();
    // 46: 64 x v_readfirstlane -> s_cmp -> s_cselect (no VALU consumer): cost of pulling a uniform value to the scalar side
    BEGIN(); asm volatile(".rept 64\n v_readfirstlane_b32 %1, %0\n s_cmp_eq_u32 %1, 0\n s_cselect_b32 %1, 1, 2\n v_add_u32 %0, 1, %0\n .endr" : "+v"(x), "+s"(s) :: "scc"); END();

    if (lane == 0) { grw[0] = x + y + z + w + s; out->cyc[63] = (uint64_t)k; }
}

// two waves of one workgroup: barrier ping-pong and LDS hand-overs
__global__ __launch_bounds__(128) void k_pair(Res* out, uint32_t* grw)
{
    __shared__ volatile uint32_t box[64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 64) box[threadIdx.x] = 0;
    __syncthreads();
    uint64_t t0, t1;
    int k = 0;
    // 0: 256 s_barrier in a row (both waves do nothing else)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t0 = CLK();
    for (int i = 0; i < 256; ++i) asm volatile("s_barrier" ::: "memory");
    t1 = CLK(); if (threadIdx.x == 0) out->cyc[k] = t1 - t0; ++k;
    // 1: hand-over through LDS with a barrier: wave 0 writes, barrier, wave 1 reads + writes, barrier, wave 0 reads  (128 round trips)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t0 = CLK();
    uint32_t v = 0;
    for (int i = 0; i < 128; ++i) {
        if (wv == 0 && lane == 0) box[0] = v + 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
        if (wv == 1) { const uint32_t r = box[0]; if (lane == 0) box[1] = r + 1; }
        asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
        if (wv == 0) v = box[1];
    }
    t1 = CLK(); if (threadIdx.x == 0) { out->cyc[k] = t1 - t0; grw[1] = v; } ++k;
    __syncthreads();
    // 2: the same hand-over with LDS flags and polling instead of barriers (128 round trips)
    if (threadIdx.x < 64) box[threadIdx.x] = 0;
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t0 = CLK();
    if (wv == 0) {
        uint32_t seq = 0;
        for (int i = 0; i < 128; ++i) {
            ++seq;
            if (lane == 0) box[0] = seq;
            while (box[1] != seq) { }
        }
        v = seq;
    } else {
        uint32_t seq = 0;
        for (int i = 0; i < 128; ++i) {
            ++seq;
            while (box[0] != seq) { }
            if (lane == 0) box[1] = seq;
        }
        v = seq;
    }
    t1 = CLK(); if (threadIdx.x == 0) { out->cyc[k] = t1 - t0; grw[2] = v; } ++k;
    __syncthreads();
    // 3: polling with s_sleep 1 between polls
    if (threadIdx.x < 64) box[threadIdx.x] = 0;
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t0 = CLK();
    if (wv == 0) {
        uint32_t seq = 0;
        for (int i = 0; i < 128; ++i) { ++seq; if (lane == 0) box[0] = seq; while (box[1] != seq) { __builtin_amdgcn_s_sleep(1); } }
        v = seq;
    } else {
        uint32_t seq = 0;
        for (int i = 0; i < 128; ++i) { ++seq; while (box[0] != seq) { __builtin_amdgcn_s_sleep(1); } if (lane == 0) box[1] = seq; }
        v = seq;
    }
    t1 = CLK(); if (threadIdx.x == 0) { out->cyc[k] = t1 - t0; grw[3] = v; } ++k;
    __syncthreads();
    // 4: wave 0 runs 256 dependent v_add while wave 1 (other SIMD) spins on LDS: does a polling partner slow the worker?
    if (threadIdx.x < 64) box[threadIdx.x] = 0;
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t0 = CLK();
    if (wv == 0) {
        uint32_t x = (uint32_t)lane;
        asm volatile(".rept 1024\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x));
        t1 = CLK();
        if (lane == 0) { box[0] = 1; grw[4] = x; out->cyc[k] = t1 - t0; }
    } else {
        while (box[0] != 1) { }
    }
    ++k;
    if (threadIdx.x == 0) out->cyc[63] = (uint64_t)k;
}

int main()
{
    const int NCH = 2 * 1024 * 1024;                 // 8 MB of chase indices
    std::vector<uint32_t> h(4096 + NCH);
    h[0] = 17;
    for (int i = 0; i < 64; ++i) h[64 + i] = (uint32_t)((i * 37 + 11) & 63);
    uint64_t st = 88172645463325252ull;
    for (int i = 0; i < NCH; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[4096 + i] = (uint32_t)(st % (uint64_t)NCH); }
    uint32_t *d_ro, *d_rw; Res *d_res, *d_res2;
    hipMalloc(&d_ro, h.size() * 4); hipMalloc(&d_rw, 4096 * 4); hipMalloc(&d_res, sizeof(Res)); hipMalloc(&d_res2, sizeof(Res));
    hipMemcpy(d_ro, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_rw, 0, 4096 * 4);
    Res r, r2;
    const char* names[] = {
        "empty (timer overhead)", "256 dependent v_add_u32", "256 v_add_u32, 4 chains", "256 dependent s_add_u32",
        "64 x readfirstlane->s_add->v_mov", "64 x v_cmp + s_cbranch_vccz NOT taken", "64 x v_cmp + s_cbranch_vccnz TAKEN",
        "64 x s_cmp + s_cbranch_scc1 NOT taken", "64 x s_cmp + s_cbranch_scc1 TAKEN", "64 x saveexec region skipped (execz taken)",
        "64 x saveexec region entered", "64 x v_cmp + v_add + v_cndmask", "64 x s_branch taken", "64 x s_waitcnt (nothing pending)",
        "64 LDS chase ds_read_b32 (vector addr)", "64 LDS chase via readfirstlane", "64 ds_write -> ds_read same addr",
        "64 ds_bpermute chain", "64 DPP row_shr chain (+s_nop 1)", "64 readlane dependent lane idx", "64 ballot->ffs->readlane",
        "64 global chase, L1-resident ring", "64 global chase, 8 MB (L2/MALL)", "64 scalar-load chase (s_load)",
        "32 global store -> nt load back", "32 volatile store -> load back", "64 global stores + wait", "64 global atomic_or + wait",
        "64 s_memtime", "64 x readlane->s_cmp->s_cselect->v_add", "64 x v_cmp->s_and exec->s_cselect->v_add", "64 dependent v_mul_lo_u32",
        "64 dependent v_mul_i32_i24", "64 dependent v_lshlrev_b64", "64 ds_read2_b64 pair chase",
        "64 x v_readlane -> (s_nop) -> v_add reads SGPR", "64 x s_add -> v_add reads SGPR", "64 x v_sub,v_ashr,v_and,v_add masks", "64 x v_cmp->sgpr, nop, v_and sgpr data, v_add",
        "64 x v_cmp->sgpr, v_cndmask, v_add", "16 x (4 global stores + 16 VALU)", "16 x 16 VALU (no stores)", "16 x (atomic_or_x2 + 16 VALU)", "16 x (ds_write_b64 + 16 VALU)",
        "16 x load->wait->2 readlane->s_and->v_add", "64 x dpp mov, cmp, cndmask, add", "64 x readfirstlane, s_cmp, s_cselect, v_add"};
    const int per[] = {1, 256, 256, 256, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 32, 32, 64, 64, 64, 64, 64, 64, 64, 64, 64,
                       64, 64, 64, 64, 64, 16, 16, 16, 16, 16, 64, 64};
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_single, dim3(1), dim3(64), 0, 0, d_res, d_ro, d_rw, NCH);
        hipLaunchKernelGGL(k_pair, dim3(1), dim3(128), 0, 0, d_res2, d_rw);
        hipDeviceSynchronize();
    }
    hipMemcpy(&r, d_res, sizeof(Res), hipMemcpyDeviceToHost);
    hipMemcpy(&r2, d_res2, sizeof(Res), hipMemcpyDeviceToHost);
    printf("# single wave, cycles (s_memtime) -- total, minus timer overhead, per item\n");
    const uint64_t ovh = r.cyc[0];
    for (int i = 0; i < (int)r.cyc[63] && i < 47; ++i)
        printf("%2d %-46s total %8llu  per %8.1f\n", i, names[i], (unsigned long long)r.cyc[i], (double)((long long)r.cyc[i] - (long long)ovh) / per[i]);
    const char* n2[] = {"256 s_barrier (2 waves)", "128 LDS hand-over round trips with 2 barriers each", "128 LDS hand-over round trips, polling",
                        "128 round trips, polling + s_sleep 1", "1024 dependent v_add beside a polling partner"};
    const int per2[] = {256, 128, 128, 128, 1024};
    printf("# two waves of one workgroup\n");
    for (int i = 0; i < (int)r2.cyc[63] && i < 5; ++i)
        printf("%2d %-52s total %8llu  per %8.1f\n", i, n2[i], (unsigned long long)r2.cyc[i], (double)r2.cyc[i] / per2[i]);
    return 0;
}
