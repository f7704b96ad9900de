// wave_sim.hpp -- TEST INFRASTRUCTURE: a lane-level CPU simulator for the HIP kernel sources of iris_lama_amd/csrc.
//
// There is no GPU in the build container and GPU minutes are scarce, so the kernel SOURCE (not a restatement of it) is compiled
// for the host against tests/sim/hip/hip_runtime.h and executed here: every thread of a workgroup is a ucontext fiber, a wave
// is 64 consecutive fibers, and the wave-level operations the kernels use (__ballot, __shfl, readlane / readfirstlane, DPP moves)
// as well as the workgroup barrier are rendezvous points of those fibers.  The result is the kernels' LOGIC executed lane by lane
// with wave64 semantics; it says nothing about timing, memory ordering or the code the device compiler generates -- the `-m gpu`
// tests on the real MI355X remain the parity tests proper.  Nothing of this is shipped, loaded or linked by the product
// (liblama_hip.so is built by hipcc from the same sources without LAMA_WAVE_SIM).
//
// Model and its limits:
//   * one workgroup runs at a time, workgroups of a grid in order (the kernels never synchronise across workgroups except
//     through atomics / a spin on another workgroup's directory allocation, which cannot happen when workgroups are serial);
//   * a wave operation must be reached by ALL live lanes of the wave at the same call site (wave-uniform control flow around
//     cross-lane operations -- the style the kernels are written in); a rendezvous at two different call sites is reported;
//   * lanes that returned from the kernel no longer take part (ballot bit 0, reading their registers is an error);
//   * atomics are plain read-modify-writes (fibers are cooperative: no preemption).
#pragma once
#include <ucontext.h>      // (fallback for targets other than x86-64: see Ctx below)

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace wsim {

struct dim3s { unsigned x, y, z; dim3s(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int WAVE = 64;

struct Wave {
    int live = 0;                  // lanes that have not returned
    int arrived = 0;
    uint64_t gen = 0;
    int site = 0;                  // call site of the rendezvous in progress
    uint64_t in[2][WAVE];          // per-lane operands, double buffered by generation parity
    uint64_t dep[2][WAVE];         // generation + 1 in which the operand was deposited (a lane that exited earlier deposited nothing)
    bool pred[2][WAVE];
    bool alive[WAVE];
    uint64_t ballot_result[2];
    uint64_t my_gen[WAVE];         // generation of the rendezvous the lane last took part in
    int lane_site[WAVE];           // last rendezvous each lane arrived at (diagnostics)
    uint64_t lane_count[WAVE];     // rendezvous each lane has taken part in
};

// A fiber switch.  glibc's swapcontext saves and restores the signal mask -- a system call per switch, and a workgroup of the
// brushfire makes millions of them; on x86-64 the switch is a dozen instructions instead (callee-saved registers + stack pointer),
// which made the simulator tests ~4x faster (round 6).  Elsewhere ucontext is used as before.
#if defined(__x86_64__)
struct Ctx { void* sp = nullptr; };
extern "C" void wsim_switch(Ctx* from, Ctx* to);
#ifndef WSIM_SWITCH_DEFINED
#define WSIM_SWITCH_DEFINED
__asm__(
    ".text\n"
    ".weak wsim_switch\n"
    ".type wsim_switch,@function\n"
    "wsim_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq (%rsi), %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size wsim_switch, .-wsim_switch\n");
#endif
inline void ctx_make(Ctx& c, char* stack, size_t bytes, void (*entry)())
{
    // stack top, 16-byte aligned; the frame wsim_switch pops: six callee-saved registers, then `ret` into `entry` with the stack as
    // the ABI wants it at a function's first instruction (a return address slot below a 16-byte boundary)
    uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
    uint64_t* sp = (uint64_t*)top;
    *--sp = 0;                          // (entry never returns: a null return address)
    *--sp = (uint64_t)(uintptr_t)entry; // popped by `ret`
    for (int i = 0; i < 6; ++i) *--sp = 0;
    c.sp = sp;
}
inline void ctx_switch(Ctx& from, Ctx& to) { wsim_switch(&from, &to); }
#else
struct Ctx { ucontext_t u; };
inline void ctx_make(Ctx& c, char* stack, size_t bytes, void (*entry)())
{
    getcontext(&c.u);
    c.u.uc_stack.ss_sp = stack; c.u.uc_stack.ss_size = bytes; c.u.uc_link = nullptr;
    makecontext(&c.u, entry, 0);
}
inline void ctx_switch(Ctx& from, Ctx& to) { swapcontext(&from.u, &to.u); }
#endif

struct Fiber {
    Ctx ctx;
    char* stack = nullptr;
    bool done = false;
    unsigned tid = 0;
};

struct Block {
    dim3s grid, block, bidx;
    unsigned nthreads = 0;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int cur = -1;
    Ctx sched;
    // workgroup barrier
    int bar_arrived = 0, bar_live = 0;
    uint64_t bar_gen = 0;
    std::function<void()> body;
    uint64_t switches = 0;
    uint64_t progress = 0;          // arrivals at a rendezvous / barrier and fibers that finished: a scheduler round without any is a deadlock
};

inline Block*& blk() { static Block* b = nullptr; return b; }
inline std::function<void()>& deadlock_hook() { static std::function<void()> h; return h; }     // set by a debug build of the code under test
inline std::vector<char*>& stack_pool() { static std::vector<char*> p; return p; }

inline unsigned tid() { return blk()->fibers[blk()->cur].tid; }
inline int lane() { return (int)(tid() & 63u); }
inline Wave& wave() { return blk()->waves[tid() >> 6]; }

inline void yield()
{
    Block* b = blk();
    ++b->switches;
    ctx_switch(b->fibers[b->cur].ctx, b->sched);
}

[[noreturn]] inline void die(const char* msg)
{
    std::fprintf(stderr, "wave_sim: %s (block %u,%u thread %u)\n", msg, blk() ? blk()->bidx.x : 0u, blk() ? blk()->bidx.y : 0u, blk() && blk()->cur >= 0 ? tid() : 0u);
    std::abort();
}

// rendezvous of all live lanes of the calling lane's wave; returns the generation parity whose buffers hold this round's operands
inline int wave_sync(int site, uint64_t operand, bool pred)
{
    Wave& w = wave();
    const int l = lane();
    const int par = (int)(w.gen & 1u);
    w.lane_site[l] = site; ++w.lane_count[l];
    w.my_gen[l] = w.gen;
    if (w.arrived == 0) w.site = site;
    else if (w.site != site) {
        std::fprintf(stderr, "wave_sim: lanes of one wave met at different cross-lane operations (source lines %d and %d): control flow around them is not wave-uniform\n", w.site, site);
        for (int i = 0; i < WAVE; ++i) std::fprintf(stderr, " lane %d: line %d (#%llu)%s", i, w.lane_site[i], (unsigned long long)w.lane_count[i], (i & 3) == 3 ? "\n" : "");
        die("divergent rendezvous");
    }
    w.in[par][l] = operand;
    w.dep[par][l] = w.gen + 1;
    w.pred[par][l] = pred;
    ++w.arrived;
    ++blk()->progress;
    if (w.arrived == w.live) {
        uint64_t m = 0;
        for (int i = 0; i < WAVE; ++i) if (w.alive[i] && w.pred[par][i]) m |= 1ull << i;
        w.ballot_result[par] = m;
        w.arrived = 0;
        ++w.gen;
    } else {
        const uint64_t g = w.gen;
        while (w.gen == g) yield();
        ++blk()->progress;              // released: what this lane does next (e.g. set a flag another wave polls) is progress too
    }
    return par;
}

inline uint64_t ballot(bool p, int site) { const int par = wave_sync(site, 0, p); return wave().ballot_result[par]; }

template <class T>
inline uint64_t to_bits(T v) { uint64_t u = 0; static_assert(sizeof(T) <= 8, "operand too wide"); std::memcpy(&u, &v, sizeof(T)); return u; }
template <class T>
inline T from_bits(uint64_t u) { T v; std::memcpy(&v, &u, sizeof(T)); return v; }

template <class T>
inline T shfl(T v, int src, int site)
{
    const int par = wave_sync(site, to_bits(v), true);
    Wave& w = wave();
    src &= 63;
    // the operand of a lane that took part in THIS rendezvous stays valid even if that lane has returned from the kernel since;
    // a lane that had exited before deposited nothing (undefined on the device; the kernels never rely on it): 0
    return from_bits<T>(w.dep[par][src] == w.my_gen[lane()] + 1 ? w.in[par][src] : 0ull);
}
// v_readlane_b32 reads the register of lane `l` whether or not the lane is active; the lane index must be wave-uniform
template <class T>
inline T readlane(T v, int l, int site) { return shfl(v, l, site); }
template <class T>
inline T readfirstlane(T v, int site)
{
    const int par = wave_sync(site, to_bits(v), true);
    Wave& w = wave();
    for (int i = 0; i < WAVE; ++i) if (w.dep[par][i] == w.my_gen[lane()] + 1) return from_bits<T>(w.in[par][i]);
    return v;
}

// v_mov_b32 with a DPP control (gfx9 encodings).  Lanes whose source is invalid keep `old` unless bound_ctrl (then 0); rows /
// banks not enabled keep `old`.
inline int dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int site)
{
    const int par = wave_sync(site, (uint64_t)(uint32_t)src, true);
    Wave& w = wave();
    const int l = lane();
    const int row = l >> 4, inrow = l & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (inrow >> 2)) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x000 && ctrl <= 0x0FF) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);                  // quad_perm
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = inrow + (ctrl & 15); if (s < 16) from = (row << 4) | s; }     // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = inrow - (ctrl & 15); if (s >= 0) from = (row << 4) | s; }     // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12F) from = (row << 4) | ((inrow - (ctrl & 15)) & 15);                             // row_ror
    else if (ctrl == 0x130) { if (l + 1 < 64) from = l + 1; }                                                              // wave_shl:1
    else if (ctrl == 0x134) from = (l + 1) & 63;                                                                           // wave_rol:1
    else if (ctrl == 0x138) { if (l - 1 >= 0) from = l - 1; }                                                              // wave_shr:1
    else if (ctrl == 0x13C) from = (l - 1) & 63;                                                                           // wave_ror:1
    else if (ctrl == 0x140) from = (row << 4) | (15 - inrow);                                                              // row_mirror
    else if (ctrl == 0x141) from = (row << 4) | (inrow < 8 ? 7 - inrow : 23 - inrow);                                      // row_half_mirror
    else if (ctrl == 0x142) { if (row >= 1) from = ((row - 1) << 4) | 15; }                                                // row_bcast15: lane 15 of the previous row
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }                                                                   // row_bcast31: lane 31 into rows 2, 3
    else die("unsupported DPP control");
    if (ctrl == 0x142 && !(row >= 1)) return old;
    if (ctrl == 0x143 && !(row >= 2)) return old;
    if (from < 0) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)(w.dep[par][from] == w.my_gen[l] + 1 ? w.in[par][from] : 0ull);
}

inline void syncthreads()
{
    Block* b = blk();
    ++b->bar_arrived;
    ++b->progress;
    if (b->bar_arrived == b->bar_live) { b->bar_arrived = 0; ++b->bar_gen; }
    else { const uint64_t g = b->bar_gen; while (b->bar_gen == g) yield(); ++b->progress; }
}

inline uint64_t clock() { return blk()->switches; }

inline void fiber_entry()
{
    Block* b = blk();
    b->body();
    Fiber& f = b->fibers[b->cur];
    f.done = true;
    ++b->progress;
    Wave& w = b->waves[f.tid >> 6];
    w.alive[f.tid & 63] = false;
    --w.live;
    --b->bar_live;
    // a lane that leaves may complete a rendezvous the others are waiting in
    if (w.live > 0 && w.arrived == w.live) {
        const int par = (int)(w.gen & 1u);
        uint64_t m = 0;
        for (int i = 0; i < WAVE; ++i) if (w.alive[i] && w.pred[par][i]) m |= 1ull << i;
        w.ballot_result[par] = m;
        w.arrived = 0;
        ++w.gen;
    }
    if (b->bar_live > 0 && b->bar_arrived == b->bar_live) { b->bar_arrived = 0; ++b->bar_gen; }
    ctx_switch(f.ctx, b->sched);
}

// Runs `body` once per thread of every workgroup of the grid (workgroups in x-major order, one at a time).
inline void launch(dim3s grid, dim3s block, std::function<void()> body)
{
    Block b;
    b.grid = grid; b.block = block;
    b.nthreads = block.x * block.y * block.z;
    b.body = std::move(body);
    b.fibers.resize(b.nthreads);
    b.waves.resize((b.nthreads + 63) / 64);
    auto& pool = stack_pool();
    while (pool.size() < b.nthreads) pool.push_back((char*)std::malloc(STACK_BYTES));
    Block* saved = blk();
    blk() = &b;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        b.bidx = dim3s(bx, by, bz);
        for (auto& w : b.waves) { w.live = 0; w.arrived = 0; w.gen = 0; std::memset(w.alive, 0, sizeof(w.alive)); std::memset(w.lane_site, 0, sizeof(w.lane_site)); std::memset(w.dep, 0, sizeof(w.dep)); std::memset(w.lane_count, 0, sizeof(w.lane_count)); }
        for (unsigned t = 0; t < b.nthreads; ++t) {
            Fiber& f = b.fibers[t];
            f.tid = t; f.done = false; f.stack = pool[t];
            ctx_make(f.ctx, f.stack, STACK_BYTES, (void (*)())fiber_entry);
            Wave& w = b.waves[t >> 6];
            w.alive[t & 63] = true; ++w.live;
        }
        b.bar_live = (int)b.nthreads; b.bar_arrived = 0; b.bar_gen = 0;
        unsigned remaining = b.nthreads;
        while (remaining) {
            const uint64_t before = b.progress;
            for (unsigned t = 0; t < b.nthreads; ++t) {
                Fiber& f = b.fibers[t];
                if (f.done) continue;
                b.cur = (int)t;
                ctx_switch(b.sched, f.ctx);
                if (f.done) --remaining;
            }
            // a whole round in which no fiber arrived anywhere or finished: every fiber is waiting for something nobody will do
            if (remaining && b.progress == before) {
                for (unsigned t = 0; t < b.nthreads; ++t) if (!b.fibers[t].done) { b.cur = (int)t; break; }
                for (size_t wv = 0; wv < b.waves.size(); ++wv)
                    std::fprintf(stderr, "wave_sim: wave %zu: %d of %d live lanes wait at the cross-lane operation of source line %d\n", wv, b.waves[wv].arrived, b.waves[wv].live, b.waves[wv].site);
                std::fprintf(stderr, "wave_sim: workgroup barrier: %d of %d arrived\n", b.bar_arrived, b.bar_live);
                if (deadlock_hook()) deadlock_hook()();
                die("deadlock: no fiber of the workgroup can make progress");
            }
        }
    }
    blk() = saved;
}

// LAMA_SIM_TIMING=1: where the simulator's time goes, per kernel (printed when the library is unloaded)
struct LaunchTimes {
    std::map<std::string, std::pair<double, uint64_t>> t;
    ~LaunchTimes()
    {
        if (!std::getenv("LAMA_SIM_TIMING")) return;
        for (auto& kv : t) std::fprintf(stderr, "wave_sim: %-60s %8.2f s in %llu workgroups\n", kv.first.c_str(), kv.second.first, (unsigned long long)kv.second.second);
    }
};
inline LaunchTimes& launch_times() { static LaunchTimes l; return l; }
inline void launch_named(const char* name, dim3s grid, dim3s block, std::function<void()> body)
{
    const auto t0 = std::chrono::steady_clock::now();
    launch(grid, block, std::move(body));
    auto& e = launch_times().t[name];
    e.first += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    e.second += (uint64_t)grid.x * grid.y * grid.z;
}

} // namespace wsim
