// tests/sim/hip/hip_runtime.h -- TEST INFRASTRUCTURE: stands in for <hip/hip_runtime.h> when the kernel sources of
// iris_lama_amd/csrc are compiled for the HOST and run under the lane-level simulator of tests/sim/wave_sim.hpp (see there).
// Device memory is host memory, a stream executes immediately, a kernel launch runs its workgroups one after the other.
#pragma once
#define LAMA_WAVE_SIM 1

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../wave_sim.hpp"

// ---- qualifiers ------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static           /* one workgroup at a time: a function-local static IS the workgroup's LDS */

using dim3 = wsim::dim3s;
struct wsim_tidx { unsigned x, y, z; };
inline wsim_tidx wsim_thread_idx() { const wsim::Block* b = wsim::blk(); const unsigned t = wsim::tid(); return {t % b->block.x, (t / b->block.x) % b->block.y, t / (b->block.x * b->block.y)}; }
#define threadIdx (wsim_thread_idx())
#define blockIdx (wsim::blk()->bidx)
#define blockDim (wsim::blk()->block)
#define gridDim (wsim::blk()->grid)

// ---- wave-level operations (rendezvous points; the source line is the call-site id) ------------------------------------------
#define __ballot(p) wsim::ballot((p) ? true : false, __LINE__)
#define __shfl(v, src, ...) wsim::shfl((v), (src), __LINE__)
#define __shfl_xor(v, m, ...) wsim::shfl((v), wsim::lane() ^ (m), __LINE__)
#define __builtin_amdgcn_readlane(v, l) wsim::readlane((int)(v), (int)(l), __LINE__)
#define __builtin_amdgcn_readfirstlane(v) wsim::readfirstlane((int)(v), __LINE__)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) wsim::dpp((int)(old), (int)(src), (ctrl), (rm), (bm), (bc), __LINE__)
// A wavefront-scope fence says "this wave's earlier stores are visible to its later loads" -- on the device that holds between LANES
// too, because a wave issues its memory instructions in program order; for cooperative fibers it has to be a rendezvous (round 6:
// the one-wave brushfire loops have nothing else between the stores of one pop and the loads of the next, and a fiber that ran
// ahead read cells its neighbours had not written yet -- found by the randomised small-room cases, never on the device)
#define __builtin_amdgcn_fence(...) ((void)__ballot(true))
#define __builtin_amdgcn_s_sleep(n) wsim::yield()
#define __builtin_readcyclecounter() wsim::clock()
#define __syncthreads() wsim::syncthreads()
#define __threadfence() ((void)0)
#define __threadfence_block() ((void)0)

inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline long long __double_as_longlong(double d) { long long r; std::memcpy(&r, &d, 8); return r; }
inline double __longlong_as_double(long long v) { double r; std::memcpy(&r, &v, 8); return r; }

// ---- atomics (cooperative fibers: a plain read-modify-write is atomic) -------------------------------------------------------
template <class T, class U> inline T atomicAdd(T* p, U v) { const T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> inline T atomicSub(T* p, U v) { const T o = *p; *p = (T)(o - (T)v); return o; }
template <class T, class U> inline T atomicOr(T* p, U v) { const T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> inline T atomicAnd(T* p, U v) { const T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> inline T atomicMin(T* p, U v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> inline T atomicMax(T* p, U v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> inline T atomicExch(T* p, U v) { const T o = *p; *p = (T)v; return o; }
template <class T, class U, class V> inline T atomicCAS(T* p, U cmp, V v) { const T o = *p; if (o == (T)cmp) *p = (T)v; return o; }

// ---- host runtime ------------------------------------------------------------------------------------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorOutOfMemory = 2;
typedef struct wsim_stream* hipStream_t;
typedef struct wsim_event* hipEvent_t;
struct wsim_stream { int dummy; };
struct wsim_event { int dummy; };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
constexpr unsigned hipHostMallocDefault = 0;

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "simulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = std::getenv("LAMA_SIM_NO_DEVICE") ? 0 : 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { *p = (T*)std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = nullptr)
{
    for (size_t r = 0; r < height; ++r) std::memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)64 << 30; *total_b = (size_t)64 << 30; return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
constexpr hipError_t hipErrorPeerAccessAlreadyEnabled = 704;
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new wsim_stream{0}; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new wsim_event{0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
struct hipDeviceProp_t { int multiProcessorCount = 256; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t{}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // launches run to completion in program order here
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

// kernel launch: every argument is copied once (as the device would receive it), every thread calls the kernel with the copies
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    wsim::launch_named(#kernel, wsim::dim3s(grid), wsim::dim3s(block), [=]() { (kernel)(__VA_ARGS__); })

// ---- small vector types / bit casts / scoped atomic loads ----------------------------------------------------------------------
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) int4 { int x, y, z, w; };
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_WAVEFRONT 1
#define __HIP_MEMORY_SCOPE_SYSTEM 4
template <class T> inline T __hip_atomic_load(const T* p, int, int) { return *(const volatile T*)p; }
template <class T, class U> inline void __hip_atomic_store(T* p, U v, int, int) { *(volatile T*)p = (T)v; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
