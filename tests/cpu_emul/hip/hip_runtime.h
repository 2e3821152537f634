// TEST INFRASTRUCTURE — a minimal "HIP on pthreads" shim so that the UNMODIFIED kernel sources of
// readtape_amd/csrc/*.hip can be compiled with g++ and exercised in the GPU-less build container
// (tests/test_emul_*.py, marked not-gpu).  One workgroup runs at a time; its threads are FIBERS
// (ucontext) of one OS thread, run round robin: __syncthreads() and the wave intrinsics (__ballot,
// __shfl_up, __shfl: a rendezvous per 64-thread wave) hand the processor to the next fiber until the
// last one has arrived - a switch costs ~100 ns where a pthread barrier of 256 threads on 8 cores
// costs ~100 us, and a run is deterministic.  It exists to debug kernel LOGIC on the CPU; it is never
// loaded by the product (readtape_amd/frontend.py only opens librtfe.so and fails loudly without it).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <ucontext.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __restrict__
#define __launch_bounds__(...)
#define __forceinline__ inline
#define RTFE_CPU_EMUL 1

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct int4 { int x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { unsigned int x, y; };
struct uint4 { unsigned int x, y, z, w; };
inline int4 make_int4(int a, int b, int c, int d) { int4 r = {a, b, c, d}; return r; }
inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { uint4 r = {a, b, c, d}; return r; }
inline uint2 make_uint2(unsigned a, unsigned b) { uint2 r = {a, b}; return r; }
struct float2 { float x, y; };
inline float2 make_float2(float a, float b) { float2 r = {a, b}; return r; }
struct float4 { float x, y, z, w; };
inline float4 make_float4(float a, float b, float c, float d) { float4 r = {a, b, c, d}; return r; }
extern dim3 threadIdx, blockIdx;      // (of the fiber that is running)
extern dim3 blockDim, gridDim;
extern unsigned char g_dyn_smem[160 * 1024] __attribute__((aligned(64)));

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { int multiProcessorCount; };
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 1; return hipSuccess; }
template <class T> inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)calloc(1, n); return *p ? hipSuccess : 1; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 1; return hipSuccess; }
typedef void *hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;       // (launches are synchronous here: a second stream is the first)
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return 1; }      // "no side stream": the caller stays in line
inline hipError_t hipDeviceGetStreamPriorityRange(int *a, int *b) { *a = 0; *b = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return 1; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
// HIP graphs: the emulator cannot capture - rtfe_scan falls back to direct launches (the path a runtime that refuses a capture takes)
typedef void *hipGraph_t;
typedef void *hipGraphExec_t;
enum { hipStreamCaptureModeThreadLocal = 1 };
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return 1; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = nullptr; return 1; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, void *, void *, size_t) { return 1; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }

namespace hipemu {
void fiber_yield();                    // give the processor to the next fiber of the block (emul_main.cpp)
struct Barrier {                       // all n fibers of a block / a wave: the last to arrive opens it, the others yield until then
   unsigned n = 0, arrived = 0, gen = 0;
   void init(unsigned k) { n = k; arrived = 0; gen = 0; }
   void wait() {
      if (++arrived == n) { arrived = 0; ++gen; return; }
      const unsigned g = gen;
      while (gen == g) fiber_yield(); } };
extern Barrier g_block_barrier;
extern Barrier g_wave_barrier[16];
extern unsigned long long g_wave_scratch[16][64];
}  // namespace hipemu

inline long long clock64() { return 0; }
inline void __threadfence() {}
inline void __syncthreads() { hipemu::g_block_barrier.wait(); }
inline unsigned long long __ballot(int pred) {
   const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
   hipemu::g_wave_scratch[w][l] = pred ? 1 : 0;
   hipemu::g_wave_barrier[w].wait();
   unsigned long long m = 0;
   const unsigned nl = std::min(64u, blockDim.x - w * 64);
   for (unsigned i = 0; i < nl; ++i) m |= hipemu::g_wave_scratch[w][i] << i;
   hipemu::g_wave_barrier[w].wait();
   return m; }
inline int __shfl_up(int v, unsigned delta) {
   const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
   hipemu::g_wave_scratch[w][l] = (unsigned long long)(unsigned)v;
   hipemu::g_wave_barrier[w].wait();
   const int r = l >= delta ? (int)(unsigned)hipemu::g_wave_scratch[w][l - delta] : v;
   hipemu::g_wave_barrier[w].wait();
   return r; }
inline int __shfl(int v, int src) {
   const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
   hipemu::g_wave_scratch[w][l] = (unsigned long long)(unsigned)v;
   hipemu::g_wave_barrier[w].wait();
   const int r = (int)(unsigned)hipemu::g_wave_scratch[w][src & 63];
   hipemu::g_wave_barrier[w].wait();
   return r; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline unsigned int __float_as_uint(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline unsigned int __umulhi(unsigned int a, unsigned int b) { return (unsigned int)(((unsigned long long)a * b) >> 32); }
inline int __popc(unsigned int v) { return __builtin_popcount(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned int atomicAdd(unsigned int *p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicExch(int *p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline int atomicAnd(int *p, int v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
inline int atomicMax(int *p, int v) { int old = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return old; }
inline int atomicMin(int *p, int v) { int old = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return old; }
inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {
   unsigned long long old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
   while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
   return old; }
using std::max;
using std::min;

namespace hipemu { void run_grid(dim3 grid, dim3 block, const std::function<void()> &body); }
template <class F, class... A>
inline void hipemu_launch(F kernel, dim3 grid, dim3 block, size_t /*smem*/, hipStream_t, A... args) {
   hipemu::run_grid(grid, block, [=]() { kernel(args...); }); }
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) hipemu_launch(kernel, grid, block, smem, stream, __VA_ARGS__)
