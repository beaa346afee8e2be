// TEST INFRASTRUCTURE ONLY -- NOT PRODUCT CODE, NEVER SHIPPED, NEVER LOADED BY automerge_classic_amd.
//
// A tiny stand-in for <hip/hip_runtime.h> that lets g++ compile the engine's .hip sources into a CPU
// "emulation" library (tests/emu/libam355_emu.so) so that kernel *logic* can be checked against the oracle in
// the GPU-less build container (`pytest -m "not gpu"`). Every kernel thread of a block becomes a fiber (ucontext) of
// the launching OS thread; blocks run one after another; __syncthreads()/__ballot()/__shfl() are barriers at which a
// fiber hands over to the next one (round robin); wave size is 64. (Until round 3 the threads were OS threads of a
// 1024-thread pool spinning on sched_yield: the same semantics at ten times the cost.) It is not a fallback: the product library
// (automerge_classic_amd/csrc/libam355.so) is built by hipcc for gfx950 only and fails loudly without a GPU.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define AM355_EMULATED 1
#define AM355_STREAMS_ORDER_ACROSS_THREADS 0   // (a launch runs to its end inside the call: am355_device.h)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNoDevice = 100, hipErrorNotReady = 600 };
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost, hipMemcpyDefault };

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated hip error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
constexpr unsigned hipDeviceScheduleSpin = 1;
inline hipError_t hipSetDeviceFlags(unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount = 1; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t{}; return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0

// ---- per-thread coordinates --------------------------------------------------------------------------
struct emu_uint3 { unsigned x, y, z; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
// LDS is ordinary memory here
#define AM355_LDS_BYTES_DEFINED 1
typedef const uint8_t* LdsBytes;

extern thread_local emu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

extern const char* emu_current_kernel;

namespace emu {

void yield();  // hand over to the next fiber of the block (cooperative launches only)
void stuck(const char* what, unsigned arrived, unsigned n);

class Barrier {  // reusable counting barrier between the fibers of a block (all on one OS thread: plain counters)
 public:
  void reset(unsigned n) { n_ = n; count_ = 0; }
  void wait() {
    unsigned g = gen_;
    if (++count_ == n_) {
      count_ = 0;
      gen_ = g + 1;
    } else {
      unsigned long long spins = 0;
      while (gen_ == g) {
        yield();
        if (++spins > 50000000ull) stuck("barrier", count_, n_);  // a barrier that never completes is a bug in a kernel or in this emulation
      }
    }
  }
 private:
  unsigned n_ = 1;
  volatile unsigned count_ = 0, gen_ = 0;
};

struct Wave {
  Barrier bar;
  unsigned long long mask = 0;
  unsigned long long shfl[64];
  unsigned lanes = 64;
};

struct Runtime {
  static constexpr unsigned MAXT = 1024;
  Barrier block_bar;
  Wave waves[MAXT / 64];
  void run(dim3 grid, dim3 block, const std::function<void()>& fn);
};
Runtime& rt();
extern thread_local Wave* cur_wave;
extern thread_local unsigned cur_lane;
extern thread_local bool in_coop;

}  // namespace emu

// Cooperative launch: kernels that use __syncthreads / wave intrinsics. Simple launch: thread-independent
// kernels (plus atomics), run as a plain loop on the calling thread (much faster under emulation).
template <class K, class... A>
inline void emu_launch(bool coop, K kernel, dim3 grid, dim3 block, A... args) {
  if (coop) {
    emu::rt().run(grid, block, [=]() { kernel(args...); });
  } else {
    gridDim = grid; blockDim = block; emu::in_coop = false;
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
      blockIdx = {bx, by, bz};
      for (unsigned tz = 0; tz < block.z; tz++) for (unsigned ty = 0; ty < block.y; ty++) for (unsigned tx = 0; tx < block.x; tx++) {
        threadIdx = {tx, ty, tz};
        kernel(args...);
      }
    }
  }
}
// The engine launches every kernel through one of these two macros (see am355_device.h).
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_current_kernel = #kernel, emu_launch(true, kernel, dim3(grid), dim3(block), __VA_ARGS__)
#define AM355_LAUNCH_INDEPENDENT(kernel, grid, block, stream, ...) emu_launch(false, kernel, dim3(grid), dim3(block), __VA_ARGS__)

inline void emu_require_coop(const char* what) {
  if (!emu::in_coop) { fprintf(stderr, "emu: %s used in a kernel launched with AM355_LAUNCH_INDEPENDENT\n", what); abort(); }
}
inline void __syncthreads() { emu_require_coop("__syncthreads"); emu::rt().block_bar.wait(); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_system() { std::atomic_thread_fence(std::memory_order_seq_cst); }

inline unsigned long long __ballot(int pred) {
  emu_require_coop("__ballot");
  emu::Wave* w = emu::cur_wave;
  if (emu::cur_lane == 0) w->mask = 0;
  w->bar.wait();
  if (pred) w->mask |= 1ull << emu::cur_lane;
  w->bar.wait();
  unsigned long long m = w->mask;
  w->bar.wait();
  return m;
}
template <class T>
inline T __shfl(T v, int src_lane, int width = 64) {
  emu_require_coop("__shfl");
  static_assert(sizeof(T) <= 8, "shfl of wide type");
  emu::Wave* w = emu::cur_wave;
  unsigned long long bits = 0;
  memcpy(&bits, &v, sizeof(T));
  w->shfl[emu::cur_lane] = bits;
  w->bar.wait();
  int base = (int)emu::cur_lane / width * width;
  int src = base + ((src_lane % width) + width) % width;
  unsigned long long r = w->shfl[src < (int)w->lanes ? src : emu::cur_lane];
  w->bar.wait();
  T out;
  memcpy(&out, &r, sizeof(T));
  return out;
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  int lane = (int)emu::cur_lane % width;
  T r = __shfl(v, lane - (int)d < 0 ? lane : lane - (int)d, width);
  return r;
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = (int)emu::cur_lane % width;
  return __shfl(v, lane + (int)d >= width ? lane : lane + (int)d, width);
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) { return __shfl(v, ((int)emu::cur_lane % width) ^ m, width); }
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_wave_barrier() { if (emu::in_coop) emu::cur_wave->bar.wait(); }
inline unsigned __lane_id() { return emu::in_coop ? emu::cur_lane : (threadIdx.x & 63); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }

// ---- atomics -----------------------------------------------------------------------------------------
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T> inline T __hip_atomic_load(const T* p, int order, int) { return __atomic_load_n(p, order); }
template <class T> inline void __hip_atomic_store(T* p, T v, int order, int) { __atomic_store_n(p, v, order); }
inline void __builtin_amdgcn_s_sleep(int) { if (emu::in_coop) emu::yield(); else std::this_thread::yield(); }
template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
template <class T> inline T atomicMax(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <class T> inline T atomicMin(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
