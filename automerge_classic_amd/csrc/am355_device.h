// Device-side common definitions for the MI355X replay engine (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Kernels whose threads never synchronise (no __syncthreads, no wave intrinsics) are launched through this
// macro; on the GPU it is an ordinary launch. (The CPU test harness in tests/emu/ supplies its own definition
// so such kernels can be run as a plain loop; that harness is not part of the product.)
#ifndef AM355_LAUNCH_INDEPENDENT
#define AM355_LAUNCH_INDEPENDENT(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
#endif

// Commands two host threads enqueue into two streams are ordered by the events between the streams: the replay lets a helper thread
// enqueue the hash stream's commands beside the calling thread (am355_replay.hip). (A runtime whose launches run to their end inside
// the call has no such order and defines this 0 -- the CPU test harness in tests/emu/ does.)
#ifndef AM355_STREAMS_ORDER_ACROSS_THREADS
#define AM355_STREAMS_ORDER_ACROSS_THREADS 1
#endif

// occupancy floor of a kernel (wavefronts per SIMD the register allocation must allow): a device-compiler attribute
#if defined(__HIP_DEVICE_COMPILE__)
#define AM355_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#else
#define AM355_WAVES_PER_EU(n)
#endif

namespace am355 {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr int WAVE = 64;
constexpr int BLOCK = 256;  // 4 waves: one per SIMD of a CU

// validity / support flags raised by kernels (OR-ed into one word; any bit set => the caller must not trust
// the result and the JS host re-runs the call on the reference path to raise the exact exception)
enum Flag : uint32_t {
  F_BAD_MAGIC = 1u << 0,        // columnar.js:689
  F_BAD_CHECKSUM = 1u << 1,     // columnar.js:703
  F_BAD_CHUNK = 1u << 2,        // columnar.js:746-747 (trailing data / wrong chunk type / truncated)
  F_BAD_COLUMNS = 1u << 3,      // columnar.js:609-624, 752-754
  F_BAD_LEB = 1u << 4,          // encoding.js:389-408 range / truncation
  F_BAD_RLE = 1u << 5,          // encoding.js:865-887 run rules
  F_BAD_ROW = 1u << 6,          // new.js:715-723 obj/key pairing, actor index out of range
  F_UNKNOWN_OBJECT = 1u << 7,   // op names an object that no applied make op created
  F_BAD_ELEM = 1u << 8,         // new.js:278,1165 reference element / list element not found
  F_BAD_PRED = 1u << 9,         // new.js:1256 no matching operation for pred
  F_DUP_OPID = 1u << 10,        // new.js:1220
  F_BAD_COUNTER = 1u << 11,     // new.js:954-956 increment for unknown counter
  F_UNSUPPORTED = 1u << 12,     // legal input outside what the GPU path serves (documented in DESIGN.md)
  F_OVERFLOW = 1u << 13,        // counters/offsets beyond the engine's 32-bit fields
  F_UNKNOWN_ACTOR_DEV = 1u << 17,  // new.js:1442-1449 actorId not known to document (== AM355_F_UNKNOWN_ACTOR)
};

// Wave issue priority inside a SIMD (s_setprio, 0..3). The latency-bound one-wave-per-change kernels of the critical path raise
// it so that ALU-dense waves of the hash stream sharing their SIMD do not stretch them (queue priorities alone do not
// guarantee that on every runtime).
__device__ __forceinline__ void wave_priority_high() { __builtin_amdgcn_s_setprio(3); }

// Byte pointer into LDS with its address space in the type. A plain `const uint8_t*` that may point to LDS or to global
// memory is a generic pointer: every load through it is a FLAT instruction (slow path, both counters), which is what a
// lane-serial parser staged in LDS must not pay per byte. (The CPU emulation of tests/emu defines it as a plain pointer.)
#ifndef AM355_LDS_BYTES_DEFINED
typedef const __attribute__((address_space(3))) uint8_t* LdsBytes;
#define AM355_LDS_IS_DISTINCT 1
#endif

__device__ __forceinline__ uint32_t gtid() { return blockIdx.x * blockDim.x + threadIdx.x; }

// Publishes `n_words` result words and then the sequence number into pinned host memory (HostSignals, am355_internal.h): called by
// ONE thread, after everything the words summarise has been written by this or an earlier kernel.
__device__ __forceinline__ void signal_host(uint32_t* host_words, const uint32_t* values, uint32_t n_words, volatile uint32_t* host_seq, uint32_t seq) {
  for (uint32_t k = 0; k < n_words; k++) host_words[k] = values[k];
  __threadfence_system();
  *host_seq = seq;
}

}  // namespace am355
