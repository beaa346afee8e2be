// Workgroup-level scan / reduction building blocks and the "carried" device-wide scan (device code, header only).
//
// wave64: a scan over the 64 lanes is six __shfl_up steps in registers; the four waves of a 256-thread workgroup exchange
// their totals through four LDS words (two barriers) -- against sixteen barriers of a Hillis-Steele pass over LDS.
//
// Carried scan: most prefix sums of the replay sit between a kernel that produces 0/1 flags (one per thread) and a kernel
// that consumes the positions. Launched on their own, such a scan is two or three more launches for an array that streams
// in a microsecond. Instead the PRODUCER publishes its workgroup's sum (carry_publish: one word per workgroup + an atomic add
// into the sum of its group of 64 workgroups) and the CONSUMER -- same grid geometry, one element per thread -- rebuilds its
// exclusive prefix from the group sums before its group, the workgroup sums before it inside its group, and a workgroup
// scan (carry_prefix). No extra launch; a few hundred extra words read per workgroup, all L2 hits.
#pragma once
#include "am355_device.h"

namespace am355 {

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x, uint32_t lane) {
  for (int d = 1; d < WAVE; d <<= 1) {
    uint32_t y = __shfl_up(x, (unsigned)d);
    if (lane >= (uint32_t)d) x += y;
  }
  return x;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
  for (int d = WAVE / 2; d >= 1; d >>= 1) x += __shfl_xor(x, d);
  return x;
}

// sum over the workgroup (BLOCK threads), returned to every thread. s: BLOCK / WAVE words of LDS. Every thread must call it.
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t* s) {
  uint32_t w = threadIdx.x / WAVE;
  uint32_t t = wave_sum_u32(v);
  if ((threadIdx.x & (WAVE - 1)) == 0) s[w] = t;
  __syncthreads();
  uint32_t total = 0;
  for (int k = 0; k < BLOCK / WAVE; k++) total += s[k];
  __syncthreads();
  return total;
}

// exclusive prefix of v over the workgroup; *total = workgroup sum. s: BLOCK / WAVE words of LDS. Every thread must call it.
__device__ __forceinline__ uint32_t block_exclusive_scan_u32(uint32_t v, uint32_t* s, uint32_t* total) {
  uint32_t lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  uint32_t incl = wave_incl_scan_u32(v, lane);
  if (lane == WAVE - 1) s[w] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (uint32_t k = 0; k < BLOCK / WAVE; k++) {
    uint32_t x = s[k];
    before += k < w ? x : 0;
    all += x;
  }
  __syncthreads();
  *total = all;
  return before + incl - v;
}

// ---- carried scan ----
constexpr uint32_t CARRY_GROUP_SHIFT = 6;  // 64 workgroups (16384 elements) per group sum
// Every group sum sits in its own 256-byte stretch: device-scope atomics on one word (and on words of one line) execute one after
// another at ~12 ns each (MI355X_MICROARCH.md, "fanin"), so the sums of different groups must not share a line -- 64 arrivals
// per word, all groups in parallel.
constexpr uint32_t CARRY_GROUP_STRIDE = 64;  // words
struct CarryScan {
  uint32_t* wg_sum;     // [workgroups]
  uint32_t* group_sum;  // [(workgroups >> CARRY_GROUP_SHIFT) + 1] x CARRY_GROUP_STRIDE words, zeroed before the producer runs
};
__host__ __device__ __forceinline__ size_t carry_words(uint32_t n_elems) {  // words of one CarryScan over n_elems elements
  size_t wgs = ((size_t)n_elems + BLOCK - 1) / BLOCK + 1;
  return wgs + (wgs >> CARRY_GROUP_SHIFT) + 2;
}
// producer side: v = this thread's value. Returns the workgroup sum. Every thread of the workgroup must call it.
__device__ __forceinline__ uint32_t carry_publish(const CarryScan& c, uint32_t v, uint32_t* s) {
  uint32_t total = block_sum_u32(v, s);
  if (threadIdx.x == 0) {
    c.wg_sum[blockIdx.x] = total;
    if (total) atomicAdd(&c.group_sum[(size_t)(blockIdx.x >> CARRY_GROUP_SHIFT) * CARRY_GROUP_STRIDE], total);
  }
  return total;
}
// consumer side (same grid as the producer): exclusive prefix of v over the whole grid. Every thread must call it.
__device__ __forceinline__ uint32_t carry_prefix(const CarryScan& c, uint32_t v, uint32_t* s) {
  uint32_t g = blockIdx.x >> CARRY_GROUP_SHIFT, first = g << CARRY_GROUP_SHIFT;
  uint32_t part = 0;
  for (uint32_t k = threadIdx.x; k < g; k += BLOCK) part += c.group_sum[(size_t)k * CARRY_GROUP_STRIDE];
  if (first + threadIdx.x < blockIdx.x) part += c.wg_sum[first + threadIdx.x];  // (at most 63 workgroups before this one in its group)
  uint32_t before = block_sum_u32(part, s);
  uint32_t total;
  return before + block_exclusive_scan_u32(v, s, &total);
}

}  // namespace am355
