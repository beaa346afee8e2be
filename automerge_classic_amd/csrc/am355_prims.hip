// Device-wide primitives for the replay engine: exclusive scan and stable LSD radix sort of (u64 key, u32 value)
// pairs. Hand-written for gfx950: 256-thread workgroups (one wave per SIMD), 64-lane ballot-based digit
// ranking, LDS histograms. These are HBM-bound integer kernels (no MFMA): each sort pass reads and writes
// every pair once (12 B in, 12 B out) plus one histogram read of the keys.
#include "am355_prims.h"
#include <algorithm>
#include "am355_scan.h"
#include "am355_canary.h"
#include <mutex>
#include <string>
#include <vector>

namespace am355 {

// ---------------------------------------------------------------------------------------------------------
// exclusive scan (uint32 -> uint32), n up to 2^32-1.
//   n <= SCAN_SINGLE          one launch: a single workgroup walks the array with a running carry
//   n <= SCAN_TILE * 1024     two launches: per-tile sums; apply, where every workgroup first sums the (at most 1024) tile sums
//                             before its own -- cheaper than a third launch for arrays that stream in a few microseconds
//   larger                    three launches: per-tile sums, scan of the tile sums by one workgroup, apply
// The workgroup scan is a wave scan (__shfl_up, log2 64 steps in registers) plus one LDS exchange of the four wave totals.
// (Standalone scans; producers / consumers that can carry the tile sums themselves use am355_scan.h instead.)
// ---------------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;
constexpr uint32_t SCAN_SINGLE = 4 * SCAN_TILE;
constexpr uint32_t SCAN_TWO_LAUNCH_TILES = 1024;

__global__ __launch_bounds__(BLOCK) void k_scan_tile_sums(const uint32_t* __restrict__ in, uint32_t* __restrict__ tile_sums, uint32_t n) {
  __shared__ uint32_t s[BLOCK / WAVE];
  uint32_t base = blockIdx.x * SCAN_TILE;
  uint32_t v[SCAN_ITEMS];  // (all loads requested before the first is used: a branch around each made them one round trip apiece)
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    uint32_t i = base + j * BLOCK + threadIdx.x;
    v[j] = i < n ? in[i] : 0u;
  }
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) sum += v[j];
  uint32_t total = block_sum_u32(sum, s);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of the tile sums in place; writes grand total
__global__ __launch_bounds__(BLOCK) void k_scan_sums(uint32_t* __restrict__ sums, uint32_t n_tiles, uint32_t* __restrict__ grand_total) {
  __shared__ uint32_t s[BLOCK / WAVE];
  uint32_t carry = 0;
  // every thread takes SUMS_PER consecutive sums per round (all loaded before any is used): a quarter of the rounds -- each a load, a
  // workgroup scan with two barriers and a store -- of one sum per thread
  constexpr uint32_t SUMS_PER = 4;
  for (uint32_t base = 0; base < n_tiles; base += BLOCK * SUMS_PER) {
    const uint32_t i0 = base + threadIdx.x * SUMS_PER;
    uint32_t v[SUMS_PER], mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < SUMS_PER; k++) {
      v[k] = i0 + k < n_tiles ? sums[i0 + k] : 0u;
      mine += v[k];
    }
    uint32_t total;
    uint32_t ex = block_exclusive_scan_u32(mine, s, &total) + carry;
#pragma unroll
    for (uint32_t k = 0; k < SUMS_PER; k++) {
      if (i0 + k < n_tiles) sums[i0 + k] = ex;
      ex += v[k];
    }
    carry += total;
  }
  if (threadIdx.x == 0 && grand_total) *grand_total = carry;
}

// PREFIXED: tile_sums[] already holds exclusive prefixes (three-launch form); else the workgroup sums the tile sums before its own
template <bool PREFIXED>
__global__ __launch_bounds__(BLOCK) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                      const uint32_t* __restrict__ tile_sums, uint32_t n, uint32_t* __restrict__ grand_total) {
  __shared__ uint32_t s[BLOCK / WAVE];
  uint32_t before;
  if (PREFIXED) before = tile_sums[blockIdx.x];
  else {
    uint32_t part = 0;
    for (uint32_t k = threadIdx.x; k < blockIdx.x; k += BLOCK) part += tile_sums[k];
    before = block_sum_u32(part, s);
  }
  // thread t owns SCAN_ITEMS consecutive elements so the in-thread prefix is sequential; full stretches of 16-byte aligned arrays
  // move as two 16-byte loads and two 16-byte stores per thread instead of eight 4-byte ones at a 32-byte stride
  static_assert(SCAN_ITEMS == 8, "k_scan_apply moves a thread's stretch as two uint4");
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  const bool wide = ((((uintptr_t)in | (uintptr_t)out) & 15) == 0) && base + SCAN_ITEMS <= n;
  uint32_t v[SCAN_ITEMS];
  if (wide) {
    const uint4 a = *(const uint4*)(in + base), b4 = *(const uint4*)(in + base + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
      uint32_t i = base + j;
      v[j] = i < n ? in[i] : 0;
    }
  }
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) sum += v[j];
  uint32_t total;
  uint32_t ex = block_exclusive_scan_u32(sum, s, &total) + before;
  if (!PREFIXED && grand_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *grand_total = before + total;
  uint32_t o[SCAN_ITEMS];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    o[j] = ex;
    ex += v[j];
  }
  if (wide) {
    *(uint4*)(out + base) = uint4{o[0], o[1], o[2], o[3]};
    *(uint4*)(out + base + 4) = uint4{o[4], o[5], o[6], o[7]};
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
      uint32_t i = base + j;
      if (i < n) out[i] = o[j];
    }
  }
}

// ---- single pass with decoupled look-back (the large regime: more than SCAN_TWO_LAUNCH_TILES tiles) ----
// Three launches move 12 bytes per element (the sums kernel reads it, apply reads it again and writes); a document load runs
// eighteen such scans over tens of millions of positions: 3 GB of its 17 GB. Here a tile is read ONCE: the workgroup publishes the sum
// of its tile (one 64-bit word: state << 32 | value, state 1 = the tile's own sum, 2 = sum of everything up to and including it), its
// first wavefront looks back over the tiles in front -- 64 words per step, one lane each -- adding sums until it meets a state-2 word,
// publishes its own state-2 word and the tile is written. Tiles are handed out by a ticket, so every tile in front of a running one has
// been started; words are exchanged with device-scope atomics (the XCDs have L2 caches of their own). 8 bytes per element, one launch
// + one small memset of the words.
constexpr unsigned long long LB_SUM = 1ull << 32, LB_PREFIX = 2ull << 32;
// one tile word, device scope: a plain 64-bit load / store that bypasses the XCD-local caches (a read-modify-write "add 0" was measured
// first: 64 lanes of every tile hammering the atomic units made the scan slower than its three-launch form)
__device__ __forceinline__ unsigned long long lb_load(const unsigned long long* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *(const volatile unsigned long long*)p;
#endif
}
__device__ __forceinline__ void lb_store(unsigned long long* p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *(volatile unsigned long long*)p = v;
#endif
}

__global__ __launch_bounds__(BLOCK) void k_scan_lookback(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, unsigned long long* __restrict__ state,
                                                         uint32_t* __restrict__ ticket, uint32_t* __restrict__ grand_total) {
  __shared__ uint32_t s[BLOCK / WAVE];
  __shared__ uint32_t s_tile, s_before;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint32_t tile = s_tile;
  static_assert(SCAN_ITEMS == 8, "a thread's stretch moves as two uint4");
  const uint32_t base = tile * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  const bool wide = ((((uintptr_t)in | (uintptr_t)out) & 15) == 0) && base + SCAN_ITEMS <= n;
  uint32_t v[SCAN_ITEMS];
  if (wide) {
    const uint4 a = *(const uint4*)(in + base), b4 = *(const uint4*)(in + base + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
      uint32_t i = base + j;
      v[j] = i < n ? in[i] : 0;
    }
  }
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) sum += v[j];
  uint32_t total;
  uint32_t ex = block_exclusive_scan_u32(sum, s, &total);
  if (threadIdx.x == 0) {
    lb_store(&state[tile], (tile ? LB_SUM : LB_PREFIX) | total);
    s_before = 0;
  }
  if (tile && threadIdx.x < WAVE) {
    const uint32_t lane = threadIdx.x;
    uint32_t before = 0;
    int look = (int)tile - 1;  // lane L inspects tile look - L
    for (;;) {
      const int idx = look - (int)lane;
      const unsigned long long w = idx >= 0 ? lb_load(&state[idx]) : LB_PREFIX;  // (in front of tile 0: nothing, known)
      const uint32_t st = (uint32_t)(w >> 32);
      const unsigned long long pending = __ballot(st == 0), prefix = __ballot(st == 2);
      const uint32_t first = prefix ? (uint32_t)__ffsll((long long)prefix) - 1 : WAVE;   // nearest tile whose word already sums everything in front of it
      const unsigned long long needed = first >= WAVE - 1 ? ~0ull : ((2ull << first) - 1);
      if (pending & needed) continue;  // a tile between here and there has not published yet: read again
      uint32_t part = lane <= first ? (uint32_t)w : 0u;
      for (int d = WAVE / 2; d; d >>= 1) part += __shfl_xor(part, d);
      before += part;
      if (prefix) break;
      look -= WAVE;
    }
    if (lane == 0) {
      s_before = before;
      lb_store(&state[tile], LB_PREFIX | (before + total));
    }
  }
  __syncthreads();
  const uint32_t before = s_before;
  if (grand_total && (size_t)(tile + 1) * SCAN_TILE >= n && threadIdx.x == 0) *grand_total = before + total;
  ex += before;
  uint32_t o[SCAN_ITEMS];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    o[j] = ex;
    ex += v[j];
  }
  if (wide) {
    *(uint4*)(out + base) = uint4{o[0], o[1], o[2], o[3]};
    *(uint4*)(out + base + 4) = uint4{o[4], o[5], o[6], o[7]};
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
      uint32_t i = base + j;
      if (i < n) out[i] = o[j];
    }
  }
}

// one workgroup, one launch: tiles in sequence with a running carry
__global__ __launch_bounds__(BLOCK) void k_scan_single(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, uint32_t* __restrict__ grand_total) {
  wave_priority_high();
  __shared__ uint32_t s[BLOCK / WAVE];
  uint32_t carry = 0;
  for (uint32_t tile = 0; tile < n; tile += SCAN_TILE) {
    uint32_t base = tile + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t sum = 0;
    for (int j = 0; j < SCAN_ITEMS; j++) {
      uint32_t i = base + j;
      v[j] = i < n ? in[i] : 0;
      sum += v[j];
    }
    uint32_t total;
    uint32_t ex = block_exclusive_scan_u32(sum, s, &total) + carry;
    for (int j = 0; j < SCAN_ITEMS; j++) {
      uint32_t i = base + j;
      if (i < n) out[i] = ex;
      ex += v[j];
    }
    carry += total;
  }
  if (threadIdx.x == 0 && grand_total) *grand_total = carry;
}

// Two independent scans over the same index range in one pass (same launch structure as the single scan).
__global__ __launch_bounds__(BLOCK) void k_scan2_tile_sums(const uint32_t* __restrict__ in_a, const uint32_t* __restrict__ in_b, uint32_t* __restrict__ sums_a,
                                                          uint32_t* __restrict__ sums_b, uint32_t n) {
  __shared__ uint32_t s[BLOCK / WAVE];
  uint32_t base = blockIdx.x * SCAN_TILE;
  uint32_t sa = 0, sb = 0;
  for (int j = 0; j < SCAN_ITEMS; j++) {
    uint32_t i = base + j * BLOCK + threadIdx.x;
    if (i < n) { sa += in_a[i]; sb += in_b[i]; }
  }
  uint32_t ta = block_sum_u32(sa, s), tb = block_sum_u32(sb, s);
  if (threadIdx.x == 0) { sums_a[blockIdx.x] = ta; sums_b[blockIdx.x] = tb; }
}
__global__ __launch_bounds__(BLOCK) void k_scan2_sums(uint32_t* __restrict__ sums_a, uint32_t* __restrict__ sums_b, uint32_t n_tiles, uint32_t* __restrict__ total_a,
                                                     uint32_t* __restrict__ total_b) {
  __shared__ uint32_t s[BLOCK / WAVE];
  uint32_t ca = 0, cb = 0;
  for (uint32_t base = 0; base < n_tiles; base += BLOCK) {
    uint32_t i = base + threadIdx.x;
    uint32_t va = i < n_tiles ? sums_a[i] : 0, vb = i < n_tiles ? sums_b[i] : 0;
    uint32_t ta, tb;
    uint32_t ea = block_exclusive_scan_u32(va, s, &ta);
    uint32_t eb = block_exclusive_scan_u32(vb, s, &tb);
    if (i < n_tiles) { sums_a[i] = ca + ea; sums_b[i] = cb + eb; }
    ca += ta;
    cb += tb;
  }
  if (threadIdx.x == 0) {
    if (total_a) *total_a = ca;
    if (total_b) *total_b = cb;
  }
}
template <bool PREFIXED>
__global__ __launch_bounds__(BLOCK) void k_scan2_apply(const uint32_t* __restrict__ in_a, uint32_t* __restrict__ out_a, const uint32_t* __restrict__ in_b,
                                                      uint32_t* __restrict__ out_b, const uint32_t* __restrict__ sums_a, const uint32_t* __restrict__ sums_b, uint32_t n,
                                                      uint32_t* __restrict__ total_a, uint32_t* __restrict__ total_b) {
  __shared__ uint32_t s[BLOCK / WAVE];
  uint32_t before_a, before_b;
  if (PREFIXED) { before_a = sums_a[blockIdx.x]; before_b = sums_b[blockIdx.x]; }
  else {
    uint32_t pa = 0, pb = 0;
    for (uint32_t k = threadIdx.x; k < blockIdx.x; k += BLOCK) { pa += sums_a[k]; pb += sums_b[k]; }
    before_a = block_sum_u32(pa, s);
    before_b = block_sum_u32(pb, s);
  }
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t va[SCAN_ITEMS], vb[SCAN_ITEMS];
  uint32_t sa = 0, sb = 0;
  for (int j = 0; j < SCAN_ITEMS; j++) {
    uint32_t i = base + j;
    va[j] = i < n ? in_a[i] : 0;
    vb[j] = i < n ? in_b[i] : 0;
    sa += va[j];
    sb += vb[j];
  }
  uint32_t ta, tb;
  uint32_t ea = block_exclusive_scan_u32(sa, s, &ta) + before_a;
  uint32_t eb = block_exclusive_scan_u32(sb, s, &tb) + before_b;
  if (!PREFIXED && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    if (total_a) *total_a = before_a + ta;
    if (total_b) *total_b = before_b + tb;
  }
  for (int j = 0; j < SCAN_ITEMS; j++) {
    uint32_t i = base + j;
    if (i < n) { out_a[i] = ea; out_b[i] = eb; }
    ea += va[j];
    eb += vb[j];
  }
}

void exclusive_scan2_u32(const uint32_t* in_a, uint32_t* out_a, uint32_t* d_total_a, const uint32_t* in_b, uint32_t* out_b, uint32_t* d_total_b, uint32_t n,
                         void* ws, hipStream_t st) {
  uint32_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (n_tiles == 0) {
    if (d_total_a) (void)hipMemsetAsync(d_total_a, 0, sizeof(uint32_t), st);
    if (d_total_b) (void)hipMemsetAsync(d_total_b, 0, sizeof(uint32_t), st);
    return;
  }
  uint32_t* sums_a = (uint32_t*)ws;
  uint32_t* sums_b = sums_a + n_tiles + 1;
  hipLaunchKernelGGL(k_scan2_tile_sums, dim3(n_tiles), dim3(BLOCK), 0, st, in_a, in_b, sums_a, sums_b, n);
  if (n_tiles <= SCAN_TWO_LAUNCH_TILES) {
    hipLaunchKernelGGL(k_scan2_apply<false>, dim3(n_tiles), dim3(BLOCK), 0, st, in_a, out_a, in_b, out_b, (const uint32_t*)sums_a, (const uint32_t*)sums_b, n, d_total_a,
                       d_total_b);
    return;
  }
  hipLaunchKernelGGL(k_scan2_sums, dim3(1), dim3(BLOCK), 0, st, sums_a, sums_b, n_tiles, d_total_a, d_total_b);
  hipLaunchKernelGGL(k_scan2_apply<true>, dim3(n_tiles), dim3(BLOCK), 0, st, in_a, out_a, in_b, out_b, (const uint32_t*)sums_a, (const uint32_t*)sums_b, n,
                     (uint32_t*)nullptr, (uint32_t*)nullptr);
}

// ---- exclusive scan of the LEB128 terminator predicate of a byte string ----
// element i (i < L) = 1 if byte i has bit 7 clear (it ends a number), element L = 0; out has L + 1 entries. The flags are never
// stored: the two passes read the bytes themselves (1 byte per element instead of a 4-byte flag written once and read twice).
__device__ __forceinline__ void term_load(const uint8_t* __restrict__ bytes, uint32_t L, uint32_t base, uint32_t (&v)[SCAN_ITEMS]) {
  static_assert(SCAN_ITEMS == 8, "a thread's stretch is one 8-byte load");
  if ((((uintptr_t)(bytes + base)) & 7) == 0 && base + SCAN_ITEMS <= L) {
    const unsigned long long x = *(const unsigned long long*)(bytes + base);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) v[j] = (uint32_t)((x >> (8 * j + 7)) & 1ull) ^ 1u;
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) { uint32_t i = base + j; v[j] = i < L ? ((bytes[i] >> 7) ^ 1u) : 0u; }
  }
}
__global__ __launch_bounds__(BLOCK) void k_scan_term_sums(const uint8_t* __restrict__ bytes, uint32_t L, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t s[BLOCK / WAVE];
  uint32_t v[SCAN_ITEMS];
  term_load(bytes, L, blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS, v);
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) sum += v[j];
  uint32_t total = block_sum_u32(sum, s);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
// (tile_sums: exclusive prefixes, by k_scan_sums)
__global__ __launch_bounds__(BLOCK) void k_scan_term_apply(const uint8_t* __restrict__ bytes, uint32_t L, uint32_t* __restrict__ out, const uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t s[BLOCK / WAVE];
  const uint32_t n = L + 1;
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  term_load(bytes, L, base, v);
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) sum += v[j];
  uint32_t total;
  uint32_t ex = block_exclusive_scan_u32(sum, s, &total) + tile_sums[blockIdx.x];
  uint32_t o[SCAN_ITEMS];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) { o[j] = ex; ex += v[j]; }
  if ((((uintptr_t)(out + base)) & 15) == 0 && base + SCAN_ITEMS <= n) {
    *(uint4*)(out + base) = uint4{o[0], o[1], o[2], o[3]};
    *(uint4*)(out + base + 4) = uint4{o[4], o[5], o[6], o[7]};
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n) out[base + j] = o[j];
  }
}
void exclusive_scan_terminators(const uint8_t* bytes, uint32_t L, uint32_t* out, uint32_t* d_total, void* ws, hipStream_t st) {
  const uint32_t n = L + 1, n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t* sums = (uint32_t*)ws;
  hipLaunchKernelGGL(k_scan_term_sums, dim3(n_tiles), dim3(BLOCK), 0, st, bytes, L, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLOCK), 0, st, sums, n_tiles, d_total);
  hipLaunchKernelGGL(k_scan_term_apply, dim3(n_tiles), dim3(BLOCK), 0, st, bytes, L, out, (const uint32_t*)sums);
}

size_t scan_workspace_bytes(uint32_t n) { return 2 * sizeof(uint32_t) * ((size_t)(n + SCAN_TILE - 1) / SCAN_TILE + 2); }  // (room for a dual scan)

void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* d_total, void* ws, hipStream_t st) {
  uint32_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t* sums = (uint32_t*)ws;
  if (n_tiles == 0) {
    if (d_total) (void)hipMemsetAsync(d_total, 0, sizeof(uint32_t), st);
    return;
  }
  if (n <= SCAN_SINGLE) {
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(BLOCK), 0, st, in, out, n, d_total);
    return;
  }
  if (n_tiles <= SCAN_TWO_LAUNCH_TILES) {
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(BLOCK), 0, st, in, sums, n);
    hipLaunchKernelGGL(k_scan_apply<false>, dim3(n_tiles), dim3(BLOCK), 0, st, in, out, (const uint32_t*)sums, n, d_total);
    return;
  }
  // (measured on the config-5 load, same box, profiles/r05_ab_scan_lookback.txt: 9.05 ms with the single pass against 8.05 ms with the
  // three launches -- a tile waits for the words of the tiles in front of it, and the three streaming passes run at full bandwidth while
  // the chain of words does not. Kept behind AM355_SCAN_LOOKBACK=1, not the default.)
  const char* lb_env = getenv("AM355_SCAN_LOOKBACK");   // (read per call: the tests switch it inside one process)
  const bool lookback = lb_env && *lb_env == '1';
  if (lookback) {
    // (the workspace holds 8 bytes per tile + 16: the tile words, then the ticket)
    unsigned long long* state = (unsigned long long*)ws;
    uint32_t* ticket = (uint32_t*)(state + n_tiles);
    (void)hipMemsetAsync(ws, 0, sizeof(unsigned long long) * n_tiles + 8, st);
    hipLaunchKernelGGL(k_scan_lookback, dim3(n_tiles), dim3(BLOCK), 0, st, in, out, n, state, ticket, d_total);
    return;
  }
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(BLOCK), 0, st, in, sums, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLOCK), 0, st, sums, n_tiles, d_total);
  hipLaunchKernelGGL(k_scan_apply<true>, dim3(n_tiles), dim3(BLOCK), 0, st, in, out, (const uint32_t*)sums, n, (uint32_t*)nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// LSD radix sort, 8-bit digits, stable.  Per pass: histogram -> scan -> scatter.
// Tile = 4 waves x 64 lanes x SORT_ITEMS; each wave owns a contiguous quarter of the tile so that
// (wave, item, lane) order equals memory order, which is what makes the pass stable.
// ---------------------------------------------------------------------------------------------------------
constexpr int SORT_ITEMS = 8;
constexpr int SORT_WAVE_SPAN = WAVE * SORT_ITEMS;          // 512 elements per wave
constexpr int SORT_TILE = (BLOCK / WAVE) * SORT_WAVE_SPAN;  // 2048 elements per workgroup
constexpr int RADIX = 256;
constexpr uint32_t SORT_FUSED_TILES = 64;

__global__ __launch_bounds__(BLOCK) void k_sort_hist(const uint64_t* __restrict__ keys, uint32_t n, int shift,
                                                      uint32_t* __restrict__ table, uint32_t n_tiles) {
  __shared__ uint32_t hist[RADIX];
  hist[threadIdx.x] = 0;
  __syncthreads();
  uint32_t base = blockIdx.x * SORT_TILE;
  for (int j = 0; j < SORT_TILE / BLOCK; j++) {
    uint32_t i = base + j * BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&hist[(uint32_t)(keys[i] >> shift) & 0xff], 1u);
  }
  __syncthreads();
  table[threadIdx.x * n_tiles + blockIdx.x] = hist[threadIdx.x];  // digit-major so one scan gives global offsets
}

// RAW_TABLE: `table` holds the histograms as k_sort_hist left them (no scan launch in between): thread t = digit t adds up its row -- the
// digit's count over all tiles, and over the tiles in front of this one -- and one workgroup scan over the rows gives the digit's base.
// For sorts of up to SORT_FUSED_TILES tiles (131 k elements): a pass is two launches instead of three.
template <bool RAW_TABLE>
__global__ __launch_bounds__(BLOCK) void k_sort_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                         uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                         int shift, const uint32_t* __restrict__ table, uint32_t n_tiles) {
  __shared__ uint32_t wave_cnt[BLOCK / WAVE][RADIX];
  __shared__ uint32_t s_red[BLOCK / WAVE];
  uint32_t t = threadIdx.x, w = t / WAVE, lane = t % WAVE;
  for (int k = 0; k < BLOCK / WAVE; k++) wave_cnt[k][t] = 0;
  __syncthreads();

  uint64_t key[SORT_ITEMS];
  uint32_t val[SORT_ITEMS], rank[SORT_ITEMS];
  uint32_t base = blockIdx.x * SORT_TILE + w * SORT_WAVE_SPAN;
  for (int j = 0; j < SORT_ITEMS; j++) {
    uint32_t i = base + j * WAVE + lane;
    bool valid = i < n;
    key[j] = valid ? keys_in[i] : 0;
    val[j] = valid ? vals_in[i] : 0;
    uint32_t d = (uint32_t)(key[j] >> shift) & 0xff;
    // lanes holding the same digit ("peers"), via 8 ballots
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < 8; b++) {
      unsigned long long m = __ballot(valid && ((d >> b) & 1));
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    uint32_t before = (uint32_t)__popcll(peers & ((1ull << lane) - 1));
    uint32_t seen = valid ? wave_cnt[w][d] : 0;   // same-digit elements of this wave in earlier items
    rank[j] = seen + before;
    __builtin_amdgcn_wave_barrier();               // every lane has read the counter before the leader bumps it
    if (valid && before == 0) wave_cnt[w][d] = seen + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // thread t now plays digit t: turn per-wave counts into global start offsets per (wave, digit)
  {
    uint32_t run;
    if (RAW_TABLE) {
      uint32_t row = 0, part = 0;
      for (uint32_t k = 0; k < n_tiles; k++) {
        uint32_t c = table[t * n_tiles + k];
        row += c;
        part += k < blockIdx.x ? c : 0u;
      }
      uint32_t total;
      run = block_exclusive_scan_u32(row, s_red, &total) + part;
    } else {
      run = table[t * n_tiles + blockIdx.x];
    }
    for (int k = 0; k < BLOCK / WAVE; k++) {
      uint32_t c = wave_cnt[k][t];
      wave_cnt[k][t] = run;
      run += c;
    }
  }
  __syncthreads();
  for (int j = 0; j < SORT_ITEMS; j++) {
    uint32_t i = base + j * WAVE + lane;
    if (i < n) {
      uint32_t d = (uint32_t)(key[j] >> shift) & 0xff;
      uint32_t pos = wave_cnt[w][d] + rank[j];
      keys_out[pos] = key[j];
      vals_out[pos] = val[j];
    }
  }
}

// maximum of a uint32 array into *out (which the caller zeroes): block reduction in LDS, one global atomic per workgroup
__global__ __launch_bounds__(BLOCK) void k_max_u32(const uint32_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ out) {
  __shared__ uint32_t smax;
  if (threadIdx.x == 0) smax = 0;
  __syncthreads();
  uint32_t m = 0;
  for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) m = v[i] > m ? v[i] : m;
  atomicMax(&smax, m);
  __syncthreads();
  if (threadIdx.x == 0 && smax) atomicMax(out, smax);
}

void max_u32(const uint32_t* v, uint32_t n, uint32_t* d_out, hipStream_t st) {
  if (!n) return;
  uint32_t blocks = (n + BLOCK - 1) / BLOCK;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_max_u32, dim3(blocks), dim3(BLOCK), 0, st, v, n, d_out);
}

size_t sort_workspace_bytes(uint32_t n) {
  uint32_t n_tiles = (n + SORT_TILE - 1) / SORT_TILE;
  size_t table = (size_t)RADIX * n_tiles;
  return sizeof(uint32_t) * table + scan_workspace_bytes((uint32_t)table) + 256;
}

// Sorts ascending by bits [begin_bit, end_bit) of the key. Buffers ping-pong; returns 0 if the result is in
// (keys_a, vals_a), 1 if in (keys_b, vals_b).
uint32_t sort_tiles(uint32_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }
bool sort_is_fused(uint32_t n) { return sort_tiles(n) <= SORT_FUSED_TILES; }
uint32_t* sort_first_table(void* ws) { return (uint32_t*)ws; }

int radix_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, uint32_t n, int begin_bit, int end_bit,
                     void* ws, hipStream_t st, bool first_hist_done) {
  if (n == 0) return 0;
  uint32_t n_tiles = (n + SORT_TILE - 1) / SORT_TILE;
  uint32_t* table = (uint32_t*)ws;
  size_t table_n = (size_t)RADIX * n_tiles;
  void* scan_ws = (void*)(table + ((table_n + 63) & ~(size_t)63));
  int cur = 0;
  for (int shift = begin_bit; shift < end_bit; shift += 8) {
    uint64_t* ki = cur ? keys_b : keys_a;
    uint32_t* vi = cur ? vals_b : vals_a;
    uint64_t* ko = cur ? keys_a : keys_b;
    uint32_t* vo = cur ? vals_a : vals_b;
    if (!(first_hist_done && shift == begin_bit)) hipLaunchKernelGGL(k_sort_hist, dim3(n_tiles), dim3(BLOCK), 0, st, ki, n, shift, table, n_tiles);
    if (n_tiles <= SORT_FUSED_TILES) {
      hipLaunchKernelGGL(k_sort_scatter<true>, dim3(n_tiles), dim3(BLOCK), 0, st, ki, vi, ko, vo, n, shift, table, n_tiles);
    } else {
      exclusive_scan_u32(table, table, (uint32_t)table_n, nullptr, scan_ws, st);
      hipLaunchKernelGGL(k_sort_scatter<false>, dim3(n_tiles), dim3(BLOCK), 0, st, ki, vi, ko, vo, n, shift, table, n_tiles);
    }
    cur ^= 1;
  }
  return cur;
}


// ---------------------------------------------------------------------------------------------------------
// forward chain marking. next[i] is a position > i (or >= n / NONE32 = end of chain); mark[] holds start nodes; on
// return mark[] is nonzero on every node reachable from a start. (Record headers of a column are the orbit of its first
// number under "skip this record"; document columns: am355_bigcol.hip.)
//
// Plain pointer doubling costs log2(n) passes over all n nodes. Chains only move forward, so they are cut into tiles
// of CH_TILE positions instead:
//   1. per tile, in LDS: exit[i] = first position outside the tile on i's chain (pointer doubling inside the tile).
//      The distinct exit targets are few per tile (chains merge within a few hops) -- they are the only positions
//      through which a chain can enter a later tile.
//   2. compact the exit targets (flag + prefix sum) into M << n nodes linked by exit[]; mark the ones reachable from
//      the starts' exits by pointer doubling over M nodes, log2(n / CH_TILE) rounds.
//   3. per tile, in LDS again: marks spread from the tile's start nodes and marked entry nodes along next[].
// HBM traffic: a handful of passes over n instead of log2(n).
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t CH_TILE = 4096;
constexpr uint32_t CH_PER = CH_TILE / BLOCK;
constexpr int CH_ROUNDS = 12;  // 2^12 = CH_TILE

// flagw / cstartw: one BIT per position (entry node / entry node reached from a start of an earlier tile); the caller zeroes them
__global__ __launch_bounds__(BLOCK) void kc_tile_exits(const uint32_t* __restrict__ next, uint32_t n, const uint32_t* __restrict__ mark,
                                                       uint32_t* __restrict__ exit1, uint32_t* __restrict__ flagw, uint32_t* __restrict__ cstartw) {
  __shared__ uint32_t cur[CH_TILE];
  uint32_t base = blockIdx.x * CH_TILE, t = threadIdx.x;
  uint32_t end = base + CH_TILE < n ? base + CH_TILE : n;
  for (uint32_t k = 0; k < CH_PER; k++) {
    uint32_t l = t + k * BLOCK, g = base + l;
    uint32_t c = g < n ? next[g] : NONE32;
    c = (c >= n || c <= g) ? NONE32 : c;  // a chain that does not move forward ends (callers guarantee next > i)
    cur[l] = c;
    // the distinct exit targets of the tile are the successors of the LAST in-tile node of every chain: one atomic per chain that
    // leaves the tile, not one store per position
    if (c != NONE32 && c >= end) atomicOr(&flagw[c >> 5], 1u << (c & 31));
  }
  __syncthreads();
  // in place: every value is always some node further along the same chain, so stale reads only slow the doubling down
  for (int r = 0; r < CH_ROUNDS; r++) {
    for (uint32_t k = 0; k < CH_PER; k++) {
      uint32_t l = t + k * BLOCK;
      uint32_t c = cur[l];
      if (c < end) cur[l] = cur[c - base];
    }
    __syncthreads();
  }
  for (uint32_t k = 0; k < CH_PER; k++) {
    uint32_t l = t + k * BLOCK, g = base + l;
    if (g >= n) continue;
    uint32_t c = cur[l];
    exit1[g] = c;
    if (c < n && mark[g]) atomicOr(&cstartw[c >> 5], 1u << (c & 31));
  }
}

// entry nodes per 32 positions (their ranks follow from the exclusive scan of these counts)
__global__ __launch_bounds__(BLOCK) void kc_word_counts(const uint32_t* __restrict__ flagw, uint32_t n_words, uint32_t* __restrict__ cnt) {
  uint32_t w = gtid();
  if (w < n_words) cnt[w] = (uint32_t)__popc(flagw[w]);
}

// one thread per word of the entry bitmap: node id = rank of the bit; its successor = rank of its exit target
__global__ __launch_bounds__(BLOCK) void kc_compact(uint32_t n, uint32_t n_words, const uint32_t* __restrict__ flagw, const uint32_t* __restrict__ wrank,
                                                    const uint32_t* __restrict__ exit1, const uint32_t* __restrict__ cstartw, uint32_t* __restrict__ cpos,
                                                    uint32_t* __restrict__ cnext, uint32_t* __restrict__ cmark) {
  uint32_t w = gtid();
  if (w >= n_words) return;
  uint32_t bits = flagw[w];
  if (!bits) return;
  const uint32_t starts = cstartw[w];
  uint32_t id = wrank[w];
  while (bits) {
    const uint32_t b = (uint32_t)__ffs((int)bits) - 1;
    bits &= bits - 1;
    const uint32_t i = 32 * w + b, e = exit1[i];
    cpos[id] = i;
    cnext[id] = e < n ? wrank[e >> 5] + (uint32_t)__popc(flagw[e >> 5] & ((1u << (e & 31)) - 1u)) : NONE32;
    cmark[id] = (starts >> b) & 1u;
    id++;
  }
}

__global__ __launch_bounds__(BLOCK) void kc_round(const uint32_t* __restrict__ n_nodes, const uint32_t* __restrict__ jin, uint32_t* __restrict__ jout,
                                                  uint32_t* __restrict__ cmark) {
  uint32_t m_total = *n_nodes;
  for (uint32_t m = gtid(); m < m_total; m += gridDim.x * BLOCK) {
    uint32_t j = jin[m];
    if (j != NONE32) {
      if (cmark[m]) cmark[j] = 1;
      j = jin[j];
    }
    jout[m] = j;
  }
}

__global__ __launch_bounds__(BLOCK) void kc_entries(const uint32_t* __restrict__ n_nodes, const uint32_t* __restrict__ cpos, const uint32_t* __restrict__ cmark,
                                                    uint32_t* __restrict__ mark) {
  uint32_t m_total = *n_nodes;
  for (uint32_t m = gtid(); m < m_total; m += gridDim.x * BLOCK)
    if (cmark[m] && !mark[cpos[m]]) mark[cpos[m]] = 1;
}

__global__ __launch_bounds__(BLOCK) void kc_tile_marks(const uint32_t* __restrict__ next, uint32_t n, uint32_t* __restrict__ mark) {
  __shared__ uint16_t ja[CH_TILE], jb[CH_TILE];
  __shared__ uint8_t mk[CH_TILE];
  __shared__ uint32_t any;
  uint32_t base = blockIdx.x * CH_TILE, t = threadIdx.x;
  uint32_t end = base + CH_TILE < n ? base + CH_TILE : n;
  if (t == 0) any = 0;
  __syncthreads();
  uint32_t have = 0;
  for (uint32_t k = 0; k < CH_PER; k++) {
    uint32_t l = t + k * BLOCK, g = base + l;
    uint32_t c = g < n ? next[g] : NONE32;
    ja[l] = (c < end && c > g) ? (uint16_t)(c - base) : (uint16_t)0xffff;
    uint32_t m = g < n ? mark[g] : 0;
    mk[l] = m ? 1 : 0;
    have |= m;
  }
  if (have) any = 1;
  __syncthreads();
  if (!any) return;  // no chain passes through this tile
  uint16_t *jin = ja, *jout = jb;
  for (int r = 0; r < CH_ROUNDS; r++) {
    for (uint32_t k = 0; k < CH_PER; k++) {
      uint32_t l = t + k * BLOCK;
      uint32_t j = jin[l];
      if (j != 0xffff) {
        if (mk[l]) mk[j] = 1;
        j = jin[j];
      }
      jout[l] = (uint16_t)j;
    }
    __syncthreads();
    uint16_t* s = jin;
    jin = jout;
    jout = s;
  }
  for (uint32_t k = 0; k < CH_PER; k++) {
    uint32_t l = t + k * BLOCK, g = base + l;
    if (g < n && mk[l] && !mark[g]) mark[g] = 1;
  }
}

size_t chain_work_bytes(uint32_t n) {
  size_t cap = ((size_t)n + 2 + 63) & ~(size_t)63;
  return 8 * 4 * cap + scan_workspace_bytes(n + 2) + 512;
}

void chain_mark(const uint32_t* next, uint32_t n, uint32_t* mark, void* work, hipStream_t st) {
  if (!n) return;
  size_t cap = ((size_t)n + 2 + 63) & ~(size_t)63;
  uint32_t* p = (uint32_t*)work;
  // (flagw | cstartw: bitmaps over the positions, in the first array; wcnt / wrank: per bitmap word)
  const uint32_t n_words = (uint32_t)((cap + 31) / 32);
  uint32_t *exit1 = p + cap, *flagw = p, *cstartw = p + n_words, *wrank = p + 2 * cap, *cpos = p + 4 * cap, *ca = p + 5 * cap, *cb = p + 6 * cap, *cmark = p + 7 * cap;
  uint32_t* n_nodes = p + 8 * cap;
  void* scan_ws = (void*)(n_nodes + 64);
  uint32_t tiles = (n + CH_TILE - 1) / CH_TILE;
  (void)hipMemsetAsync(flagw, 0, 2 * 4 * (size_t)n_words, st);
  hipLaunchKernelGGL(kc_tile_exits, dim3(tiles), dim3(BLOCK), 0, st, next, n, (const uint32_t*)mark, exit1, flagw, cstartw);
  AM355_LAUNCH_INDEPENDENT(kc_word_counts, dim3((n_words + BLOCK - 1) / BLOCK), dim3(BLOCK), st, (const uint32_t*)flagw, n_words, wrank);
  uint32_t* wrank_ex = p + 3 * cap;  // (its own array: the scan kernels take `in` and `out` as __restrict__)
  exclusive_scan_u32(wrank, wrank_ex, n_words, n_nodes, scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(kc_compact, dim3((n_words + BLOCK - 1) / BLOCK), dim3(BLOCK), st, n, n_words, (const uint32_t*)flagw, (const uint32_t*)wrank_ex,
                           (const uint32_t*)exit1, (const uint32_t*)cstartw, cpos, ca, cmark);
  // a compact chain visits every tile at most once
  int rounds = 1;
  while (rounds < 32 && ((tiles + 1) >> rounds)) rounds++;
  uint32_t blocks = (n / 8 + BLOCK - 1) / BLOCK;
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
  for (int r = 0; r < rounds; r++) {
    AM355_LAUNCH_INDEPENDENT(kc_round, dim3(blocks), dim3(BLOCK), st, (const uint32_t*)n_nodes, (const uint32_t*)ca, cb, cmark);
    uint32_t* t = ca;
    ca = cb;
    cb = t;
  }
  AM355_LAUNCH_INDEPENDENT(kc_entries, dim3(blocks), dim3(BLOCK), st, (const uint32_t*)n_nodes, (const uint32_t*)cpos, (const uint32_t*)cmark, mark);
  hipLaunchKernelGGL(kc_tile_marks, dim3(tiles), dim3(BLOCK), 0, st, next, n, mark);
}

// ---------------------------------------------------------------------------------------------------------
// AM355_CANARY=1: red zones behind every carve-out (am355_canary.h)
// ---------------------------------------------------------------------------------------------------------
namespace {
struct CanaryZone { unsigned long long addr; uint32_t len, tag; };
struct CanaryEntry { CanaryZone z; int device; std::string block; uint32_t index; bool armed; };
std::mutex g_canary_mu;
std::vector<CanaryEntry> g_canary;
std::string g_canary_block = "?";
uint32_t g_canary_index = 0;
constexpr uint8_t CANARY_BYTE = 0xC5;
}  // namespace

bool canary_on() {
  static const bool on = [] { const char* e = getenv("AM355_CANARY"); return e && *e && *e != '0'; }();
  return on;
}

void canary_scope(const char* block) {
  if (!canary_on()) return;
  std::lock_guard<std::mutex> g(g_canary_mu);
  g_canary_block = block;
  g_canary_index = 0;
}

static void canary_forget_locked(unsigned long long lo, unsigned long long hi, int device) {
  size_t k = 0;
  for (size_t i = 0; i < g_canary.size(); i++) {
    const CanaryEntry& e = g_canary[i];
    bool overlap = e.device == device && e.z.addr < hi && e.z.addr + e.z.len > lo;
    if (!overlap) { if (k != i) g_canary[k] = g_canary[i]; k++; }
  }
  g_canary.resize(k);
}

void canary_note(const void* base, size_t used) {
  if (!canary_on()) return;
  int device = 0;
  (void)hipGetDevice(&device);
  std::lock_guard<std::mutex> g(g_canary_mu);
  const uint32_t index = g_canary_index++;
  const unsigned long long lo = (unsigned long long)(uintptr_t)base, z0 = lo + used, z1 = lo + carve_round(used);
  for (CanaryEntry& e : g_canary)
    if (e.device == device && e.z.addr == z0 && e.z.addr + e.z.len == z1) {  // the same carve-out as last time: its zone stands (and was verified)
      e.block = g_canary_block; e.index = index;
      return;
    }
  canary_forget_locked(lo, z1, device);
  g_canary.push_back(CanaryEntry{CanaryZone{z0, (uint32_t)(z1 - z0), 0}, device, g_canary_block, index, false});
}

void canary_forget(const void* base, size_t bytes) {
  if (!canary_on() || !base) return;
  int device = 0;
  (void)hipGetDevice(&device);
  std::lock_guard<std::mutex> g(g_canary_mu);
  canary_forget_locked((unsigned long long)(uintptr_t)base, (unsigned long long)(uintptr_t)base + bytes, device);
}

void canary_allow(const void* base, size_t bytes) {
  if (!canary_on() || !base || !bytes) return;
  int device = 0;
  (void)hipGetDevice(&device);
  std::lock_guard<std::mutex> g(g_canary_mu);
  const unsigned long long lo = (unsigned long long)(uintptr_t)base, hi = lo + bytes;
  size_t k = 0;
  for (size_t i = 0; i < g_canary.size(); i++) {
    CanaryEntry e = g_canary[i];
    if (e.device == device && e.z.addr < hi && e.z.addr + e.z.len > lo) {
      const unsigned long long end = e.z.addr + e.z.len;
      if (end <= hi || e.z.addr < lo) continue;  // inside the range (or around its start: the carve-out in front is part of the fill)
      e.z.len = (uint32_t)(end - hi);              // straddles the end: keep what lies behind it
      e.z.addr = hi;
      e.armed = false;
    }
    g_canary[k++] = e;
  }
  g_canary.resize(k);
}

// one workgroup per zone. fill: writes the pattern; check: the first damaged zone (smallest table index) and the offset of its first
// damaged byte go to result[0..1]
__global__ __launch_bounds__(BLOCK) void k_canary(const CanaryZone* __restrict__ zones, uint32_t n, int fill, uint32_t* __restrict__ result) {
  const CanaryZone z = zones[blockIdx.x];
  uint8_t* p = (uint8_t*)(uintptr_t)z.addr;
  for (uint32_t i = threadIdx.x; i < z.len; i += BLOCK) {
    if (fill) p[i] = CANARY_BYTE;
    else if (p[i] != CANARY_BYTE) {
      uint32_t prev = atomicMin(&result[0], blockIdx.x);
      if (prev >= blockIdx.x) atomicMin(&result[2 + (blockIdx.x & 1023u)], i);
    }
  }
}

static bool canary_run(std::vector<CanaryZone>& zones, int fill, uint32_t* out_first, uint32_t* out_off) {
  if (zones.empty()) return true;
  CanaryZone* dz = nullptr;
  uint32_t* dres = nullptr;
  std::vector<uint32_t> res(2 + 1024, 0xffffffffu);
  bool ok = hipMalloc((void**)&dz, sizeof(CanaryZone) * zones.size()) == hipSuccess && hipMalloc((void**)&dres, 4 * res.size()) == hipSuccess &&
            hipMemcpy(dz, zones.data(), sizeof(CanaryZone) * zones.size(), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(dres, res.data(), 4 * res.size(), hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_canary, dim3((uint32_t)zones.size()), dim3(BLOCK), 0, (hipStream_t)0, (const CanaryZone*)dz, (uint32_t)zones.size(), fill, dres);
    ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(res.data(), dres, 4 * res.size(), hipMemcpyDeviceToHost) == hipSuccess;
  }
  if (dz) (void)hipFree(dz);
  if (dres) (void)hipFree(dres);
  if (!ok) { *out_first = 0xfffffffeu; return false; }
  *out_first = res[0];
  *out_off = res[0] != 0xffffffffu ? res[2 + (res[0] & 1023u)] : 0;
  return res[0] == 0xffffffffu;
}

void canary_arm() {
  if (!canary_on()) return;
  int device = 0;
  (void)hipGetDevice(&device);
  std::vector<CanaryZone> zones;
  {
    std::lock_guard<std::mutex> g(g_canary_mu);
    for (CanaryEntry& e : g_canary)
      if (e.device == device && !e.armed) { zones.push_back(e.z); e.armed = true; }
  }
  if (zones.empty()) return;
  (void)hipDeviceSynchronize();  // (kernels of an earlier layout may still be writing where the new zones lie)
  uint32_t a = 0, b = 0;
  (void)canary_run(zones, 1, &a, &b);
}

bool canary_check(char* msg, size_t msg_len) {
  if (!canary_on()) return true;
  int device = 0;
  (void)hipGetDevice(&device);
  std::vector<CanaryZone> zones;
  std::vector<size_t> which;
  {
    std::lock_guard<std::mutex> g(g_canary_mu);
    for (size_t i = 0; i < g_canary.size(); i++)
      if (g_canary[i].device == device && g_canary[i].armed) { zones.push_back(g_canary[i].z); which.push_back(i); }
  }
  (void)hipDeviceSynchronize();
  uint32_t first = 0, off = 0;
  if (canary_run(zones, 0, &first, &off)) return true;
  std::lock_guard<std::mutex> g(g_canary_mu);
  if (first < which.size() && which[first] < g_canary.size()) {
    CanaryEntry& e = g_canary[which[first]];
    snprintf(msg, msg_len, "AM355_CANARY: carve-out #%u of block '%s' was overrun (red zone of %u bytes damaged from byte +%u on)", e.index, e.block.c_str(), e.z.len, off);
    e.armed = false;  // (filled again by the next canary_arm: one report per overrun)
  } else snprintf(msg, msg_len, "AM355_CANARY: the check itself failed (%s)", hipGetErrorString(hipGetLastError()));
  return false;
}

// ---------------------------------------------------------------------------------------------------------
// fills: several word ranges in one launch (am355_prims.h FillRanges)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_fill_ranges(FillRanges f) {
  const uint32_t tid = gtid(), stride = gridDim.x * BLOCK;
  for (uint32_t k = 0; k < f.n; k++) {
    uint32_t* p = f.p[k];
    const uint32_t n = f.n_words[k], v = f.value[k];
    if (n <= 64) {  // (small ranges may overlap: one thread, in order)
      if (tid == 0) for (uint32_t i = 0; i < n; i++) p[i] = v;
      continue;
    }
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u) / 4u;  // words up to the first 16-byte boundary
    const uint32_t n4 = (n - head) / 4;
    uint4* q = (uint4*)(p + head);
    uint4 vv;
    vv.x = vv.y = vv.z = vv.w = v;
    for (uint32_t i = tid; i < n4; i += stride) q[i] = vv;
    if (tid < head) p[tid] = v;
    const uint32_t tail = head + 4 * n4;
    if (tid < n - tail) p[tail + tid] = v;
  }
}

__global__ __launch_bounds__(BLOCK) void k_copy_ranges(CopyRanges r) {
  const uint32_t tid = gtid(), stride = gridDim.x * BLOCK;
  for (uint32_t k = 0; k < r.n; k++) {
    uint8_t* d = (uint8_t*)r.dst[k];
    const uint8_t* s = (const uint8_t*)r.src[k];
    const uint32_t n = r.bytes[k];
    if ((((uintptr_t)d ^ (uintptr_t)s) & 15u) == 0) {
      // the same misalignment on both sides: bytes up to the first 16-byte boundary, 16-byte words, the last bytes
      uint32_t head = (uint32_t)((16u - ((uintptr_t)d & 15u)) & 15u);
      if (head > n) head = n;
      const uint32_t n16 = (n - head) / 16;
      const uint4* s4 = (const uint4*)(s + head);
      uint4* d4 = (uint4*)(d + head);
      for (uint32_t i = tid; i < n16; i += stride) d4[i] = s4[i];
      if (tid < head) d[tid] = s[tid];
      const uint32_t tail = head + 16 * n16;
      if (tid < n - tail) d[tail + tid] = s[tail + tid];
    } else {
      for (uint32_t i = tid; i < n; i += stride) d[i] = s[i];
    }
  }
}

void launch_copy_ranges(const CopyRanges& r, hipStream_t st) {
  if (!r.n) return;
  size_t most = 0;
  for (uint32_t k = 0; k < r.n; k++) most = std::max<size_t>(most, r.bytes[k]);
  const uint32_t grid = (uint32_t)std::min<size_t>((most / 16 + BLOCK - 1) / BLOCK + 1, 256);
  hipLaunchKernelGGL(k_copy_ranges, dim3(grid), dim3(BLOCK), 0, st, r);
}

__global__ __launch_bounds__(WAVE) void k_signal_words(const uint32_t* __restrict__ src_a, uint32_t n_a, const uint32_t* __restrict__ src_b, uint32_t n_b,
                                                      uint32_t* host_words, volatile uint32_t* host_seq, uint32_t seq) {
  if (gtid() != 0) return;
  for (uint32_t k = 0; k < n_a; k++) host_words[k] = src_a[k];
  for (uint32_t k = 0; k < n_b; k++) host_words[n_a + k] = src_b[k];
  __threadfence_system();
  *host_seq = seq;
}

void launch_signal_words(const uint32_t* src_a, uint32_t n_a, const uint32_t* src_b, uint32_t n_b, uint32_t* host_words, volatile uint32_t* host_seq, uint32_t seq,
                         hipStream_t st) {
  hipLaunchKernelGGL(k_signal_words, dim3(1), dim3(WAVE), 0, st, src_a, n_a, src_b, n_b, host_words, host_seq, seq);
}

void launch_fill_ranges(const FillRanges& f, hipStream_t st) {
  if (!f.n) return;
  size_t words = 0;
  for (uint32_t k = 0; k < f.n; k++) words += f.n_words[k];
  uint32_t grid = (uint32_t)std::min<size_t>((words / 4 + BLOCK - 1) / BLOCK + 1, 2048);
  hipLaunchKernelGGL(k_fill_ranges, dim3(grid), dim3(BLOCK), 0, st, f);
}

}  // namespace am355
