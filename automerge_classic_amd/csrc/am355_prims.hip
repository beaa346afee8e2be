// Device-wide primitives for the replay engine: exclusive scan and stable LSD radix sort of (u64 key, u32 value)
// pairs. Hand-written for gfx950: 256-thread workgroups (one wave per SIMD), 64-lane ballot-based digit
// ranking, LDS histograms. These are HBM-bound integer kernels (no MFMA): each sort pass reads and writes
// every pair once (12 B in, 12 B out) plus one histogram read of the keys.
#include "am355_prims.h"

namespace am355 {

// ---------------------------------------------------------------------------------------------------------
// exclusive scan (uint32 -> uint32), n up to 2^32-1.  Three launches: per-tile sums, scan of tile sums, apply.
// ---------------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *total gets the block sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s /* [BLOCK] LDS */, uint32_t* total) {
  uint32_t t = threadIdx.x;
  s[t] = v;
  __syncthreads();
  for (uint32_t off = 1; off < BLOCK; off <<= 1) {
    uint32_t add = t >= off ? s[t - off] : 0;
    __syncthreads();
    s[t] += add;
    __syncthreads();
  }
  uint32_t incl = s[t];
  *total = s[BLOCK - 1];
  __syncthreads();
  return incl - v;
}

__global__ __launch_bounds__(BLOCK) void k_scan_tile_sums(const uint32_t* __restrict__ in, uint32_t* __restrict__ tile_sums, uint32_t n) {
  __shared__ uint32_t s[BLOCK];
  uint32_t base = blockIdx.x * SCAN_TILE;
  uint32_t sum = 0;
  for (int j = 0; j < SCAN_ITEMS; j++) {
    uint32_t i = base + j * BLOCK + threadIdx.x;
    if (i < n) sum += in[i];
  }
  uint32_t total;
  block_exclusive_scan(sum, s, &total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of the tile sums in place; writes grand total
__global__ __launch_bounds__(BLOCK) void k_scan_sums(uint32_t* __restrict__ sums, uint32_t n_tiles, uint32_t* __restrict__ grand_total) {
  __shared__ uint32_t s[BLOCK];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n_tiles; base += BLOCK) {
    uint32_t i = base + threadIdx.x;
    uint32_t v = i < n_tiles ? sums[i] : 0;
    uint32_t total;
    uint32_t ex = block_exclusive_scan(v, s, &total);
    if (i < n_tiles) sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0 && grand_total) *grand_total = carry;
}

__global__ __launch_bounds__(BLOCK) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                      const uint32_t* __restrict__ tile_sums, uint32_t n) {
  __shared__ uint32_t s[BLOCK];
  // thread t owns SCAN_ITEMS consecutive elements so the in-thread prefix is sequential
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t sum = 0;
  for (int j = 0; j < SCAN_ITEMS; j++) {
    uint32_t i = base + j;
    v[j] = i < n ? in[i] : 0;
    sum += v[j];
  }
  uint32_t total;
  uint32_t ex = block_exclusive_scan(sum, s, &total) + tile_sums[blockIdx.x];
  for (int j = 0; j < SCAN_ITEMS; j++) {
    uint32_t i = base + j;
    if (i < n) out[i] = ex;
    ex += v[j];
  }
}

size_t scan_workspace_bytes(uint32_t n) { return sizeof(uint32_t) * ((size_t)(n + SCAN_TILE - 1) / SCAN_TILE + 1); }

void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* d_total, void* ws, hipStream_t st) {
  uint32_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t* sums = (uint32_t*)ws;
  if (n_tiles == 0) {
    if (d_total) (void)hipMemsetAsync(d_total, 0, sizeof(uint32_t), st);
    return;
  }
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(BLOCK), 0, st, in, sums, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLOCK), 0, st, sums, n_tiles, d_total);
  hipLaunchKernelGGL(k_scan_apply, dim3(n_tiles), dim3(BLOCK), 0, st, in, out, sums, n);
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

__global__ __launch_bounds__(BLOCK) void k_sort_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                         uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                         int shift, const uint32_t* __restrict__ table, uint32_t n_tiles) {
  __shared__ uint32_t wave_cnt[BLOCK / WAVE][RADIX];
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
    uint32_t run = table[t * n_tiles + blockIdx.x];
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
int radix_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, uint32_t n, int begin_bit, int end_bit,
                     void* ws, hipStream_t st) {
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
    hipLaunchKernelGGL(k_sort_hist, dim3(n_tiles), dim3(BLOCK), 0, st, ki, n, shift, table, n_tiles);
    exclusive_scan_u32(table, table, (uint32_t)table_n, nullptr, scan_ws, st);
    hipLaunchKernelGGL(k_sort_scatter, dim3(n_tiles), dim3(BLOCK), 0, st, ki, vi, ko, vo, n, shift, table, n_tiles);
    cur ^= 1;
  }
  return cur;
}

}  // namespace am355
