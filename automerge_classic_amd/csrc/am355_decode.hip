// Stage 1 of the replay engine: binary change container parse + SHA-256 + column decode into fixed-width rows.
//
// Reference behaviour reproduced (automerge-classic, paths relative to the reference tree):
//   container / header   backend/columnar.js:688-708 decodeContainerHeader, :635-652 decodeChangeHeader,
//                        :609-624 decodeColumnInfo, :741-765 decodeChangeColumns
//   change hash          SHA-256 (FIPS 180-4) over [chunk type | uLEB length | chunk data]; columnar.js:693-705
//   LEB128               backend/encoding.js:389-488
//   RLE / Delta / Bool   backend/encoding.js:789-920, 1004-1051, 1141-1207
//   row assembly         backend/new.js:570-610 readOperation, :678-724 readNextChangeOp
//
// Work decomposition (bulk replay = many small changes): one lane per change for parse+hash, and one lane per
// (change, column group) for decode, with the column group uniform across a wavefront so that lanes of a wave
// execute the same decoder. All arithmetic is integer/byte work; the kernels stream the encoded bytes once
// and write each fixed-width field once.
#include "am355_decode.h"

#include <cstdlib>

namespace am355 {

// ---------------------------------------------------------------------------------------------------------
// byte cursor + LEB128
// ---------------------------------------------------------------------------------------------------------
// Byte cursor with an 8-byte register window: column streams are consumed a byte at a time by the LEB128 readers,
// and one 8-byte load (any alignment) per 8 bytes instead of one load per byte cuts the dependent memory round
// trips that dominate these latency-bound decoders.
struct __attribute__((packed)) U8B {
  uint64_t v;
};
struct Cur {
  const uint8_t* p;
  uint32_t off, len;
  uint64_t win;
  uint32_t win_off;  // window holds bytes [win_off, win_off + 8); WIN_EMPTY makes every offset miss
  __device__ __forceinline__ Cur() {}
  static constexpr uint32_t WIN_EMPTY = 0xffffff00u;  // o - WIN_EMPTY = o + 256 >= 8 for every valid offset
  __device__ __forceinline__ Cur(const uint8_t* p_, uint32_t off_, uint32_t len_) : p(p_), off(off_), len(len_), win(0), win_off(WIN_EMPTY) {}
  __device__ __forceinline__ uint32_t byte_at(uint32_t o) {
    uint32_t d = o - win_off;
    if (d >= 8) {
      // refill; near the end of the buffer fall back to byte loads so nothing beyond `len` is touched
      if (o + 8 <= len) win = ((const U8B*)(p + o))->v;
      else {
        win = 0;
        for (uint32_t k = 0; o + k < len && k < 8; k++) win |= (uint64_t)p[o + k] << (8 * k);
      }
      win_off = o;
      d = 0;
    }
    return (uint32_t)(win >> (8 * d)) & 0xff;
  }
};

constexpr uint64_t MAX_SAFE = 9007199254740991ull;  // 2^53 - 1

// encoding.js:389-396 + 410-436: at most 10 bytes / 64 bits, result must fit in 53 bits
__device__ __forceinline__ bool read_uleb(Cur& c, uint64_t& out) {
  uint64_t v = 0;
  int shift = 0;
  while (c.off < c.len) {
    uint32_t b = c.byte_at(c.off);
    if (shift == 63 && (b & 0xfe)) return false;
    v |= (uint64_t)(b & 0x7f) << shift;
    shift += 7;
    c.off++;
    if (!(b & 0x80)) {
      out = v;
      return v <= MAX_SAFE;
    }
  }
  return false;
}

// encoding.js:398-408 + 438-488
__device__ __forceinline__ bool read_sleb(Cur& c, int64_t& out) {
  uint64_t v = 0;
  int shift = 0;
  while (c.off < c.len) {
    uint32_t b = c.byte_at(c.off);
    if (shift == 63 && b != 0 && b != 0x7f) return false;
    v |= (uint64_t)(b & 0x7f) << shift;
    shift += 7;
    c.off++;
    if (!(b & 0x80)) {
      if ((b & 0x40) && shift < 64) v |= ~0ull << shift;
      int64_t s = (int64_t)v;
      out = s;
      return s <= (int64_t)MAX_SAFE && s >= -(int64_t)MAX_SAFE;
    }
  }
  return false;
}

__device__ __forceinline__ bool skip_bytes(Cur& c, uint64_t n) {
  if (n > (uint64_t)(c.len - c.off)) return false;
  c.off += (uint32_t)n;
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// SHA-256, one lane per message
// ---------------------------------------------------------------------------------------------------------
__device__ const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

__device__ __forceinline__ void sha_rounds(uint32_t h[8], uint32_t w[16]) {
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t wi;
    if (i < 16) {
      wi = w[i];
    } else {
      uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    uint32_t t1 = hh + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + wi;
    uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// 16-byte loads at arbitrary alignment (changes are packed back to back in the arena; gfx950 global loads are
// alignment-agnostic, so this compiles to global_load_dwordx4)
struct __attribute__((packed)) U4 {
  uint32_t x, y, z, w;
};

__device__ void sha256_bytes(const uint8_t* p, uint32_t len, uint8_t out[32]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  uint32_t w[16];
  uint32_t i = 0;
  for (; i + 64 <= len; i += 64) {
    for (int k = 0; k < 4; k++) {
      U4 v = *(const U4*)(p + i + 16 * k);
      w[4 * k] = __builtin_bswap32(v.x);
      w[4 * k + 1] = __builtin_bswap32(v.y);
      w[4 * k + 2] = __builtin_bswap32(v.z);
      w[4 * k + 3] = __builtin_bswap32(v.w);
    }
    sha_rounds(h, w);
  }
  // final one or two padded blocks
  uint32_t rem = len - i;
  uint64_t bits = (uint64_t)len * 8;
  int n_final = rem < 56 ? 1 : 2;
  for (int blk = 0; blk < n_final; blk++) {
    for (int k = 0; k < 16; k++) {
      uint32_t word = 0;
      for (int j = 0; j < 4; j++) {
        uint32_t pos = blk * 64 + k * 4 + j;  // position within the tail
        uint32_t byte;
        if (pos < rem) byte = p[i + pos];
        else if (pos == rem) byte = 0x80;
        else byte = 0;
        word = word << 8 | byte;
      }
      w[k] = word;
    }
    if (blk == n_final - 1) {
      w[14] = (uint32_t)(bits >> 32);
      w[15] = (uint32_t)bits;
    }
    sha_rounds(h, w);
  }
  for (int k = 0; k < 8; k++) {
    out[4 * k] = h[k] >> 24; out[4 * k + 1] = h[k] >> 16; out[4 * k + 2] = h[k] >> 8; out[4 * k + 3] = h[k];
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_parse_changes: one lane per change.  Container header, checksum, change header, column directory, row and
// pred counts (run-level scan of the action / predNum columns).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int col_slot(uint64_t id) {
  switch (id) {
    case 0x01: return C_OBJ_ACTOR;
    case 0x02: return C_OBJ_CTR;
    case 0x11: return C_KEY_ACTOR;
    case 0x13: return C_KEY_CTR;
    case 0x15: return C_KEY_STR;
    case 0x34: return C_INSERT;
    case 0x42: return C_ACTION;
    case 0x56: return C_VAL_LEN;
    case 0x57: return C_VAL_RAW;
    case 0x70: return C_PRED_NUM;
    case 0x71: return C_PRED_ACTOR;
    case 0x73: return C_PRED_CTR;
    default: return -1;  // unknown columns are carried through by the reference and never affect patches
  }
}

// number of values and their sum in an RLE-uint column (run level; validity of individual values is checked
// again by the decode kernel)
__device__ bool rle_count_sum(const uint8_t* p, uint32_t len, uint64_t& count, uint64_t& sum) {
  Cur c(p, 0, len);
  count = 0;
  sum = 0;
  while (c.off < c.len) {
    int64_t n;
    if (!read_sleb(c, n)) return false;
    if (n > 1) {
      uint64_t v;
      if (!read_uleb(c, v)) return false;
      count += (uint64_t)n;
      sum += (uint64_t)n * v;
    } else if (n < 0) {
      for (int64_t k = 0; k < -n; k++) {
        uint64_t v;
        if (!read_uleb(c, v)) return false;
        sum += v;
      }
      count += (uint64_t)(-n);
    } else if (n == 0) {
      uint64_t z;
      if (!read_uleb(c, z)) return false;
      count += z;
    } else {
      return false;  // repetition count 1
    }
    if (count > 0xfffffff0ull || sum > 0xfffffff0ull) return false;
  }
  return true;
}

__global__ __launch_bounds__(WAVE) void k_parse_changes(const uint8_t* __restrict__ arena, const uint64_t* __restrict__ offsets,
                                                         uint32_t n_changes, ChangeMeta* __restrict__ metas, uint32_t* __restrict__ n_entries) {
  uint32_t c = gtid();
  if (c >= n_changes) return;
  ChangeMeta m;
  m.base = offsets[c];
  uint64_t len64 = offsets[c + 1] - offsets[c];
  m.len = (uint32_t)len64;
  m.flags = 0;
  m.seq = m.start_op = 0;
  m.n_deps = m.deps_off = m.actor_off = m.actor_len = m.n_other = m.others_off = m.n_ops = m.n_preds = 0;
  m.n_entries = 0;
  m.author_slot = m.max_first = NONE32;
  m.pad = 0;
  for (int k = 0; k < C_NUM; k++) m.col_off[k] = m.col_len[k] = 0;
  const uint8_t* p = arena + m.base;
  do {
    if (len64 > 0xfffffff0ull) { m.flags |= F_OVERFLOW; break; }
    if (m.len < 10) { m.flags |= F_BAD_CHUNK; break; }
    if (p[0] != 0x85 || p[1] != 0x6f || p[2] != 0x4a || p[3] != 0x83) { m.flags |= F_BAD_MAGIC; break; }
    Cur cur(p, 9, m.len);
    uint64_t chunk_len;
    if (!read_uleb(cur, chunk_len)) { m.flags |= F_BAD_LEB; break; }
    // the raw arena holds uncompressed (type 1) chunks only; exactly one container per change, no trailing bytes
    if (p[8] != 1 || chunk_len != (uint64_t)(m.len - cur.off)) { m.flags |= F_BAD_CHUNK; break; }
    // change header
    uint64_t v;
    int64_t sv;
    bool ok = read_uleb(cur, v);
    m.n_deps = (uint32_t)v;
    m.deps_off = cur.off;
    ok = ok && skip_bytes(cur, v * 32);
    ok = ok && read_uleb(cur, v);
    m.actor_off = cur.off;
    m.actor_len = (uint32_t)v;
    ok = ok && skip_bytes(cur, v);
    ok = ok && read_uleb(cur, m.seq) && read_uleb(cur, m.start_op) && read_sleb(cur, sv);
    ok = ok && read_uleb(cur, v) && skip_bytes(cur, v);  // message
    ok = ok && read_uleb(cur, v);
    m.n_other = (uint32_t)v;
    m.others_off = cur.off;
    if (ok && v > m.len) ok = false;
    for (uint64_t k = 0; ok && k < m.n_other; k++) {
      uint64_t l;
      ok = read_uleb(cur, l) && skip_bytes(cur, l);
    }
    if (!ok) { m.flags |= F_BAD_LEB; break; }
    if (m.actor_len >= 65536) { m.flags |= F_UNSUPPORTED; break; }
    m.n_entries = m.n_other + 1;
    // column directory: ids strictly ascending ignoring the deflate bit (bit 3), which a change must not use
    uint64_t ncols;
    if (!read_uleb(cur, ncols) || ncols > m.len) { m.flags |= F_BAD_LEB; break; }
    uint32_t dir_off = cur.off;
    int64_t last = -1;
    uint64_t total = 0;
    for (uint64_t k = 0; k < ncols; k++) {
      uint64_t id, l;
      if (!read_uleb(cur, id) || !read_uleb(cur, l)) { m.flags |= F_BAD_LEB; break; }
      if ((int64_t)(id & ~8ull) <= last) { m.flags |= F_BAD_COLUMNS; break; }
      last = (int64_t)(id & ~8ull);
      if (id & 8) { m.flags |= F_BAD_COLUMNS; break; }
      total += l;
    }
    if (m.flags) break;
    if (total > (uint64_t)(m.len - cur.off)) { m.flags |= F_BAD_CHUNK; break; }
    uint32_t data_off = cur.off;
    Cur dir(p, dir_off, m.len);
    for (uint64_t k = 0; k < ncols; k++) {
      uint64_t id, l;
      read_uleb(dir, id);
      read_uleb(dir, l);
      int s = col_slot(id);
      if (s >= 0) { m.col_off[s] = data_off; m.col_len[s] = (uint32_t)l; }
      data_off += (uint32_t)l;
    }
    // whatever follows the columns is `extraBytes` (columnar.js:757-760): preserved by the reference, unused here
    uint64_t cnt, sum;
    if (!rle_count_sum(p + m.col_off[C_ACTION], m.col_len[C_ACTION], cnt, sum)) { m.flags |= F_BAD_RLE; break; }
    m.n_ops = (uint32_t)cnt;
    if (!rle_count_sum(p + m.col_off[C_PRED_NUM], m.col_len[C_PRED_NUM], cnt, sum)) { m.flags |= F_BAD_RLE; break; }
    // only the first n_ops rows of predNum count (a longer column is ignored; a shorter one is padded with nulls)
    m.n_preds = (uint32_t)sum;
    if (m.start_op + m.n_ops > 0xfffffff0ull) m.flags |= F_OVERFLOW;
  } while (0);
  metas[c] = m;
  n_entries[c] = m.flags ? 0 : m.n_entries;
}

// ---------------------------------------------------------------------------------------------------------
// k_hash_changes: SHA-256 of every change (one lane per change) + checksum verification. Runs on its own stream:
// nothing on the decode/merge critical path needs the hashes (they feed dependency resolution and the heads).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_hash_changes(const uint8_t* __restrict__ arena, const uint64_t* __restrict__ offsets, uint32_t n_changes,
                                                        uint8_t* __restrict__ hashes, uint32_t* __restrict__ min_idx, uint32_t* __restrict__ flags) {
  uint32_t c = gtid();
  if (c >= n_changes) return;
  const uint8_t* p = arena + offsets[c];
  uint64_t len = offsets[c + 1] - offsets[c];
  min_idx[c] = c;
  uint8_t h[32];
  for (int k = 0; k < 32; k++) h[k] = 0;
  if (len >= 10 && len < 0xfffffff0ull) {
    sha256_bytes(p + 8, (uint32_t)len - 8, h);
    if (h[0] != p[4] || h[1] != p[5] || h[2] != p[6] || h[3] != p[7]) atomicOr(flags, (uint32_t)F_BAD_CHECKSUM);  // columnar.js:702-704
  }
  for (int k = 0; k < 32; k++) hashes[32 * (size_t)c + k] = h[k];
}

__device__ __forceinline__ bool equal32(const uint8_t* a, const uint8_t* b) {
  U4 a0 = *(const U4*)a, a1 = *(const U4*)(a + 16), b0 = *(const U4*)b, b1 = *(const U4*)(b + 16);
  return a0.x == b0.x && a0.y == b0.y && a0.z == b0.z && a0.w == b0.w && a1.x == b1.x && a1.y == b1.y && a1.z == b1.z && a1.w == b1.w;
}
__device__ __forceinline__ uint32_t hash_slot(const uint8_t* h, uint32_t mask) {
  uint64_t v = 0;
  for (int k = 0; k < 8; k++) v |= (uint64_t)h[k] << (8 * k);
  return (uint32_t)((v * 0x9e3779b97f4a7c15ull) >> 32) & mask;
}

// device hash table of change hashes: tab[slot] = (index of the first change that claimed the slot) + 1;
// min_idx[claimer] = smallest input index among the changes with that same hash (duplicates)
__global__ __launch_bounds__(BLOCK) void k_hash_insert(const uint8_t* __restrict__ hashes, uint32_t n, uint32_t* __restrict__ tab, uint32_t mask,
                                                       uint32_t* __restrict__ min_idx) {
  uint32_t c = gtid();
  if (c >= n) return;
  const uint8_t* h = hashes + 32 * (size_t)c;
  uint32_t i = hash_slot(h, mask);
  for (uint32_t probes = 0; probes <= mask; probes++) {
    uint32_t old = atomicCAS(&tab[i], 0u, c + 1);
    if (old == 0) return;
    if (equal32(hashes + 32 * (size_t)(old - 1), h)) { atomicMin(&min_idx[old - 1], c); return; }
    i = (i + 1) & mask;
  }
}

__device__ __forceinline__ uint32_t hash_find(const uint8_t* __restrict__ hashes, const uint32_t* __restrict__ tab, uint32_t mask,
                                              const uint32_t* __restrict__ min_idx, const uint8_t* h) {
  uint32_t i = hash_slot(h, mask);
  for (uint32_t probes = 0; probes <= mask; probes++) {
    uint32_t v = tab[i];
    if (v == 0) return NONE32;
    if (equal32(hashes + 32 * (size_t)(v - 1), h)) return min_idx[v - 1];
    i = (i + 1) & mask;
  }
  return NONE32;
}

// every dependency must be an earlier change of the batch for the in-order fast path (new.js:1562-1567).
// One wavefront per change, one lane per dependency hash.
__global__ __launch_bounds__(WAVE) void k_deps_resolve(const uint8_t* __restrict__ arena, const ChangeMeta* __restrict__ metas,
                                                       const uint8_t* __restrict__ hashes, uint32_t n, const uint32_t* __restrict__ tab, uint32_t mask,
                                                       const uint32_t* __restrict__ min_idx, uint8_t* __restrict__ has_dependent,
                                                       uint32_t* __restrict__ fast_flags) {
  uint32_t c = blockIdx.x, lane = threadIdx.x;
  if (c >= n) return;
  const ChangeMeta* m = &metas[c];
  if (m->flags) return;
  uint32_t ff = 0;
  if (lane == 0 && hash_find(hashes, tab, mask, min_idx, hashes + 32 * (size_t)c) != c) ff |= FF_DUP_HASH;
  const uint8_t* deps = arena + m->base + m->deps_off;
  for (uint32_t k = lane; k < m->n_deps; k += WAVE) {
    uint32_t d = hash_find(hashes, tab, mask, min_idx, deps + 32 * (size_t)k);
    if (d == NONE32) ff |= FF_MISSING_DEP;
    else {
      if (d >= c) ff |= FF_LATE_DEP;
      has_dependent[d] = 1;
    }
  }
  if (ff) atomicOr(fast_flags, ff);
}

// ---------------------------------------------------------------------------------------------------------
// actor ids: device-side interning.  slots[i] = ((arena offset of the id bytes + 1) << 16) | length, 0 = empty.
// The slot index is the provisional actor number; the host ranks the (few) distinct ids lexicographically.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t actor_find_or_insert(const uint8_t* __restrict__ arena, unsigned long long* __restrict__ slots, uint32_t mask,
                                                         uint32_t off, uint32_t len) {
  const uint8_t* p = arena + off;
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint32_t k = 0; k < len; k++) h = (h ^ p[k]) * 0x100000001b3ull;
  uint32_t i = (uint32_t)(h >> 20) & mask;
  unsigned long long mine = ((unsigned long long)off + 1) << 16 | len;
  for (uint32_t probes = 0; probes <= mask; probes++) {
    unsigned long long v = slots[i];
    if (v == 0) {
      v = atomicCAS(&slots[i], 0ull, mine);
      if (v == 0) return i;
    }
    if ((uint32_t)(v & 0xffff) == len) {
      const uint8_t* q = arena + ((v >> 16) - 1);
      bool eq = true;
      for (uint32_t k = 0; k < len && eq; k++) eq = p[k] == q[k];
      if (eq) return i;
    }
    i = (i + 1) & mask;
  }
  return NONE32;
}

// One wavefront per change: lane 0 walks the length-prefixed table (sequential by nature) 64 entries at a time
// into LDS, then every lane interns one entry.
__global__ __launch_bounds__(WAVE) void k_actor_intern(const uint8_t* __restrict__ arena, ChangeMeta* __restrict__ metas, uint32_t n,
                                                        const uint32_t* __restrict__ amap_base, uint32_t* __restrict__ amap, uint32_t amap_cap,
                                                        unsigned long long* __restrict__ slots, uint32_t mask, uint32_t* __restrict__ first_idx,
                                                        uint32_t* __restrict__ flags, uint32_t* __restrict__ fast_flags) {
  __shared__ uint32_t s_off[WAVE], s_len[WAVE];
  uint32_t c = blockIdx.x, lane = threadIdx.x;
  if (c >= n) return;
  ChangeMeta* m = &metas[c];
  if (m->flags) return;
  uint32_t base = amap_base[c];
  if ((uint64_t)base + m->n_entries > amap_cap) {
    if (lane == 0) atomicOr(fast_flags, (uint32_t)FF_CAPACITY);
    return;
  }
  const uint8_t* p = arena + m->base;
  uint32_t abs0 = (uint32_t)m->base;
  uint32_t n_other = m->n_other;
  if (lane == 0) {
    uint32_t s = actor_find_or_insert(arena, slots, mask, abs0 + m->actor_off, m->actor_len);
    if (s == NONE32) { atomicOr(flags, (uint32_t)F_UNSUPPORTED); s = 0; }
    amap[base] = s;
    m->author_slot = s;
    atomicMin(&first_idx[s], c);
  }
  Cur cur(p, m->others_off, m->len);  // advanced by lane 0 only
  for (uint32_t k0 = 0; k0 < n_other; k0 += WAVE) {
    uint32_t cnt = n_other - k0 < WAVE ? n_other - k0 : WAVE;
    if (lane == 0) {
      for (uint32_t j = 0; j < cnt; j++) {
        uint64_t l = 0;
        read_uleb(cur, l);
        s_off[j] = abs0 + cur.off;
        s_len[j] = l < 65536 ? (uint32_t)l : NONE32;
        skip_bytes(cur, l);
      }
    }
    __syncthreads();
    if (lane < cnt) {
      uint32_t t = s_len[lane] != NONE32 ? actor_find_or_insert(arena, slots, mask, s_off[lane], s_len[lane]) : NONE32;
      if (t == NONE32) { atomicOr(flags, (uint32_t)F_UNSUPPORTED); t = 0; }
      amap[base + 1 + k0 + lane] = t;
    }
    __syncthreads();
  }
}

// every actor a change mentions must already be in the document when the change is read (new.js:1442-1449):
// with in-order application that means its first change has an index <= this one
__global__ __launch_bounds__(BLOCK) void k_actor_check(ChangeMeta* __restrict__ metas, uint32_t n, const uint32_t* __restrict__ amap_base,
                                                       const uint32_t* __restrict__ amap, uint32_t amap_cap, const uint32_t* __restrict__ first_idx,
                                                       uint32_t* __restrict__ flags, uint32_t* __restrict__ fast_flags) {
  uint32_t c = gtid();
  if (c >= n) return;
  ChangeMeta* m = &metas[c];
  if (m->flags) return;
  uint32_t base = amap_base[c];
  if ((uint64_t)base + m->n_entries > amap_cap) return;
  uint32_t mx = 0;
  for (uint32_t k = 0; k < m->n_entries; k++) {
    uint32_t f = first_idx[amap[base + k]];
    mx = f > mx ? f : mx;  // NONE32 (no change by that actor in the batch) also lands on the general path
  }
  m->max_first = mx;
  if (mx > c) atomicOr(fast_flags, (uint32_t)FF_LATE_ACTOR);
}

// ---------------------------------------------------------------------------------------------------------
// column decoders (per lane, sequential over one column)
// ---------------------------------------------------------------------------------------------------------
struct Rle {
  Cur c;
  int state;       // 0 none, 1 repetition, 2 literal, 3 nulls
  int64_t count;
  bool have_last, last_null;
  int64_t last;
  uint32_t last_off, last_len;  // utf8
};

__device__ __forceinline__ void rle_init(Rle& r, const uint8_t* p, uint32_t len) {
  r.c = Cur(p, 0, len);
  r.state = 0;
  r.count = 0;
  r.have_last = false;
  r.last_null = true;
  r.last = 0;
  r.last_off = r.last_len = 0;
}

enum { RT_UINT = 0, RT_INT = 1, RT_UTF8 = 2 };

struct RVal {
  bool is_null;
  int64_t i;
  uint32_t off, len;
};

template <int TYPE>
__device__ __forceinline__ bool rle_raw(Rle& r, RVal& v) {
  v.is_null = false;
  if (TYPE == RT_UINT) {
    uint64_t u;
    if (!read_uleb(r.c, u)) return false;
    v.i = (int64_t)u;
  } else if (TYPE == RT_INT) {
    if (!read_sleb(r.c, v.i)) return false;
  } else {
    uint64_t n;
    if (!read_uleb(r.c, n)) return false;
    v.off = r.c.off;
    v.len = (uint32_t)n;
    if (!skip_bytes(r.c, n)) return false;
  }
  return true;
}

template <int TYPE>
__device__ __forceinline__ bool rle_same(const Rle& r, const RVal& v) {
  if (!r.have_last || r.last_null) return false;
  if (TYPE == RT_UTF8) {
    if (r.last_len != v.len) return false;
    for (uint32_t k = 0; k < v.len; k++)
      if (r.c.p[r.last_off + k] != r.c.p[v.off + k]) return false;
    return true;
  }
  return r.last == v.i;
}

template <int TYPE>
__device__ __forceinline__ void rle_set_last(Rle& r, const RVal& v) {
  r.have_last = true;
  r.last_null = v.is_null;
  r.last = v.i;
  r.last_off = v.off;
  r.last_len = v.len;
}

// next value of an RLE column; past the end every value is null (encoding.js:821). false => malformed.
template <int TYPE>
__device__ bool rle_next(Rle& r, RVal& v) {
  if (r.count == 0 && r.c.off >= r.c.len) {
    v.is_null = true;
    v.i = 0;
    v.off = v.len = 0;
    return true;
  }
  if (r.count == 0) {
    int64_t n;
    if (!read_sleb(r.c, n)) return false;
    if (n > 1) {
      RVal x;
      x.off = x.len = 0;
      x.i = 0;
      if (!rle_raw<TYPE>(r, x)) return false;
      if ((r.state == 1 || r.state == 2) && rle_same<TYPE>(r, x)) return false;  // successive equal repetitions
      r.state = 1;
      rle_set_last<TYPE>(r, x);
      r.count = n;
    } else if (n == 1) {
      return false;
    } else if (n < 0) {
      if (r.state == 2) return false;  // successive literals
      r.state = 2;
      r.count = -n;
    } else {
      if (r.state == 3) return false;  // successive null runs
      uint64_t z;
      if (!read_uleb(r.c, z) || z == 0) return false;
      r.count = (int64_t)z;
      r.state = 3;
      RVal x;
      x.is_null = true;
      x.i = 0;
      x.off = x.len = 0;
      rle_set_last<TYPE>(r, x);
    }
  }
  r.count--;
  if (r.state == 2) {
    RVal x;
    x.off = x.len = 0;
    x.i = 0;
    if (!rle_raw<TYPE>(r, x)) return false;
    if (rle_same<TYPE>(r, x)) return false;  // repetition inside a literal
    rle_set_last<TYPE>(r, x);
    v = x;
  } else {
    v.is_null = r.last_null;
    v.i = r.last;
    v.off = r.last_off;
    v.len = r.last_len;
  }
  return true;
}

struct Delta {
  Rle r;
  int64_t abs;
};
__device__ __forceinline__ bool delta_next(Delta& d, RVal& v) {
  if (!rle_next<RT_INT>(d.r, v)) return false;
  if (!v.is_null) {
    d.abs += v.i;
    if (d.abs > (int64_t)MAX_SAFE || d.abs < -(int64_t)MAX_SAFE) return false;
    v.i = d.abs;
  }
  return true;
}

struct BoolDec {
  Cur c;
  bool last, first;
  uint64_t count;
};
__device__ __forceinline__ bool bool_next(BoolDec& b, bool& v) {
  if (b.count == 0 && b.c.off >= b.c.len) { v = false; return true; }
  while (b.count == 0) {
    if (!read_uleb(b.c, b.count)) return false;
    b.last = !b.last;
    if (b.count == 0 && !b.first) return false;
    b.first = false;
  }
  b.count--;
  v = b.last;
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// k_decode_columns: grid.y selects the column group (uniform per wave), one lane per applied change.
// ---------------------------------------------------------------------------------------------------------
enum Task { T_OBJ, T_KEY, T_KEYSTR, T_INSERT, T_ACTION, T_VALUE, T_PREDNUM, T_PREDS, T_NUM };

// change-local actor index -> global actor rank. `amap` holds either final ranks (slot_rank == nullptr: table built by
// the host scheduler) or device actor-table slots that `slot_rank` maps to ranks.
struct ActorXlate {
  const uint32_t* amap;
  const uint32_t* slot_rank;
};
__device__ __forceinline__ uint32_t xlate_actor(const ActorXlate& x, const ChangePlan& pl, int64_t local, uint32_t& err) {
  if (local < 0 || (uint64_t)local >= pl.n_actors) { err |= F_BAD_ROW; return 0; }
  uint32_t v = x.amap[pl.amap_base + (uint32_t)local];
  return x.slot_rank ? x.slot_rank[v] : v;
}

__global__ __launch_bounds__(WAVE) void k_decode_columns(const uint8_t* __restrict__ arena, const ChangeMeta* __restrict__ metas,
                                                          const ChangePlan* __restrict__ plans, uint32_t n_plans,
                                                          ActorXlate amap, OpCols o, uint32_t* __restrict__ flags, int task_base) {
  uint32_t pi = gtid();
  if (pi >= n_plans) return;
  const ChangePlan pl = plans[pi];
  const ChangeMeta* m = &metas[pl.change];
  const uint8_t* p = arena + m->base;
  uint32_t n = m->n_ops, base = pl.op_base;
  uint32_t err = 0;
  uint32_t abs0 = (uint32_t)m->base;  // arena is < 4 GiB (checked on the host), so absolute offsets fit 32 bits
  int task = blockIdx.y + task_base;
  RVal v;
  v.off = v.len = 0;
  v.i = 0;
  v.is_null = true;
  if (task == T_OBJ) {
    Rle a, c;
    rle_init(a, p + m->col_off[C_OBJ_ACTOR], m->col_len[C_OBJ_ACTOR]);
    rle_init(c, p + m->col_off[C_OBJ_CTR], m->col_len[C_OBJ_CTR]);
    for (uint32_t i = 0; i < n; i++) {
      RVal va, vc;
      if (!rle_next<RT_UINT>(a, va) || !rle_next<RT_UINT>(c, vc)) { err |= F_BAD_RLE; break; }
      if (va.is_null != vc.is_null) err |= F_BAD_ROW;  // new.js:715-718
      if (!vc.is_null && (uint64_t)vc.i >= NONE32) err |= F_OVERFLOW;
      o.obj_actor[base + i] = va.is_null ? NONE32 : xlate_actor(amap, pl, va.i, err);
      o.obj_ctr[base + i] = vc.is_null ? 0 : (uint32_t)vc.i;
    }
  } else if (task == T_KEY) {
    Rle a;
    Delta c;
    rle_init(a, p + m->col_off[C_KEY_ACTOR], m->col_len[C_KEY_ACTOR]);
    rle_init(c.r, p + m->col_off[C_KEY_CTR], m->col_len[C_KEY_CTR]);
    c.abs = 0;
    for (uint32_t i = 0; i < n; i++) {
      RVal va, vc;
      if (!rle_next<RT_UINT>(a, va) || !delta_next(c, vc)) { err |= F_BAD_RLE; break; }
      // new.js:719-723
      if ((vc.is_null && !va.is_null) || (!vc.is_null && vc.i == 0 && !va.is_null) || (!vc.is_null && vc.i > 0 && va.is_null)) err |= F_BAD_ROW;
      if (!vc.is_null && (vc.i < 0 || (uint64_t)vc.i >= NONE32)) err |= F_OVERFLOW;
      o.key_actor[base + i] = va.is_null ? NONE32 : xlate_actor(amap, pl, va.i, err);
      o.key_ctr[base + i] = vc.is_null ? NONE32 : (uint32_t)vc.i;
    }
  } else if (task == T_KEYSTR) {
    Rle s;
    rle_init(s, p + m->col_off[C_KEY_STR], m->col_len[C_KEY_STR]);
    uint32_t col_abs = abs0 + m->col_off[C_KEY_STR];
    for (uint32_t i = 0; i < n; i++) {
      if (!rle_next<RT_UTF8>(s, v)) { err |= F_BAD_RLE; break; }
      o.key_off[base + i] = v.is_null ? 0 : col_abs + v.off;
      o.key_len[base + i] = v.is_null ? NONE32 : v.len;
    }
  } else if (task == T_INSERT) {
    BoolDec b;
    b.c = Cur(p + m->col_off[C_INSERT], 0, m->col_len[C_INSERT]);
    b.last = true;
    b.first = true;
    b.count = 0;
    for (uint32_t i = 0; i < n; i++) {
      bool x;
      if (!bool_next(b, x)) { err |= F_BAD_RLE; break; }
      o.insert[base + i] = x ? 1 : 0;
    }
  } else if (task == T_ACTION) {
    Rle a;
    rle_init(a, p + m->col_off[C_ACTION], m->col_len[C_ACTION]);
    for (uint32_t i = 0; i < n; i++) {
      if (!rle_next<RT_UINT>(a, v)) { err |= F_BAD_RLE; break; }
      if (v.is_null) err |= F_UNSUPPORTED;
      if ((uint64_t)v.i >= NONE32) err |= F_OVERFLOW;
      o.action[base + i] = (uint32_t)v.i;
      // ops carry no id columns in a change: op i is (startOp + i, author)  (new.js:708-709)
      o.id_ctr[base + i] = (uint32_t)m->start_op + i;
      o.id_actor[base + i] = pl.author;
    }
  } else if (task == T_VALUE) {
    Rle l;
    rle_init(l, p + m->col_off[C_VAL_LEN], m->col_len[C_VAL_LEN]);
    uint64_t used = 0;
    uint32_t raw_abs = abs0 + m->col_off[C_VAL_RAW];
    for (uint32_t i = 0; i < n; i++) {
      if (!rle_next<RT_UINT>(l, v)) { err |= F_BAD_RLE; break; }
      uint64_t tl = v.is_null ? 0 : (uint64_t)v.i;
      if (tl >= NONE32) { err |= F_OVERFLOW; tl = 0; }
      if (used + (tl >> 4) > m->col_len[C_VAL_RAW]) { err |= F_BAD_CHUNK; break; }  // readRawBytes past the column
      o.val_tl[base + i] = (uint32_t)tl;
      o.val_off[base + i] = raw_abs + (uint32_t)used;
      used += tl >> 4;
    }
  } else if (task == T_PREDNUM) {
    Rle a;
    rle_init(a, p + m->col_off[C_PRED_NUM], m->col_len[C_PRED_NUM]);
    uint64_t run = 0;
    for (uint32_t i = 0; i < n; i++) {
      if (!rle_next<RT_UINT>(a, v)) { err |= F_BAD_RLE; break; }
      uint64_t k = v.is_null ? 0 : (uint64_t)v.i;
      if (run + k > m->n_preds) { err |= F_BAD_RLE; break; }
      o.pred_num[base + i] = (uint32_t)k;
      o.pred_first[base + i] = pl.pred_base + (uint32_t)run;
      run += k;
    }
  } else if (task == T_PREDS) {
    // the pred group is consumed for the first n_ops rows only; its total was measured by k_parse_changes over
    // the whole predNum column, so re-measure here to stay exact when that column is longer than the action column
    Rle num, a;
    Delta c;
    rle_init(num, p + m->col_off[C_PRED_NUM], m->col_len[C_PRED_NUM]);
    rle_init(a, p + m->col_off[C_PRED_ACTOR], m->col_len[C_PRED_ACTOR]);
    rle_init(c.r, p + m->col_off[C_PRED_CTR], m->col_len[C_PRED_CTR]);
    c.abs = 0;
    uint32_t w = pl.pred_base;
    uint64_t written = 0;
    for (uint32_t i = 0; i < n && !err; i++) {
      if (!rle_next<RT_UINT>(num, v)) { err |= F_BAD_RLE; break; }
      uint64_t k = v.is_null ? 0 : (uint64_t)v.i;
      if (written + k > m->n_preds) { err |= F_BAD_RLE; break; }
      for (uint64_t j = 0; j < k; j++) {
        RVal va, vc;
        if (!rle_next<RT_UINT>(a, va) || !delta_next(c, vc)) { err |= F_BAD_RLE; break; }
        if (va.is_null || vc.is_null) { err |= F_UNSUPPORTED; break; }
        if (vc.i < 0 || (uint64_t)vc.i >= NONE32) err |= F_OVERFLOW;
        o.pred_actor[w] = xlate_actor(amap, pl, va.i, err);
        o.pred_ctr[w] = (uint32_t)vc.i;
        w++;
        written++;
      }
    }
  }
  if (err) atomicOr(flags, err);
}

void launch_parse_changes(const uint8_t* arena, const uint64_t* offsets, uint32_t n_changes, ChangeMeta* metas, uint32_t* n_entries, hipStream_t st) {
  if (!n_changes) return;
  AM355_LAUNCH_INDEPENDENT(k_parse_changes, dim3((n_changes + WAVE - 1) / WAVE), dim3(WAVE), st, arena, offsets, n_changes, metas, n_entries);
}

void launch_hash_changes(const uint8_t* arena, const uint64_t* offsets, uint32_t n, uint8_t* hashes, uint32_t* min_idx, uint32_t* hash_tab,
                         uint32_t tab_mask, uint32_t* flags, hipStream_t st) {
  if (!n) return;
  AM355_LAUNCH_INDEPENDENT(k_hash_changes, dim3((n + WAVE - 1) / WAVE), dim3(WAVE), st, arena, offsets, n, hashes, min_idx, flags);
  AM355_LAUNCH_INDEPENDENT(k_hash_insert, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), st, (const uint8_t*)hashes, n, hash_tab, tab_mask, min_idx);
}

void launch_deps_resolve(const uint8_t* arena, const ChangeMeta* metas, const uint8_t* hashes, uint32_t n, const uint32_t* hash_tab, uint32_t tab_mask,
                         const uint32_t* min_idx, uint8_t* has_dependent, uint32_t* fast_flags, hipStream_t st) {
  if (!n) return;
  AM355_LAUNCH_INDEPENDENT(k_deps_resolve, dim3(n), dim3(WAVE), st, arena, metas, hashes, n, hash_tab, tab_mask, min_idx, has_dependent, fast_flags);
}

void launch_actor_intern(const uint8_t* arena, ChangeMeta* metas, uint32_t n, const uint32_t* amap_base, uint32_t* amap, uint32_t amap_cap,
                         unsigned long long* slots, uint32_t slot_mask, uint32_t* first_idx, uint32_t* flags, uint32_t* fast_flags, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_actor_intern, dim3(n), dim3(WAVE), 0, st, arena, metas, n, amap_base, amap, amap_cap, slots, slot_mask, first_idx, flags,
                     fast_flags);
  AM355_LAUNCH_INDEPENDENT(k_actor_check, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), st, metas, n, amap_base, (const uint32_t*)amap, amap_cap,
                           (const uint32_t*)first_idx, flags, fast_flags);
}

void launch_decode_columns(const uint8_t* arena, const ChangeMeta* metas, const ChangePlan* plans, uint32_t n_plans, const uint32_t* amap,
                           const uint32_t* slot_rank, OpCols cols, uint32_t* flags, hipStream_t st) {
  if (!n_plans) return;
  ActorXlate x{amap, slot_rank};
  static const bool split = getenv("AM355_SPLIT_DECODE") != nullptr;  // diagnostic: one launch per column group so a profiler can time them
  if (split) {
    for (int t = 0; t < T_NUM; t++)
      AM355_LAUNCH_INDEPENDENT(k_decode_columns, dim3((n_plans + WAVE - 1) / WAVE, 1), dim3(WAVE), st, arena, metas, plans, n_plans, x, cols, flags, t);
    return;
  }
  AM355_LAUNCH_INDEPENDENT(k_decode_columns, dim3((n_plans + WAVE - 1) / WAVE, T_NUM), dim3(WAVE), st, arena, metas, plans, n_plans, x, cols, flags, 0);
}

}  // namespace am355
