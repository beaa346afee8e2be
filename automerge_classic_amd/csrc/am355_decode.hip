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
#include "am355_scan.h"

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
__host__ __device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) { return ((const U8B*)p)->v; }
#ifdef AM355_LDS_IS_DISTINCT
__device__ __forceinline__ uint64_t load_u64_unaligned(LdsBytes p) { return ((const __attribute__((address_space(3))) U8B*)p)->v; }
#endif

// P = `const uint8_t*` (global memory) or LdsBytes
template <class P>
struct CurT {
  P p;
  uint32_t off, len;
  uint64_t win;
  uint32_t win_off;  // window holds bytes [win_off, win_off + 8); WIN_EMPTY makes every offset miss
  __host__ __device__ __forceinline__ CurT() {}
  static constexpr uint32_t WIN_EMPTY = 0xffffff00u;  // o - WIN_EMPTY = o + 256 >= 8 for every valid offset
  __host__ __device__ __forceinline__ CurT(P p_, uint32_t off_, uint32_t len_) : p(p_), off(off_), len(len_), win(0), win_off(WIN_EMPTY) {}
  __host__ __device__ __forceinline__ uint32_t byte_at(uint32_t o) {
    uint32_t d = o - win_off;
    if (d >= 8) {
      // refill; near the end of the buffer fall back to byte loads so nothing beyond `len` is touched
      if (o + 8 <= len) win = load_u64_unaligned(p + o);
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
using Cur = CurT<const uint8_t*>;
using CurLds = CurT<LdsBytes>;

// 16-byte loads at arbitrary alignment (changes are packed back to back in the arena; gfx950 global loads are
// alignment-agnostic, so this compiles to global_load_dwordx4)
struct __attribute__((packed)) U4 {
  uint32_t x, y, z, w;
};
struct alignas(16) V4 {  // the same 16 bytes at a 16-byte aligned LDS address
  uint32_t x, y, z, w;
};
__device__ __forceinline__ void stage_to_lds(uint8_t* dst /* 16-byte aligned */, const uint8_t* __restrict__ src, uint32_t total, uint32_t lane) {
  uint32_t vec = total & ~15u;
  for (uint32_t i = lane * 16; i < vec; i += WAVE * 16) {
    U4 v = *(const U4*)(src + i);
    V4 w{v.x, v.y, v.z, v.w};
    *(V4*)(dst + i) = w;
  }
  for (uint32_t i = vec + lane; i < total; i += WAVE) dst[i] = src[i];
}

// byte-string equality, eight bytes per load (`limit` = end of the buffer both ranges live in: no read beyond it)
template <class P>
__device__ __forceinline__ bool bytes_equal(P p, uint32_t a, uint32_t b, uint32_t len, uint32_t limit) {
  uint32_t k = 0;
  for (; k + 8 <= len && a + k + 8 <= limit && b + k + 8 <= limit; k += 8)
    if (load_u64_unaligned(p + a + k) != load_u64_unaligned(p + b + k)) return false;
  for (; k < len; k++)
    if (p[a + k] != p[b + k]) return false;
  return true;
}

constexpr uint64_t MAX_SAFE = 9007199254740991ull;  // 2^53 - 1

// encoding.js:389-396 + 410-436: at most 10 bytes / 64 bits, result must fit in 53 bits
template <class C>
__host__ __device__ __forceinline__ bool read_uleb(C& c, uint64_t& out) {
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
template <class C>
__host__ __device__ __forceinline__ bool read_sleb(C& c, int64_t& out) {
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

template <class C>
__host__ __device__ __forceinline__ bool skip_bytes(C& c, uint64_t n) {
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
// three-input xor: ONE v_bitop3_b32 (truth table 0x96) on gfx950 -- the compiler leaves two v_xor in the Sigma functions otherwise
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

// One 64-byte block. Fully unrolled so that the 16-word message window lives in fixed registers (a rolled loop makes the
// compiler index the register file dynamically, several times slower) and the a..h rotation is pure renaming.
#define AM355_SHA_ROUND(a, b, c, d, e, f, g, hh, i)                                                            \
  {                                                                                                            \
    uint32_t wi;                                                                                               \
    if ((i) < 16) {                                                                                            \
      wi = w[(i)&15];                                                                                          \
    } else {                                                                                                   \
      uint32_t w15 = w[((i)-15) & 15], w2 = w[((i)-2) & 15];                                                   \
      uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);                                             \
      uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);                                              \
      wi = w[(i)&15] + s0 + w[((i)-7) & 15] + s1;                                                              \
      w[(i)&15] = wi;                                                                                          \
    }                                                                                                          \
    uint32_t t1 = hh + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + wi;  \
    uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));              \
    d += t1;                                                                                                   \
    hh = t1 + t2;                                                                                              \
  }

__device__ __forceinline__ void sha_rounds(uint32_t h[8], uint32_t w[16]) {
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
  for (int i = 0; i < 64; i += 8) {
    AM355_SHA_ROUND(a, b, c, d, e, f, g, hh, i + 0)
    AM355_SHA_ROUND(hh, a, b, c, d, e, f, g, i + 1)
    AM355_SHA_ROUND(g, hh, a, b, c, d, e, f, i + 2)
    AM355_SHA_ROUND(f, g, hh, a, b, c, d, e, i + 3)
    AM355_SHA_ROUND(e, f, g, hh, a, b, c, d, i + 4)
    AM355_SHA_ROUND(d, e, f, g, hh, a, b, c, i + 5)
    AM355_SHA_ROUND(c, d, e, f, g, hh, a, b, i + 6)
    AM355_SHA_ROUND(b, c, d, e, f, g, hh, a, i + 7)
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

__device__ void sha256_bytes(const uint8_t* p, uint32_t len, uint8_t out[32]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  uint32_t w[16];
  uint32_t i = 0;
  // software prefetch: the next 64-byte block is requested before the 64 dependent rounds of the current one, so the
  // memory round trip hides under ~2000 serial ALU instructions instead of preceding them
  U4 cur[4], nxt[4];
  if (len >= 64)
    for (int k = 0; k < 4; k++) cur[k] = *(const U4*)(p + 16 * k);
  for (; i + 64 <= len; i += 64) {
    bool more = i + 128 <= len;
    if (more)
      for (int k = 0; k < 4; k++) nxt[k] = *(const U4*)(p + i + 64 + 16 * k);
    for (int k = 0; k < 4; k++) {
      w[4 * k] = __builtin_bswap32(cur[k].x);
      w[4 * k + 1] = __builtin_bswap32(cur[k].y);
      w[4 * k + 2] = __builtin_bswap32(cur[k].z);
      w[4 * k + 3] = __builtin_bswap32(cur[k].w);
    }
    sha_rounds(h, w);
    if (more)
      for (int k = 0; k < 4; k++) cur[k] = nxt[k];
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
__host__ __device__ __forceinline__ int col_slot(uint64_t id) {
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
// (Numbers of one byte -- nearly all of them -- are taken straight from the column: the general LEB128 reader with its byte window costs
// ~100 instructions per byte, and this walk is one lane's: for a change of 313 ops with pred lists it was 80 us of k_parse_changes,
// measured with the kernel's clock. Literal runs of one-byte values are summed eight bytes at a time.)
template <class P>
__host__ __device__ __forceinline__ bool rle_count_sum(P p, uint32_t len, uint64_t& count, uint64_t& sum) {
  uint32_t off = 0;
  count = 0;
  sum = 0;
  auto general_u = [&](uint64_t& v) {
    CurT<P> c(p, off, len);
    bool ok = read_uleb(c, v);
    off = c.off;
    return ok;
  };
  auto general_s = [&](int64_t& v) {
    CurT<P> c(p, off, len);
    bool ok = read_sleb(c, v);
    off = c.off;
    return ok;
  };
  while (off < len) {
    int64_t n;
    uint32_t b = p[off];
    if (b < 0x80) { n = (b & 0x40) ? (int64_t)b - 128 : (int64_t)b; off++; }
    else if (!general_s(n)) return false;
    if (n > 1) {
      uint64_t v;
      b = off < len ? (uint32_t)p[off] : 0x80u;
      if (b < 0x80) { v = b; off++; }
      else if (!general_u(v)) return false;
      count += (uint64_t)n;
      sum += (uint64_t)n * v;
    } else if (n < 0) {
      int64_t left = -n;
      // eight one-byte values at a time (the high bits of all eight clear)
      while (left >= 8 && off + 8 <= len) {
        const uint64_t w = load_u64_unaligned(p + off);
        if (w & 0x8080808080808080ull) break;
        uint64_t t = (w & 0x00ff00ff00ff00ffull) + (w >> 8 & 0x00ff00ff00ff00ffull);
        t = (t & 0x0000ffff0000ffffull) + (t >> 16 & 0x0000ffff0000ffffull);
        sum += (t & 0xffffffffull) + (t >> 32);
        off += 8;
        left -= 8;
      }
      for (; left > 0; left--) {
        uint64_t v;
        b = off < len ? (uint32_t)p[off] : 0x80u;
        if (b < 0x80) { v = b; off++; }
        else if (!general_u(v)) return false;
        sum += v;
      }
      count += (uint64_t)(-n);
    } else if (n == 0) {
      uint64_t z;
      b = off < len ? (uint32_t)p[off] : 0x80u;
      if (b < 0x80) { z = b; off++; }
      else if (!general_u(z)) return false;
      count += z;
    } else {
      return false;  // repetition count 1
    }
    if (count > 0xfffffff0ull || sum > 0xfffffff0ull) return false;
  }
  return true;
}

// One wavefront per change: the change is staged into LDS with wide coalesced loads (one memory round trip instead of
// one per dependent header field), then lane 0 parses it there. Changes larger than the staging buffer are parsed
// straight from global memory by the same code.
constexpr uint32_t PARSE_STAGE = 8192;

// header + column directory of one change, read through `p` (LDS when the change was staged, else global memory)
template <class P>
__host__ __device__ __forceinline__ void parse_change(P p, uint64_t base64, uint64_t len64, ChangeMeta* out, uint32_t* n_entries_out) {
  bool stored = false;
  uint32_t unknown_cols = 0;
  ChangeMeta m;
  m.base = base64;
  m.len = (uint32_t)len64;
  m.flags = 0;
  m.seq = m.start_op = 0;
  m.n_deps = m.deps_off = m.actor_off = m.actor_len = m.n_other = m.others_off = m.n_ops = m.n_preds = 0;
  m.n_entries = 0;
  m.author_slot = m.max_first = NONE32;
  m.pad = 0;
  for (int k = 0; k < C_NUM; k++) m.col_off[k] = m.col_len[k] = 0;
  do {
    if (len64 > 0xfffffff0ull) { m.flags |= F_OVERFLOW; break; }
    if (m.len < 10) { m.flags |= F_BAD_CHUNK; break; }
    if (p[0] != 0x85 || p[1] != 0x6f || p[2] != 0x4a || p[3] != 0x83) { m.flags |= F_BAD_MAGIC; break; }
    CurT<P> cur(p, 9, m.len);
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
      // actor ids are 16 bytes in practice: a one-byte length is the fast path of this (serial) walk
      uint32_t b0 = cur.off < cur.len ? cur.byte_at(cur.off) : 0x80u;
      if (b0 < 0x80) { cur.off++; ok = skip_bytes(cur, b0); continue; }
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
    CurT<P> dir(p, dir_off, m.len);
    // the directory entries go straight to the output record: indexing a local copy by the column slot would push the
    // whole struct into scratch memory
    *out = m;
    stored = true;
    uint32_t act_off = 0, act_len = 0, pn_off = 0, pn_len = 0;
    for (uint64_t k = 0; k < ncols; k++) {
      uint64_t id, l;
      read_uleb(dir, id);
      read_uleb(dir, l);
      int s = col_slot(id);
      if (s >= 0) { out->col_off[s] = data_off; out->col_len[s] = (uint32_t)l; }
      else if (l) unknown_cols = 1;  // preserved by the reference (new.js:1387-1425); irrelevant to the patch, but save() must keep them
      if (s == C_ACTION) { act_off = data_off; act_len = (uint32_t)l; }
      if (s == C_PRED_NUM) { pn_off = data_off; pn_len = (uint32_t)l; }
      data_off += (uint32_t)l;
    }
    // whatever follows the columns is `extraBytes` (columnar.js:757-760): preserved by the reference, unused here
    uint64_t cnt, sum;
    if (!rle_count_sum(p + act_off, act_len, cnt, sum)) { m.flags |= F_BAD_RLE; break; }
    m.n_ops = (uint32_t)cnt;
    if (!rle_count_sum(p + pn_off, pn_len, cnt, sum)) { m.flags |= F_BAD_RLE; break; }
    // only the first n_ops rows of predNum count (a longer column is ignored; a shorter one is padded with nulls)
    m.n_preds = (uint32_t)sum;
    if (m.start_op + m.n_ops > 0xfffffff0ull) m.flags |= F_OVERFLOW;
  } while (0);
  if (!stored) *out = m;
  else { out->flags = m.flags; out->n_ops = m.n_ops; out->n_preds = m.n_preds; out->pad = unknown_cols; }
  *n_entries_out = m.flags ? 0 : m.n_entries;
}

// ---- the header of a staged change by the whole wavefront --------------------------------------------------------------------
// The lane-serial parse above is ~150 dependent LDS reads behind ~4500 instructions of one lane, and a wavefront costs its SIMD the
// same four cycles per instruction whatever the number of active lanes: 4 k changes x 4500 instructions were the 50 us of this
// kernel. Here the 64 lanes look at a 64-byte window at once: the terminator bytes (< 0x80) of the LEB128 numbers come from one
// ballot, the lane that holds the terminator of token r decodes it (encoding.js:389-408, same range rules) and leaves value, end and
// validity in LDS for everybody. The header is five such windows (chunk length + dependency count | actor length | seq, startOp,
// time, message length | number of other actors | column directory), the other-actor table is checked for the uniform 16-byte ids
// in one step, and the two run-level column walks (row and pred counts) run on two lanes side by side.
// ANY irregularity -- a malformed or oversized number, actor ids of another length, a directory that does not fit one window, a
// rule violation -- makes this return false and lane 0 parse the change with parse_change(), which alone decides the flags.
struct TokScratch {
  uint64_t v[WAVE];
  uint8_t end[WAVE];  // bytes of the window consumed up to and including the token
  uint8_t ok[WAVE];
  uint32_t col_off[C_NUM], col_len[C_NUM];
  uint32_t unknown;
};

template <class P>
__device__ __forceinline__ uint32_t wave_tokenize(P p, uint32_t off, uint32_t len, uint32_t lane, uint64_t signed_mask, TokScratch& S) {
  __syncthreads();  // (one wavefront per workgroup: orders the previous window's reads before this one's writes)
  uint32_t pos = off + lane;
  uint32_t b = pos < len ? (uint32_t)p[pos] : 0x80u;
  uint64_t term = __ballot(b < 0x80);
  if (b < 0x80) {
    uint64_t below = term & ((1ull << lane) - 1);
    uint32_t r = (uint32_t)__popcll(below);
    uint32_t start = below ? 64u - (uint32_t)__clzll(below) : 0u;
    bool sg = (signed_mask >> r) & 1, ok = true;
    uint64_t v = 0;
    int shift = 0;
    uint32_t bb = 0;
    for (uint32_t k = start; k <= lane; k++) {
      bb = p[off + k];
      if (shift == 63 && (sg ? (bb != 0 && bb != 0x7f) : (bb & 0xfe) != 0)) { ok = false; break; }
      v |= (uint64_t)(bb & 0x7f) << shift;
      shift += 7;
    }
    if (ok) {
      if (sg) {
        if ((bb & 0x40) && shift < 64) v |= ~0ull << shift;
        int64_t s = (int64_t)v;
        ok = s <= (int64_t)MAX_SAFE && s >= -(int64_t)MAX_SAFE;
      } else ok = v <= MAX_SAFE;
    }
    S.v[r] = v;
    S.end[r] = (uint8_t)(lane + 1);
    S.ok[r] = ok;
  }
  __syncthreads();
  return (uint32_t)__popcll(term);
}

// Rows of the action column / sum of the predNum column of a FAT change by the whole wavefront (k_parse_changes<true>). Two lanes
// walking the two columns record by record (rle_count_sum) are two or three dependent LDS reads per record: 94 us for the map
// workload's changes -- three hundred short records in predNum -- and the parse kernel is as long as its slowest wavefront. Here: the
// column's numbers by ballots (as the decoder does), every number read as a record header names the next header, the true headers
// are the orbit of the first number (pointer doubling), and a header's rows / sum come from prefix sums over the numbers. Accepts and
// rejects exactly what rle_count_sum does; anything unusual -- a number of more than four bytes, a column that does not end on a
// number, more than FAT_TOKENS numbers -- is left to rle_count_sum itself (`handled` false).
constexpr uint32_t FAT_TOKENS = 1024;
struct FatScratch {
  uint32_t tok[FAT_TOKENS];    // 7-bit groups assembled (<= 28 bits) | byte count << 28
  uint32_t pre[FAT_TOKENS + 1];  // exclusive prefix sums of the numbers read as unsigned values
  uint16_t jump[FAT_TOKENS];
  uint8_t mark[FAT_TOKENS];
};
__device__ __forceinline__ int32_t fat_signed(uint32_t w) {
  const uint32_t nb = w >> 28, raw = w & 0x0fffffffu, bits = 7 * nb;
  return (raw >> (bits - 1)) & 1 ? (int32_t)(raw | (~0u << bits)) : (int32_t)raw;
}
__device__ __forceinline__ uint32_t wave_sum_u32x(uint32_t v) {
  for (int d = WAVE / 2; d; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ bool wave_count_sum(LdsBytes col, uint32_t len, uint32_t lane, FatScratch& F, uint64_t& count, uint64_t& sum, bool& handled) {
  handled = false;
  count = sum = 0;
  if (len == 0) { handled = true; return true; }
  if (len > FAT_TOKENS) return true;
  // numbers
  uint32_t T = 0, carry_start = 0;
  bool plain = true;
  __syncthreads();
  for (uint32_t chunk = 0; chunk < len; chunk += WAVE) {
    const uint32_t pos = chunk + lane;
    const bool in = pos < len;
    const uint32_t b = in ? (uint32_t)col[pos] : 0x80u;
    const bool term = in && !(b & 0x80);
    const unsigned long long mask = __ballot(term);
    if (term) {
      const unsigned long long below = mask & ((1ull << lane) - 1);
      const uint32_t start = below ? chunk + (63 - (uint32_t)__clzll(below)) + 1 : carry_start;
      const uint32_t nb = pos - start + 1, idx = T + (uint32_t)__popcll(below);
      if (nb > 4) plain = false;
      else {
        uint32_t v = 0;
        for (uint32_t k = 0; k < nb; k++) v |= ((uint32_t)col[start + k] & 0x7f) << (7 * k);
        F.tok[idx] = v | nb << 28;
      }
    }
    if (mask) carry_start = chunk + (63 - (uint32_t)__clzll(mask)) + 1;
    T += (uint32_t)__popcll(mask);
  }
  if (__ballot(!plain) || carry_start != len) return true;  // (rle_count_sum decides)
  __syncthreads();
  // prefix sums of the numbers as unsigned values; what every number says as a header
  uint64_t run = 0;   // (64 numbers of 28 bits: a chunk's sum stays below 2^34; the running total is kept in 64 bits)
  for (uint32_t chunk = 0; chunk < T; chunk += WAVE) {
    const uint32_t t = chunk + lane;
    const uint64_t v = t < T ? (uint64_t)(F.tok[t] & 0x0fffffffu) : 0ull;
    uint64_t incl = v;
    for (int d = 1; d < WAVE; d <<= 1) { const uint64_t o = (uint64_t)__shfl_up((long long)incl, d); if ((int)lane >= d) incl += o; }
    if (t < T) F.pre[t] = (uint32_t)(run + incl - v);
    run += (uint64_t)__shfl((long long)incl, WAVE - 1);
  }
  if (run > 0xffffffffull) return true;  // (32-bit prefix sums would wrap: rle_count_sum decides -- it rejects sums beyond 2^32 - 16)
  if (lane == 0) F.pre[T] = (uint32_t)run;
  for (uint32_t t = lane; t < T; t += WAVE) {
    const int32_t cnt = fat_signed(F.tok[t]);
    uint32_t next = T;
    if (cnt > 1) { if (t + 1 < T) next = t + 2; }
    else if (cnt < 0) { if ((uint32_t)(-cnt) <= T - t - 1) next = t + 1 + (uint32_t)(-cnt); }
    else if (cnt == 0) { if (t + 1 < T) next = t + 2; }
    F.jump[t] = (uint16_t)(next > T ? T : next);
    F.mark[t] = t == 0 ? 1 : 0;
  }
  __syncthreads();
  for (uint32_t hop = 1; hop < T; hop <<= 1) {
    uint16_t nj[FAT_TOKENS / WAVE];
#pragma unroll
    for (uint32_t k = 0; k < FAT_TOKENS / WAVE; k++) {
      const uint32_t t = lane + k * WAVE;
      nj[k] = (uint16_t)T;
      if (t < T) {
        const uint32_t j = F.jump[t];
        if (F.mark[t] && j < T) F.mark[j] = 1;
        if (j < T) nj[k] = F.jump[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < FAT_TOKENS / WAVE; k++) {
      const uint32_t t = lane + k * WAVE;
      if (t < T) F.jump[t] = nj[k];
    }
    __syncthreads();
    if (F.jump[0] >= T) {
      for (uint32_t t = lane; t < T; t += WAVE) { const uint32_t j = F.jump[t]; if (F.mark[t] && j < T) F.mark[j] = 1; }
      __syncthreads();
      break;
    }
  }
  // rows and sum of the marked headers; a header rle_count_sum would stop at makes the column malformed
  uint64_t rows = 0, total = 0;
  bool bad = false;
  for (uint32_t t = lane; t < T; t += WAVE) {
    if (!F.mark[t]) continue;
    const int32_t cnt = fat_signed(F.tok[t]);
    if (cnt > 1) {
      if (t + 1 >= T) bad = true;
      else { rows += (uint64_t)cnt; total += (uint64_t)cnt * (F.tok[t + 1] & 0x0fffffffu); }
    } else if (cnt == 1) bad = true;
    else if (cnt < 0) {
      const uint32_t k = (uint32_t)(-cnt);
      if (k > T - t - 1) bad = true;
      else { rows += k; total += F.pre[t + 1 + k] - F.pre[t + 1]; }
    } else {
      if (t + 1 >= T) bad = true;
      else rows += F.tok[t + 1] & 0x0fffffffu;
    }
  }
  handled = true;
  if (__ballot(bad)) return false;
  for (int d = WAVE / 2; d; d >>= 1) {
    rows += (uint64_t)__shfl_xor((long long)rows, d);
    total += (uint64_t)__shfl_xor((long long)total, d);
  }
  // (rle_count_sum gives up as soon as a running total passes 2^32 - 16; both totals only grow: the final ones decide the same)
  if (rows > 0xfffffff0ull || total > 0xfffffff0ull) return false;
  count = rows;
  sum = total;
  return true;
}

template <class P>
__device__ __forceinline__ bool parse_change_wave(P p, uint64_t base64, uint64_t len64, ChangeMeta* out, uint32_t* n_entries_out, uint32_t lane,
                                                  TokScratch& S, FatScratch* F = nullptr) {
  if (len64 > 0xfffffff0ull || len64 < 10) return false;
  const uint32_t len = (uint32_t)len64;
  if (p[0] != 0x85 || p[1] != 0x6f || p[2] != 0x4a || p[3] != 0x83 || p[8] != 1) return false;
  // window 1: chunk length, number of dependencies
  uint32_t off = 9;
  uint32_t n = wave_tokenize(p, off, len, lane, 0, S);
  if (n < 2 || !S.ok[0] || !S.ok[1]) return false;
  if (S.v[0] != (uint64_t)(len - (off + S.end[0]))) return false;
  uint64_t n_deps = S.v[1];
  off += S.end[1];
  const uint32_t deps_off = off;
  if (n_deps > (uint64_t)(len - off) / 32) return false;
  off += (uint32_t)n_deps * 32;
  // window 2: length of the author's id
  n = wave_tokenize(p, off, len, lane, 0, S);
  if (n < 1 || !S.ok[0]) return false;
  uint64_t actor_len = S.v[0];
  off += S.end[0];
  const uint32_t actor_off = off;
  if (actor_len > (uint64_t)(len - off) || actor_len >= 65536) return false;
  off += (uint32_t)actor_len;
  // window 3: seq, startOp, time (signed), message length
  n = wave_tokenize(p, off, len, lane, 1ull << 2, S);
  if (n < 4 || !S.ok[0] || !S.ok[1] || !S.ok[2] || !S.ok[3]) return false;
  const uint64_t seq = S.v[0], start_op = S.v[1], msg_len = S.v[3];
  off += S.end[3];
  if (msg_len > (uint64_t)(len - off)) return false;
  off += (uint32_t)msg_len;
  // window 4: number of other actors, then their table
  n = wave_tokenize(p, off, len, lane, 0, S);
  if (n < 1 || !S.ok[0]) return false;
  uint64_t n_other = S.v[0];
  off += S.end[0];
  const uint32_t others_off = off;
  if (n_other > (uint64_t)(len - off) / 17) return false;  // (ids of other lengths: serial walk)
  bool uniform = true;
  for (uint32_t k = lane; k < (uint32_t)n_other; k += WAVE) uniform = uniform && p[off + 17 * k] == 16;
  if (__ballot(!uniform)) return false;
  off += 17 * (uint32_t)n_other;
  // window 5: number of columns and the (id, length) directory
  n = wave_tokenize(p, off, len, lane, 0, S);
  if (n < 1 || !S.ok[0]) return false;
  const uint64_t ncols = S.v[0];
  if (ncols > 31 || n < 1 + 2 * (uint32_t)ncols) return false;
  const uint32_t nc = (uint32_t)ncols;
  const uint32_t dir_end = off + S.end[2 * nc];
  bool good = true;
  uint64_t id = 0, l = 0;
  if (lane < nc) {
    id = S.v[1 + 2 * lane];
    l = S.v[2 + 2 * lane];
    good = S.ok[1 + 2 * lane] && S.ok[2 + 2 * lane] && !(id & 8) && l <= (uint64_t)len;
    if (lane > 0 && S.v[2 * lane - 1] >= id) good = false;  // ids strictly ascending (no deflate bit on either: plain comparison)
  }
  if (__ballot(!good)) return false;
  uint32_t incl = wave_incl_scan_u32((uint32_t)l, lane);  // (l <= len < 2^32 each, at most 31 of them: the sum is checked in 64 bits below)
  uint64_t total = 0;
  for (uint32_t k = 0; k < nc; k++) total += S.v[2 + 2 * k];
  if (total > (uint64_t)(len - dir_end)) return false;
  __syncthreads();
  if (lane < C_NUM) S.col_off[lane] = S.col_len[lane] = 0;
  if (lane == 0) S.unknown = 0;
  __syncthreads();
  if (lane < nc) {
    int s = col_slot(id);
    if (s >= 0) { S.col_off[s] = dir_end + incl - (uint32_t)l; S.col_len[s] = (uint32_t)l; }
    else if (l) S.unknown = 1;
  }
  __syncthreads();
  // rows = values of the action column, preds = sum of the predNum column: two lanes, one column each
  uint64_t cnt = 0, sum = 0;
  bool rle_ok = true;
  bool by_wave[2] = {false, false};
  uint64_t wave_cnt = 0, wave_sum = 0;
  if (F) {
    // fat changes (k_parse_changes<true>): a column of some length is counted by the whole wavefront
    for (int which = 0; which < 2; which++) {
      const int s = which == 0 ? C_ACTION : C_PRED_NUM;
      if (S.col_len[s] < 96) continue;
      uint64_t c0, s0;
      bool handled;
      const bool ok = wave_count_sum((LdsBytes)(p + S.col_off[s]), S.col_len[s], lane, *F, c0, s0, handled);
      if (!handled) continue;
      if (!ok) return false;
      by_wave[which] = true;
      if (which == 0) wave_cnt = c0; else wave_sum = s0;
    }
  }
  if (lane < 2 && !by_wave[lane]) {
    int s = lane == 0 ? C_ACTION : C_PRED_NUM;
    rle_ok = rle_count_sum(p + S.col_off[s], S.col_len[s], cnt, sum);
  }
  if (__ballot(!rle_ok)) return false;
  const uint64_t n_ops = by_wave[0] ? wave_cnt : __shfl(cnt, 0), n_preds = by_wave[1] ? wave_sum : __shfl(sum, 1);
  if (start_op + n_ops > 0xfffffff0ull) return false;
  if (lane < C_NUM) { out->col_off[lane] = S.col_off[lane]; out->col_len[lane] = S.col_len[lane]; }
  if (lane == 0) {
    out->base = base64;
    out->len = len;
    out->flags = 0;
    out->seq = seq;
    out->start_op = start_op;
    out->n_entries = (uint32_t)n_other + 1;
    out->author_slot = out->max_first = NONE32;
    out->pad = S.unknown;
    out->n_deps = (uint32_t)n_deps;
    out->deps_off = deps_off;
    out->actor_off = actor_off;
    out->actor_len = (uint32_t)actor_len;
    out->n_other = (uint32_t)n_other;
    out->others_off = others_off;
    out->n_ops = (uint32_t)n_ops;
    out->n_preds = (uint32_t)n_preds;
    *n_entries_out = (uint32_t)n_other + 1;
  }
  return true;
}

// (Four changes per wavefront -- four lanes parsing four staged changes in lockstep -- was measured and dropped in round 2: changes of
// different shape make the lanes diverge, which cost the map workload (few, fat changes with different keys) 0.12 ms per replay.)
// `fills`: word ranges the kernels behind this one expect cleared (flag words, the actor hash table, the merge stage's counter block).
// They depend on nothing of the replay; as fills on a stream of their own they cost the main stream an event wait in front of the
// next kernel -- here every workgroup clears a slice on its way in (a few hundred bytes each).
// FAT (chosen by the host from the batch's bytes per change): 11 KB more of LDS per wavefront for wave_count_sum. A batch of thousands of
// small changes keeps the lean form: every one of its wavefronts is resident at once with 9 KB each, and its columns are a few records.
template <bool FAT>
__global__ __launch_bounds__(WAVE) void k_parse_changes(const uint8_t* __restrict__ arena, const uint64_t* __restrict__ offsets,
                                                         uint32_t n_changes, ChangeMeta* __restrict__ metas, uint32_t* __restrict__ n_entries, ParseFills fills) {
  __shared__ alignas(16) uint8_t stage[PARSE_STAGE];
  __shared__ TokScratch scratch;
  FatScratch* fat = nullptr;
  if constexpr (FAT) {
    __shared__ FatScratch fat_scratch;
    fat = &fat_scratch;
  }
  wave_priority_high();
  uint32_t c = blockIdx.x, lane = threadIdx.x;
  for (uint32_t r = 0; r < fills.n; r++) {
    uint32_t* __restrict__ q = fills.p[r];
    const uint32_t v = fills.value[r];
    for (uint64_t w = (uint64_t)c * WAVE + lane; w < fills.n_words[r]; w += (uint64_t)gridDim.x * WAVE) q[w] = v;
  }
  if (c >= n_changes) return;
  uint64_t base64 = offsets[c], total64 = offsets[c + 1] - offsets[c];
  bool staged = total64 <= PARSE_STAGE;
  if (staged) stage_to_lds(stage, arena + base64, (uint32_t)total64, lane);
  __syncthreads();
  if (staged && parse_change_wave((LdsBytes)stage, base64, total64, &metas[c], &n_entries[c], lane, scratch, fat)) return;
  if (lane != 0) return;
  // two instantiations so that the staged case reads LDS with ds_read instead of FLAT loads through a generic pointer
  if (staged) parse_change((LdsBytes)stage, base64, total64, &metas[c], &n_entries[c]);
  else parse_change(arena + base64, base64, total64, &metas[c], &n_entries[c]);
}

// The same records for a handful of changes whose bytes the host has at hand (am355_replay.hip replay_resident: the batch of a
// Backend.applyChanges onto a kept state): parse_change() above, compiled for the host -- a kernel launch, a copy back and the wait for
// both cost ~25 us, the walk of a change's header a microsecond. The device never sees a different parser: this IS parse_change.
void parse_changes_host(const uint8_t* arena, const uint64_t* offsets, uint32_t n_changes, ChangeMeta* metas, uint32_t* n_entries) {
  for (uint32_t i = 0; i < n_changes; i++) {
    const uint64_t base = offsets[i], len = offsets[i + 1] - offsets[i];
    parse_change<const uint8_t*>(arena + base, base, len, &metas[i], &n_entries[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_hash_changes: SHA-256 of every change + checksum verification (columnar.js:693-705). Runs on its own stream: nothing on
// the decode/merge critical path needs the hashes (they feed dependency resolution and the heads).
//
// The blocks of one message are a dependent chain, and a lone wavefront issues one instruction every four cycles whatever
// it is: the time of this kernel is (instructions per block on the chain) x (blocks of the longest change) x 4 cycles. Rounds
// 1-3 ran everything in the one lane that owns the change: ~2000 instructions per block, 51 blocks, 236 us. But the message
// SCHEDULE (W[16..63], 13 of the ~30 instructions of a round, plus the loads and byte swaps) does not depend on the chaining
// value. A workgroup is now TWO wavefronts over the same 64 changes, on two SIMDs of one CU:
//   producer  lane c loads block k+1 of change c (prefetched one step ahead), pads it if it is one of the last two, expands the
//             schedule and leaves W[t] + K[t] for the 64 rounds in LDS (one 16-byte store per four rounds, [round/4][lane][4]:
//             consecutive lanes on consecutive banks);
//   consumer  lane c runs the 64 rounds of block k from LDS: Sigma1, Ch (one bit-select), Sigma0, Maj (bit-select + xor), adds.
// One barrier per block; two LDS buffers of 16 KB. The chain per block is the consumer's ~1000 instructions.
// ---------------------------------------------------------------------------------------------------------
constexpr int HASH_LANES = 64;  // changes per workgroup

#define AM355_SHA_SCHED(i)                                                                                 \
  {                                                                                                        \
    uint32_t w15 = w[((i)-15) & 15], w2 = w[((i)-2) & 15];                                                 \
    uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);                                          \
    uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);                                           \
    w[(i)&15] = w[(i)&15] + s0 + w[((i)-7) & 15] + s1;                                                     \
  }
// (Ch = bit-select of f / g by e; Maj = bit-select of c / b by a ^ b: one v_bfi_b32 each)
#define AM355_SHA_ROUND_WK(a, b, c, d, e, f, g, hh, wk)                                                    \
  {                                                                                                        \
    uint32_t t1 = hh + xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)) + (((f ^ g) & e) ^ g) + (wk);      \
    uint32_t t2 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)) + (((a ^ b) & (c ^ b)) ^ b);          \
    d += t1;                                                                                               \
    hh = t1 + t2;                                                                                          \
  }

__global__ __launch_bounds__(2 * HASH_LANES) void k_hash_changes(const uint8_t* __restrict__ arena, const uint64_t* __restrict__ offsets, uint32_t n_changes,
                                                                  uint8_t* __restrict__ hashes, uint32_t* __restrict__ min_idx, uint32_t* __restrict__ flags) {
  __shared__ V4 wk[2][16 * HASH_LANES];
  __shared__ uint32_t s_steps;
  const uint32_t lane = threadIdx.x & (HASH_LANES - 1);
  const bool producer = threadIdx.x >= HASH_LANES;
  const uint32_t c = blockIdx.x * HASH_LANES + lane;
  if (threadIdx.x == 0) s_steps = 0;
  const uint8_t* p = arena;
  uint32_t mlen = 0, nblk = 0;  // message = the change without its first eight bytes (magic + checksum)
  bool hashed = false;
  if (c < n_changes) {
    const uint64_t o = offsets[c], len = offsets[c + 1] - o;
    if (len >= 10 && len < 0xfffffff0ull) {
      hashed = true;
      p = arena + o + 8;
      mlen = (uint32_t)len - 8;
      nblk = (mlen + 9 + 63) / 64;
    }
  }
  __syncthreads();
  if (nblk) atomicMax(&s_steps, nblk);
  __syncthreads();
  const uint32_t steps = s_steps;
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  U4 nxt[4] = {};
  // (a block's 64 bytes are loaded whole when any of them belongs to the message: up to 63 bytes behind the change, inside the
  // 64 bytes of slack every arena is allocated with)
  if (producer && nblk && mlen)
    for (int q = 0; q < 4; q++) nxt[q] = *(const U4*)(p + 16 * q);
  for (uint32_t k = 0; k <= steps; k++) {
    if (producer) {
      if (k < nblk) {
        uint32_t w[16];
        for (int q = 0; q < 4; q++) {
          w[4 * q] = __builtin_bswap32(nxt[q].x);
          w[4 * q + 1] = __builtin_bswap32(nxt[q].y);
          w[4 * q + 2] = __builtin_bswap32(nxt[q].z);
          w[4 * q + 3] = __builtin_bswap32(nxt[q].w);
        }
        const uint32_t o = 64 * k;
        if (k + 1 < nblk && o + 64 < mlen)  // the next block holds message bytes: requested before this block's schedule
          for (int q = 0; q < 4; q++) nxt[q] = *(const U4*)(p + o + 64 + 16 * q);
        if (o + 64 > mlen) {
          // one of the last two blocks: message bytes, 0x80, zeros, and the bit length in the last eight bytes of the last one
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const int rem = (int)mlen - (int)(o + 4 * j);  // message bytes from this word on
            uint32_t v = 0;
            if (rem >= 4) v = w[j];
            else if (rem > 0) v = (w[j] & (0xffffffffu << (32 - 8 * rem))) | (0x80u << (24 - 8 * rem));
            else if (rem == 0) v = 0x80000000u;
            w[j] = v;
          }
          if (k + 1 == nblk) {
            w[14] = mlen >> 29;
            w[15] = mlen << 3;
          }
        }
        V4* dst = wk[k & 1] + lane;
#define AM355_SHA_PUT(r) dst[((r) >> 2) * HASH_LANES] = V4{w[(r)&15] + SHA_K[(r)], w[((r) + 1) & 15] + SHA_K[(r) + 1], w[((r) + 2) & 15] + SHA_K[(r) + 2], w[((r) + 3) & 15] + SHA_K[(r) + 3]};
        AM355_SHA_PUT(0) AM355_SHA_PUT(4) AM355_SHA_PUT(8) AM355_SHA_PUT(12)
#pragma unroll
        for (int r = 16; r < 64; r += 4) {
          AM355_SHA_SCHED(r) AM355_SHA_SCHED(r + 1) AM355_SHA_SCHED(r + 2) AM355_SHA_SCHED(r + 3)
          AM355_SHA_PUT(r)
        }
#undef AM355_SHA_PUT
      }
    } else if (k >= 1 && k - 1 < nblk) {
      const V4* src = wk[(k - 1) & 1] + lane;
      uint32_t a = h[0], b = h[1], cc = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
      for (int r = 0; r < 64; r += 8) {
        const V4 x = src[(r >> 2) * HASH_LANES], y = src[((r >> 2) + 1) * HASH_LANES];
        AM355_SHA_ROUND_WK(a, b, cc, d, e, f, g, hh, x.x)
        AM355_SHA_ROUND_WK(hh, a, b, cc, d, e, f, g, x.y)
        AM355_SHA_ROUND_WK(g, hh, a, b, cc, d, e, f, x.z)
        AM355_SHA_ROUND_WK(f, g, hh, a, b, cc, d, e, x.w)
        AM355_SHA_ROUND_WK(e, f, g, hh, a, b, cc, d, y.x)
        AM355_SHA_ROUND_WK(d, e, f, g, hh, a, b, cc, y.y)
        AM355_SHA_ROUND_WK(cc, d, e, f, g, hh, a, b, y.z)
        AM355_SHA_ROUND_WK(b, cc, d, e, f, g, hh, a, y.w)
      }
      h[0] += a; h[1] += b; h[2] += cc; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    __syncthreads();
  }
  if (producer || c >= n_changes) return;
  min_idx[c] = c;
  V4 lo = V4{0, 0, 0, 0}, hi = V4{0, 0, 0, 0};
  if (hashed) {
    lo = V4{__builtin_bswap32(h[0]), __builtin_bswap32(h[1]), __builtin_bswap32(h[2]), __builtin_bswap32(h[3])};
    hi = V4{__builtin_bswap32(h[4]), __builtin_bswap32(h[5]), __builtin_bswap32(h[6]), __builtin_bswap32(h[7])};
    const uint8_t* q = p - 4;  // the container's checksum field: the first four bytes of the hash (columnar.js:702-704)
    const uint32_t sum = (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24;
    if (sum != lo.x) atomicOr(flags, (uint32_t)F_BAD_CHECKSUM);
  }
  V4* out = (V4*)(hashes + 32 * (size_t)c);
  out[0] = lo;
  out[1] = hi;
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
                                                       uint32_t* __restrict__ fast_flags, uint32_t* __restrict__ dep_idx, uint32_t* __restrict__ self_idx) {
  uint32_t c = blockIdx.x, lane = threadIdx.x;
  if (c >= n) return;
  const ChangeMeta* m = &metas[c];
  if (m->flags) return;
  uint32_t ff = 0;
  if (lane == 0) {
    uint32_t first = hash_find(hashes, tab, mask, min_idx, hashes + 32 * (size_t)c);  // first change of the batch with this hash
    if (first != c) ff |= FF_DUP_HASH;
    self_idx[c] = first;
  }
  const uint8_t* deps = arena + m->base + m->deps_off;
  for (uint32_t k = lane; k < m->n_deps; k += WAVE) {
    uint32_t d = hash_find(hashes, tab, mask, min_idx, deps + 32 * (size_t)k);
    // for the host's general scheduler: dependency -> index of the change it names (NONE32: not in the batch), addressed by the
    // dependency's place in the arena (32-byte hashes never overlap, so byte offset / 32 is a unique slot)
    dep_idx[(m->base + m->deps_off + 32 * (uint64_t)k) >> 5] = d;
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
// `distinct`: word 0 = number of claimed slots, words [1, 1+CAP) their slot indexes, then (8-byte aligned at word
// 2+CAP) CAP 64-bit slot values -- everything the host needs about the actor table in one small copy
constexpr uint32_t DISTINCT_CAP = 4096;
constexpr uint32_t PLAN_RANK_MAX = 1024;   // distinct actors ranked on the device (LDS: 40 bytes each)
constexpr uint32_t PLAN_ID_MAX = 32;       // bytes of an actor id the device ranking handles (ids are 16 bytes in practice)
// `rank_ids`: what the ranking workgroup (rank_actors) needs about distinct actor k, left by the lane that claimed its slot -- the id as
// four big-endian words, zero padded, and its length -- so that the ranking is ONE memory round trip (count and records side by side)
// instead of count -> slot value -> id bytes in the arena.
struct RankId {
  unsigned long long w[PLAN_ID_MAX / 8];
  uint32_t len, pad;
};
__device__ __forceinline__ uint32_t actor_find_or_insert(const uint8_t* __restrict__ arena, unsigned long long* __restrict__ slots, uint32_t mask,
                                                         uint32_t off, uint32_t len, uint32_t* __restrict__ distinct, RankId* __restrict__ rank_ids) {
  const uint8_t* p = arena + off;
  // actor ids are 16 bytes in practice: two 8-byte loads (any alignment) instead of a dependent chain of sixteen byte loads for the
  // hash, and again for every comparison (each load of such a chain is a round trip to L2: this kernel is bound by them)
  const bool wide = len == 16;
  uint64_t p0 = 0, p1 = 0, h = 0xcbf29ce484222325ull;
  if (wide) {
    p0 = load_u64_unaligned(p);
    p1 = load_u64_unaligned(p + 8);
    h = (p0 ^ (p1 * 0x9e3779b97f4a7c15ull)) * 0xff51afd7ed558ccdull;
    h ^= h >> 29;
  } else {
    for (uint32_t k = 0; k < len; k++) h = (h ^ p[k]) * 0x100000001b3ull;
    h = (h ^ h >> 33) * 0xff51afd7ed558ccdull;  // (FNV-1a keeps the last bytes in the low bits: spread them before the slot is cut out)
  }
  uint32_t i = (uint32_t)(h >> 20) & mask;
  unsigned long long mine = ((unsigned long long)off + 1) << 16 | len;
  for (uint32_t probes = 0; probes <= mask; probes++) {
    unsigned long long v = slots[i];
    if (v == 0) {
      v = atomicCAS(&slots[i], 0ull, mine);
      if (v == 0) {
        uint32_t k = atomicAdd(&distinct[0], 1u);
        if (k < DISTINCT_CAP) {
          distinct[1 + k] = i;
          ((unsigned long long*)(distinct + 2 + DISTINCT_CAP))[k] = mine;
        }
        if (k < PLAN_RANK_MAX) {
          RankId r;
          for (uint32_t wd = 0; wd < PLAN_ID_MAX / 8; wd++) r.w[wd] = 0;
          r.len = len;
          r.pad = 0;
          if (wide) {
            r.w[0] = __builtin_bswap64(p0);
            r.w[1] = __builtin_bswap64(p1);
          } else if (len <= PLAN_ID_MAX) {
            for (uint32_t b = 0; b < len; b++) r.w[b >> 3] |= (unsigned long long)p[b] << (56 - 8 * (b & 7));
          }
          rank_ids[k] = r;
        }
        return i;
      }
    }
    if ((uint32_t)(v & 0xffff) == len) {
      const uint8_t* q = arena + ((v >> 16) - 1);
      bool eq = true;
      if (wide) eq = load_u64_unaligned(q) == p0 && load_u64_unaligned(q + 8) == p1;
      else for (uint32_t k = 0; k < len && eq; k++) eq = p[k] == q[k];
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
                                                        uint32_t* __restrict__ flags, uint32_t* __restrict__ fast_flags, uint32_t* __restrict__ distinct,
                                                        RankId* __restrict__ rank_ids) {
  __shared__ uint32_t s_off[WAVE], s_len[WAVE];
  wave_priority_high();
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
    uint32_t s = actor_find_or_insert(arena, slots, mask, abs0 + m->actor_off, m->actor_len, distinct, rank_ids);
    if (s == NONE32) { atomicOr(flags, (uint32_t)F_UNSUPPORTED); s = 0; }
    amap[base] = s;
    m->author_slot = s;
    atomicMin(&first_idx[s], c);
  }
  // The table of the other actors is length-prefixed, so finding entry k means walking entries 0 .. k-1 -- unless every entry has the
  // same one-byte length (ids of 16 bytes: the ordinary case), which all lanes check at once: entry k then sits at a fixed stride.
  if (n_other && n_other <= WAVE) {
    const uint32_t l0 = p[m->others_off], stride = l0 + 1;
    bool fits = l0 < 0x80 && (uint64_t)m->others_off + (uint64_t)stride * n_other <= m->len;
    bool mine_ok = !fits || lane >= n_other || p[m->others_off + stride * lane] == l0;
    if (fits && __ballot(!mine_ok) == 0) {
      if (lane < n_other) {
        uint32_t t = actor_find_or_insert(arena, slots, mask, abs0 + m->others_off + stride * lane + 1, l0, distinct, rank_ids);
        if (t == NONE32) { atomicOr(flags, (uint32_t)F_UNSUPPORTED); t = 0; }
        amap[base + 1 + lane] = t;
      }
      return;
    }
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
      uint32_t t = s_len[lane] != NONE32 ? actor_find_or_insert(arena, slots, mask, s_off[lane], s_len[lane], distinct, rank_ids) : NONE32;
      if (t == NONE32) { atomicOr(flags, (uint32_t)F_UNSUPPORTED); t = 0; }
      amap[base + 1 + k0 + lane] = t;
    }
    __syncthreads();
  }
}

// every actor a change mentions must already be in the document when the change is read (new.js:1442-1449):
// with in-order application that means its first change has an index <= this one
__host__ __device__ __forceinline__ int wave_class_of(const ChangeMeta& m);

// Device half of the in-order plan, part 1 (with k_plan_apply below): while it writes the digests, every workgroup also publishes the
// sums of its 256 changes (ops, preds, actor entries, plans per decoder class), and one EXTRA workgroup (blockIdx == gridDim - 1)
// ranks the distinct actor ids lexicographically from LDS. plan_words: [0] fallback, [1] max op id, [2] OR of the changes' validity
// flags, [3] unknown columns seen (cleared by the caller).
constexpr uint32_t PLAN_SUMS = 8;          // words per workgroup in block_sums: ops, preds, entries, small, large, serial plans

__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long x, uint32_t lane) {
  for (int d = 1; d < WAVE; d <<= 1) {
    unsigned long long y = __shfl_up(x, (unsigned)d);
    if (lane >= (uint32_t)d) x += y;
  }
  return x;
}
// exclusive prefix over a BLOCK-thread workgroup of three 64-bit counters at once; totals returned through t[]
__device__ __forceinline__ void block_scan3(unsigned long long a, unsigned long long b, unsigned long long c, unsigned long long (*s)[3], unsigned long long ex[3],
                                            unsigned long long t[3]) {
  const uint32_t lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  unsigned long long ia = wave_incl_scan_u64(a, lane), ib = wave_incl_scan_u64(b, lane), ic = wave_incl_scan_u64(c, lane);
  if (lane == WAVE - 1) { s[w][0] = ia; s[w][1] = ib; s[w][2] = ic; }
  __syncthreads();
  unsigned long long ba = 0, bb = 0, bc = 0, ta = 0, tb = 0, tc = 0;
  for (uint32_t k = 0; k < BLOCK / WAVE; k++) {
    unsigned long long xa = s[k][0], xb = s[k][1], xc = s[k][2];
    if (k < w) { ba += xa; bb += xb; bc += xc; }
    ta += xa; tb += xb; tc += xc;
  }
  __syncthreads();
  ex[0] = ba + ia - a; ex[1] = bb + ib - b; ex[2] = bc + ic - c;
  t[0] = ta; t[1] = tb; t[2] = tc;
}

// lexicographic ranks of the distinct actor ids (a proper prefix sorts first) = order of the hex strings (new.js:65); one workgroup
__device__ void rank_actors(const uint32_t* __restrict__ distinct, const RankId* __restrict__ rank_ids, uint32_t* __restrict__ slot_rank,
                            uint32_t* __restrict__ plan_words) {
  __shared__ unsigned long long s_id[PLAN_RANK_MAX][PLAN_ID_MAX / 8];  // big-endian words, zero padded
  __shared__ uint32_t s_len[PLAN_RANK_MAX];
  __shared__ uint32_t s_fallback;
  static_assert(PLAN_RANK_MAX % BLOCK == 0 && PLAN_RANK_MAX <= DISTINCT_CAP, "rank_actors: records per thread");
  constexpr uint32_t PER = PLAN_RANK_MAX / BLOCK;
  const uint32_t t = threadIdx.x;
  // the records and the slot indexes of this thread's actors are requested before the count is known (the tables are allocated in
  // full; entries beyond the count hold leftovers and are not looked at): one round trip for everything
  RankId rec[PER];
  uint32_t slot_of[PER];
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    rec[k] = rank_ids[t + k * BLOCK];
    slot_of[k] = distinct[1 + t + k * BLOCK];
  }
  const uint32_t nd = distinct[0];
  if (t == 0) s_fallback = nd > PLAN_RANK_MAX ? 1u : 0u;
  __syncthreads();
  if (nd <= PLAN_RANK_MAX) {
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t a = t + k * BLOCK;
      if (a < nd) {
        s_len[a] = rec[k].len;
        if (rec[k].len > PLAN_ID_MAX) s_fallback = 1;
        for (uint32_t wd = 0; wd < PLAN_ID_MAX / 8; wd++) s_id[a][wd] = rec[k].w[wd];
      }
    }
  }
  __syncthreads();
  if (s_fallback) { if (t == 0) plan_words[0] = 1; return; }
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t a = t + k * BLOCK;
    if (a >= nd) break;
    uint32_t rank = 0;
    const uint32_t my_len = rec[k].len;
    static_assert(PLAN_ID_MAX == 32, "rank_actors compares four 64-bit words");
    const unsigned long long m0 = rec[k].w[0], m1 = rec[k].w[1], m2 = rec[k].w[2], m3 = rec[k].w[3];
    for (uint32_t j = 0; j < nd; j++) {
      const unsigned long long x0 = s_id[j][0], x1 = s_id[j][1], x2 = s_id[j][2], x3 = s_id[j][3];
      // equal up to the padding: the shorter id first (distinct ids differ somewhere)
      const bool less = x0 != m0 ? x0 < m0 : x1 != m1 ? x1 < m1 : x2 != m2 ? x2 < m2 : x3 != m3 ? x3 < m3 : s_len[j] < my_len;
      rank += less ? 1u : 0u;
    }
    slot_rank[slot_of[k]] = rank;
  }
}

// every actor a change mentions must already have a change in the document when the change is read: one wavefront per change, one
// lane per entry of its actor table (a lane per change walking its 64 entries was 64 pairs of dependent loads: most of k_actor_check)
__global__ __launch_bounds__(WAVE) void k_actor_first(ChangeMeta* __restrict__ metas, uint32_t n, const uint32_t* __restrict__ amap_base,
                                                       const uint32_t* __restrict__ amap, uint32_t amap_cap, const uint32_t* __restrict__ first_idx,
                                                       uint32_t* __restrict__ fast_flags) {
  wave_priority_high();
  uint32_t c = blockIdx.x, lane = threadIdx.x;
  if (c >= n) return;
  ChangeMeta* m = &metas[c];
  if (m->flags) return;
  uint32_t base = amap_base[c], ne = m->n_entries;
  if ((uint64_t)base + ne > amap_cap) return;
  uint32_t mx = 0;
  for (uint32_t k = lane; k < ne; k += WAVE) {
    uint32_t f = first_idx[amap[base + k]];
    mx = f > mx ? f : mx;  // NONE32 (no change by that actor in the batch) also lands on the general path
  }
  for (int d = WAVE / 2; d >= 1; d >>= 1) {
    uint32_t o = __shfl_xor(mx, d);
    mx = o > mx ? o : mx;
  }
  if (lane == 0) {
    m->max_first = mx;
    if (mx > c) atomicOr(fast_flags, (uint32_t)FF_LATE_ACTOR);
  }
}

// (The ranking of the distinct actor ids -- one workgroup, ~20 us of dependent loads: count -> slot -> id bytes -- rides as the LAST
// workgroup of k_actor_check: both only need what k_actor_intern left, and the consumer of the ranks is the kernel behind. As a
// kernel of its own in front of k_actor_check it was 20 us of the critical path for one workgroup's worth of work.)
__global__ __launch_bounds__(BLOCK) void k_actor_check(const uint8_t* __restrict__ arena, ChangeMeta* __restrict__ metas, uint32_t n,
                                                       const uint32_t* __restrict__ amap_base, const uint32_t* __restrict__ amap, uint32_t amap_cap,
                                                       const uint32_t* __restrict__ first_idx, uint32_t* __restrict__ flags, uint32_t* __restrict__ fast_flags,
                                                       ChangeBrief* __restrict__ briefs, const uint32_t* __restrict__ distinct, const RankId* __restrict__ rank_ids,
                                                       uint32_t* __restrict__ slot_rank, unsigned long long* __restrict__ block_sums, uint32_t* __restrict__ plan_words,
                                                       ChangeBrief* __restrict__ host_briefs) {
  wave_priority_high();
  __shared__ unsigned long long s_scan[BLOCK / WAVE][3];
  if (blockIdx.x + 1 == gridDim.x) {  // the extra workgroup
    rank_actors(distinct, rank_ids, slot_rank, plan_words);
    return;
  }
  uint32_t c = gtid();
  const bool in_range = c < n;
  ChangeBrief br{};
  if (in_range) {
    ChangeMeta* m = &metas[c];
    br.seq = m->seq;
    br.start_op = (uint32_t)m->start_op;
    br.n_ops = m->n_ops;
    br.n_preds = m->n_preds;
    br.n_entries = m->n_entries;
    br.author_slot = m->author_slot;
    br.flags_fits = m->flags;
    if (!m->flags) {
      int wc = wave_class_of(*m);
      if (wc >= 1) br.flags_fits |= 0x80000000u;
      if (wc == 2) br.flags_fits |= 0x40000000u;
      if (m->pad & 1) br.flags_fits |= 0x20000000u;  // the change carries columns this engine does not model
    }
    briefs[c] = br;
    // (the host's half of the plan reads the briefs: written into its pinned memory right here -- 32 bytes per change over the link --
    // they are there when the NEXT kernel, k_plan_apply, signals; a D2H copy behind that kernel cost an event record in the main
    // stream, a copy dispatch on another and a blocking wait on the host, which is the critical path between plan and k_resolve)
    if (host_briefs) host_briefs[c] = br;
  }
  // ---- sums of this workgroup's changes for k_plan_apply (a malformed change counts nothing: the host rejects the batch on its flags) ----
  const bool valid = in_range && !(br.flags_fits & 0x1fffffffu);
  const bool has = valid && br.n_ops != 0;
  const bool small = has && (br.flags_fits & 0x40000000u), large = has && !small && (br.flags_fits & 0x80000000u), serial = has && !small && !large;
  unsigned long long ex[3], t1[3], t2[3];
  block_scan3(valid ? br.n_ops : 0u, valid ? br.n_preds : 0u, valid ? br.n_entries : 0u, s_scan, ex, t1);
  block_scan3(small ? 1u : 0u, large ? 1u : 0u, serial ? 1u : 0u, s_scan, ex, t2);
  uint32_t mx_op = has ? br.start_op + br.n_ops - 1 : 0u, bad = in_range ? (br.flags_fits & 0x1fffffffu) : 0u, unknown = in_range ? (br.flags_fits & 0x20000000u) : 0u;
  for (int d = WAVE / 2; d >= 1; d >>= 1) {
    uint32_t o = __shfl_xor(mx_op, d), ob = __shfl_xor(bad, d), ou = __shfl_xor(unknown, d);
    mx_op = o > mx_op ? o : mx_op;
    bad |= ob;
    unknown |= ou;
  }
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    if (mx_op) atomicMax(&plan_words[1], mx_op);
    if (bad) atomicOr(&plan_words[2], bad);
    if (unknown) atomicOr(&plan_words[3], 1u);
  }
  if (threadIdx.x == 0) {
    unsigned long long* out = block_sums + (size_t)blockIdx.x * PLAN_SUMS;
    out[0] = t1[0]; out[1] = t1[1]; out[2] = t1[2]; out[3] = t2[0]; out[4] = t2[1]; out[5] = t2[2];
  }
}

// part 2 (same grid without the extra workgroup): prefix sums in input order -> the ChangePlan of every change with ops. plans: [n] -- the
// small class fills it from the front, the large class from the back (a plan's place inside its class does not matter: every plan is
// an independent unit of decode work); plans_serial: [n] the changes left to the lane-serial decoder. The last workgroup reports the
// totals (and the stage-1 words the host decides on) through HostSignals.
__global__ __launch_bounds__(BLOCK) void k_plan_apply(const ChangeBrief* __restrict__ briefs, uint32_t n, const uint32_t* __restrict__ slot_rank, uint32_t slot_mask,
                                                      const unsigned long long* __restrict__ block_sums, ChangePlan* __restrict__ plans,
                                                      ChangePlan* __restrict__ plans_serial, const uint32_t* __restrict__ words, const uint32_t* __restrict__ plan_words,
                                                      const uint32_t* __restrict__ distinct, HostSignals* sig, uint32_t seq, PlanTotals* __restrict__ dev_totals,
                                                      uint32_t* __restrict__ host_s1) {
  wave_priority_high();
  __shared__ unsigned long long s_scan[BLOCK / WAVE][3];
  __shared__ unsigned long long s_base[6];
  // sums of the workgroups before this one (L2 hits: a few words per workgroup)
  {
    unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += BLOCK) {
      const unsigned long long* q = block_sums + (size_t)j * PLAN_SUMS;
      p0 += q[0]; p1 += q[1]; p2 += q[2]; p3 += q[3]; p4 += q[4]; p5 += q[5];
    }
    unsigned long long ex[3], t1[3], t2[3];
    block_scan3(p0, p1, p2, s_scan, ex, t1);
    block_scan3(p3, p4, p5, s_scan, ex, t2);
    if (threadIdx.x == 0) { s_base[0] = t1[0]; s_base[1] = t1[1]; s_base[2] = t1[2]; s_base[3] = t2[0]; s_base[4] = t2[1]; s_base[5] = t2[2]; }
    __syncthreads();
  }
  const uint32_t c = gtid();
  const bool in_range = c < n;
  ChangeBrief br{};
  if (in_range) br = briefs[c];
  const bool valid = in_range && !(br.flags_fits & 0x1fffffffu);
  const bool has = valid && br.n_ops != 0;
  const bool small = has && (br.flags_fits & 0x40000000u), large = has && !small && (br.flags_fits & 0x80000000u), serial = has && !small && !large;
  unsigned long long e1[3], e2[3], t1[3], t2[3];
  block_scan3(valid ? br.n_ops : 0u, valid ? br.n_preds : 0u, valid ? br.n_entries : 0u, s_scan, e1, t1);
  block_scan3(small ? 1u : 0u, large ? 1u : 0u, serial ? 1u : 0u, s_scan, e2, t2);
  if (has) {
    // (author_slot is only meaningful once k_actor_intern has run for the change: a capacity retry leaves it unset in the first attempt)
    ChangePlan pl{c, (uint32_t)(s_base[0] + e1[0]), (uint32_t)(s_base[1] + e1[1]), (uint32_t)(s_base[2] + e1[2]),
                  br.author_slot <= slot_mask ? slot_rank[br.author_slot] : 0u, br.n_entries};
    if (small) plans[(uint32_t)(s_base[3] + e2[0])] = pl;
    else if (large) plans[n - 1 - (uint32_t)(s_base[4] + e2[1])] = pl;
    else plans_serial[(uint32_t)(s_base[5] + e2[2])] = pl;
  }
  if (blockIdx.x + 1 == gridDim.x && host_s1) {
    // the stage-1 words and the distinct actor ids (count, slots, (offset, length) records: what the host ranks the actors from) into
    // the host's mirror of the block -- written by earlier kernels, so complete; in place before the signal below
    const uint32_t nd = distinct[0] < DISTINCT_CAP ? distinct[0] : DISTINCT_CAP;
    const uint32_t* src = words;  // (the block: 16 words | distinct[0 .. ] at word 16)
    for (uint32_t i = threadIdx.x; i < 16; i += BLOCK) host_s1[i] = src[i];
    uint32_t* hd = host_s1 + 16;
    for (uint32_t i = threadIdx.x; i < 1 + nd; i += BLOCK) hd[i] = distinct[i];
    const unsigned long long* ids = (const unsigned long long*)(distinct + 2 + DISTINCT_CAP);
    unsigned long long* hids = (unsigned long long*)(hd + 2 + DISTINCT_CAP);
    for (uint32_t i = threadIdx.x; i < nd; i += BLOCK) hids[i] = ids[i];
    __threadfence_system();
    __syncthreads();
  }
  if (blockIdx.x + 1 == gridDim.x && threadIdx.x == 0) {
    unsigned long long ops = s_base[0] + t1[0], preds = s_base[1] + t1[1], ent = s_base[2] + t1[2];
    PlanTotals z{};
    z.n_ops = (uint32_t)ops; z.n_preds = (uint32_t)preds; z.n_entries = (uint32_t)ent;
    z.n_small = (uint32_t)(s_base[3] + t2[0]); z.n_large = (uint32_t)(s_base[4] + t2[1]); z.n_serial = (uint32_t)(s_base[5] + t2[2]);
    z.max_op = plan_words[1];
    z.fallback = (plan_words[0] || ops >= 0x7ffffff0ull || preds >= 0xfffffff0ull || ent >= 0xfffffff0ull) ? 1u : 0u;
    z.flags_a = words[0] | plan_words[2]; z.fast_a = words[1]; z.total_entries = words[2]; z.n_distinct = distinct[0];
    z.reserved[0] = plan_words[3];  // some change carries columns this engine does not model (the reference's save keeps them)
    if (dev_totals) *dev_totals = z;  // (for the decode kernels enqueued right behind this one: they start without the host in between)
    signal_host((uint32_t*)&sig->plan, (const uint32_t*)&z, sizeof(PlanTotals) / 4, &sig->plan_seq, seq);
  }
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
    return r.last_len == v.len && bytes_equal(r.c.p, r.last_off, v.off, v.len, r.c.len);
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
enum Task { T_OBJ, T_KEY, T_KEYSTR, T_INSERT, T_ACTION, T_VALUE, T_PREDNUM, T_PREDS, T_NUM, T_ID = T_NUM, T_NUM_DOC };  // T_ID: documents only

// change-local actor index -> global actor rank. `amap` holds either final ranks (slot_rank == nullptr: table built by
// the host scheduler) or device actor-table slots that `slot_rank` maps to ranks.
struct ActorXlate {
  const uint32_t* amap;
  const uint32_t* slot_rank;
  uint32_t shard_rank = 0, shard_world = 1;   // objectId sharding (am355_set_shard): which objects' rows this rank merges
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
      if (pl.author != NONE32) {
        // ops carry no id columns in a change: op i is (startOp + i, author)  (new.js:708-709)
        o.id_ctr[base + i] = (uint32_t)m->start_op + i;
        o.id_actor[base + i] = pl.author;
      }
    }
  } else if (task == T_ID) {
    // documents store the op ids explicitly (idActor / idCtr columns, columnar.js:62-63)
    Rle a;
    Delta c;
    rle_init(a, p + m->col_off[C_ID_ACTOR], m->col_len[C_ID_ACTOR]);
    rle_init(c.r, p + m->col_off[C_ID_CTR], m->col_len[C_ID_CTR]);
    c.abs = 0;
    for (uint32_t i = 0; i < n; i++) {
      RVal va, vc;
      if (!rle_next<RT_UINT>(a, va) || !delta_next(c, vc)) { err |= F_BAD_RLE; break; }
      if (va.is_null || vc.is_null || vc.i <= 0) err |= F_BAD_ROW;
      if (!vc.is_null && (vc.i < 0 || (uint64_t)vc.i >= NONE32)) err |= F_OVERFLOW;
      o.id_actor[base + i] = va.is_null ? 0 : xlate_actor(amap, pl, va.i, err);
      o.id_ctr[base + i] = vc.is_null ? 0 : (uint32_t)vc.i;
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

// ---------------------------------------------------------------------------------------------------------
// k_decode_wave: one wavefront per change, run-level decode.
//
// The lane-serial decoder above walks every VALUE of a column; here the sequential part is one step per RECORD.
// For each column: (1) the bytes are staged in LDS with coalesced loads; (2) LEB128 tokens are found in parallel --
// a byte with bit 7 clear ends a number, so one ballot per 64 bytes plus popcount/clz gives every token's index
// and start, and each terminating lane assembles its token; (3) lane 0 walks the records over the token table
// (a literal of k values is ONE step); (4) all lanes expand runs into rows (binary search in the run table), apply
// wave prefix sums where the format needs them (delta columns, value offsets, pred list offsets) and store rows
// with consecutive lanes writing consecutive addresses. Columns longer than the large class's COLMAX bytes (rare: the change is
// then routed to the lane-serial kernel by the host) and the UTF-8 key column (string bytes are not LEB tokens;
// walked by lane 0) are the exceptions.
// ---------------------------------------------------------------------------------------------------------
// LDS working set of one wave, in two sizes: the number of waves a CU can hold is set by this struct (160 KB of LDS per
// CU), and typical changes (a few hundred ops, columns of ~100 bytes) need a fraction of what the largest
// wave-decodable ones do. COLMAX = longest tokenised column, REGION = all columns staged at once when they fit,
// ACTMAX = entries of the change's actor table.
template <uint32_t COLMAX_, uint32_t REGION_, uint32_t ACTMAX_>
struct WaveLdsT {
  static constexpr uint32_t COLMAX = COLMAX_, REGION = REGION_, ACTMAX = ACTMAX_;
  static constexpr uint32_t RUNMAX = COLMAX_ / 2;  // every record is at least two tokens
  alignas(16) uint8_t region[REGION_];
  uint8_t bytes[COLMAX_];
  uint64_t tok[COLMAX_];  // one LEB128 number per word: see wv_pack_token
  uint32_t aux[COLMAX_ / 2 + 1];  // key column: string length of each run
  uint32_t run_start[COLMAX_ / 2 + 1], run_tok[COLMAX_ / 2];
  uint8_t run_kind[COLMAX_ / 2];
  uint32_t n_runs, n_tokens, total_rows, err;
  uint32_t rank[ACTMAX_];
};
using WaveLdsSmall = WaveLdsT<256, 1024, 64>;     // ~5.4 KB: 29 waves per CU
using WaveLdsLarge = WaveLdsT<1024, 4096, 1024>;  // ~24 KB: 6 waves per CU
enum { RK_REP = 1, RK_LIT = 2, RK_NUL = 3 };

// 2: small wave class, 1: large wave class, 0: lane-serial decoder
__host__ __device__ __forceinline__ int wave_class_of(const ChangeMeta& m) {
  const int tokenised[] = {C_OBJ_ACTOR, C_OBJ_CTR, C_KEY_ACTOR, C_KEY_CTR, C_INSERT, C_ACTION, C_VAL_LEN, C_PRED_NUM, C_PRED_ACTOR, C_PRED_CTR};
  uint32_t longest = 0;
  for (int k = 0; k < 10; k++) longest = m.col_len[tokenised[k]] > longest ? m.col_len[tokenised[k]] : longest;
  if (longest <= WaveLdsSmall::COLMAX && m.n_entries <= WaveLdsSmall::ACTMAX) return 2;
  if (longest <= WaveLdsLarge::COLMAX && m.n_entries <= WaveLdsLarge::ACTMAX) return 1;
  return 0;
}

// A LEB128 number as one LDS word, so that reading a value is one ds_read instead of four: bits 0-52 the value (53 bits, JS safe
// integers; for a negative number the low 53 bits of its two's complement), 53-56 the byte count, 57 the sign bit of the last
// byte, 58 / 59 "valid as unsigned" / "valid as signed" (encoding.js:389-488: at most 10 bytes, within +-(2^53 - 1)).
__device__ __forceinline__ uint64_t wv_pack_token(uint64_t raw, uint32_t nb, uint32_t last) {
  const uint64_t M53 = (1ull << 53) - 1;
  bool len_ok = nb >= 1 && nb <= 10;
  bool ok_u = len_ok && !(nb == 10 && (last & 0xfe)) && raw <= MAX_SAFE;
  uint64_t sv = raw;
  if ((last & 0x40) && 7 * nb < 64) sv |= ~0ull << (7 * nb);
  bool ok_s = len_ok && !(nb == 10 && last != 0 && last != 0x7f) && (int64_t)sv <= (int64_t)MAX_SAFE && (int64_t)sv >= -(int64_t)MAX_SAFE;
  return (sv & M53) | (uint64_t)(nb <= 15 ? nb : 15) << 53 | (uint64_t)((last >> 6) & 1) << 57 | (uint64_t)ok_u << 58 | (uint64_t)ok_s << 59;
}
template <class WL>
__device__ __forceinline__ bool wv_tok_uint(const WL& L, uint32_t t, uint64_t& v) {
  uint64_t w = L.tok[t];
  uint32_t bits = 7 * (uint32_t)((w >> 53) & 15);
  v = w & ((1ull << (bits < 53 ? bits : 53)) - 1);  // the field is sign-extended from bit 7*nb: the unsigned value is below it
  return (w >> 58) & 1;
}
template <class WL>
__device__ __forceinline__ bool wv_tok_sint(const WL& L, uint32_t t, int64_t& out) {
  uint64_t w = L.tok[t];
  uint64_t v = w & ((1ull << 53) - 1);
  if ((w >> 57) & 1) v |= ~0ull << 53;  // a valid negative number is >= -(2^53 - 1): every higher bit is set
  out = (int64_t)v;
  return (w >> 59) & 1;
}

#ifndef AM355_WV_SERIAL_TOKENS
#define AM355_WV_SERIAL_TOKENS 40   // (built with 2 once per change to this file: every column of every fixture and mutation campaign through the wavefront walk)
#endif
constexpr uint32_t WV_SERIAL_TOKENS = AM355_WV_SERIAL_TOKENS;
constexpr uint32_t WV_SERIAL_RECORDS = 12;
__device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t x, uint32_t lane);

// what token t says when read as a record header: kind, rows, the header after it (T = none), the first error the serial walk would
// find AT this record given the kind of the record in front (0 = no error)
template <class WL>
__device__ __forceinline__ void wv_header(const WL& L, uint32_t t, uint32_t T, uint32_t& kind, uint64_t& count, uint32_t& next, uint32_t& err_self, uint32_t& err_after_same) {
  kind = 0; count = 0; next = T; err_self = 0; err_after_same = 0;
  int64_t cnt;
  if (!wv_tok_sint(L, t, cnt)) { err_self = F_BAD_LEB; return; }
  if (cnt > 1) {
    kind = RK_REP; count = (uint64_t)cnt;
    if (t + 1 >= T) { err_self = F_BAD_LEB; return; }
    next = t + 2;
  } else if (cnt == 1) {
    err_self = F_BAD_RLE;  // repetition count of 1
  } else if (cnt < 0) {
    kind = RK_LIT; count = (uint64_t)(-cnt);
    err_after_same = F_BAD_RLE;  // successive literals (checked before the length of the literal)
    if (count > (uint64_t)(T - t - 1)) { err_self = F_BAD_LEB; return; }
    next = t + 1 + (uint32_t)count;
  } else {
    kind = RK_NUL;
    err_after_same = F_BAD_RLE;  // successive null runs (checked before the run's own length)
    uint64_t z;
    if (t + 1 >= T || !wv_tok_uint(L, t + 1, z)) { err_self = F_BAD_LEB; return; }
    if (z == 0) { err_self = F_BAD_RLE; return; }
    count = z;
    next = t + 2;
  }
  if (next > T) next = T;
}

// The record tables of the column whose T tokens are in L.tok, by the whole wavefront (see wv_load_column). Scratch: L.bytes (the
// column's bytes are not needed once tokenised) holds one mark per token, L.aux (the key column's, unused here) the jump table.
template <class WL>
__device__ __forceinline__ void wv_records_parallel(WL& L, uint32_t T, uint32_t lane) {
  uint16_t* jump = (uint16_t*)L.aux;   // [COLMAX] 16-bit entries in (COLMAX / 2 + 1) words
  uint8_t* mark = L.bytes;
  for (uint32_t t = lane; t < T; t += WAVE) {
    uint32_t kind, nx, e1, e2; uint64_t cnt;
    wv_header(L, t, T, kind, cnt, nx, e1, e2);
    jump[t] = (uint16_t)(e1 ? T : nx);   // (a header in error ends the chain: the walk stops there)
    mark[t] = t == 0 ? 1 : 0;
  }
  __syncthreads();
  // orbit of token 0: after round k every one of the first 2^k headers is marked
  for (uint32_t hop = 1; hop < T; hop <<= 1) {
    uint16_t nj[WL::COLMAX / WAVE];   // (fixed trip count: the new jumps stay in registers)
#pragma unroll
    for (uint32_t k = 0; k < WL::COLMAX / WAVE; k++) {
      const uint32_t t = lane + k * WAVE;
      nj[k] = (uint16_t)T;
      if (t < T) {
        const uint32_t j = jump[t];
        if (mark[t] && j < T) mark[j] = 1;
        if (j < T) nj[k] = jump[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < WL::COLMAX / WAVE; k++) {
      const uint32_t t = lane + k * WAVE;
      if (t < T) jump[t] = nj[k];
    }
    __syncthreads();
    if (jump[0] >= T) {   // the chain from token 0 has reached the end: one more marking pass with the final jumps is still due
      for (uint32_t t = lane; t < T; t += WAVE) { const uint32_t j = jump[t]; if (mark[t] && j < T) mark[j] = 1; }
      __syncthreads();
      break;
    }
  }
  // records in order: rank of every marked token, its kind / first value token / rows; first error in chain order
  uint32_t base = 0, first_bad = NONE32, first_err = 0;
  for (uint32_t chunk = 0; chunk < T; chunk += WAVE) {
    const uint32_t t = chunk + lane;
    const bool m = t < T && mark[t];
    const unsigned long long mm = __ballot(m);
    if (m) {
      const uint32_t r = base + (uint32_t)__popcll(mm & ((1ull << lane) - 1));
      uint32_t kind, nx, e1, e2; uint64_t cnt;
      wv_header(L, t, T, kind, cnt, nx, e1, e2);
      if (r < WL::RUNMAX) {
        L.run_kind[r] = (uint8_t)kind;
        L.run_tok[r] = t + 1;
        L.run_start[r] = cnt > 0xfffffff0ull ? 0xfffffff1u : (uint32_t)cnt;   // rows of the record for now: scanned below
      }
    }
    base += (uint32_t)__popcll(mm);
  }
  const uint32_t R = base < WL::RUNMAX ? base : WL::RUNMAX;
  __syncthreads();
  // errors: a record's own, or "same kind as the record in front" for literals and null runs -- the earliest record decides
  for (uint32_t r = lane; r < R; r += WAVE) {
    const uint32_t t = L.run_tok[r] - 1;
    uint32_t kind, nx, e1, e2; uint64_t cnt;
    wv_header(L, t, T, kind, cnt, nx, e1, e2);
    uint32_t e = 0;
    if (!kind && e1) e = e1;                                              // not a count at all / repetition count of 1
    else if (e2 && r > 0 && L.run_kind[r - 1] == kind) e = e2;            // successive literals / null runs
    else e = e1;
    if (e && r < first_bad) { first_bad = r; first_err = e; }
  }
  for (int d = WAVE / 2; d; d >>= 1) {
    const uint32_t ob = __shfl_xor(first_bad, d), oe = __shfl_xor(first_err, d);
    if (ob < first_bad) { first_bad = ob; first_err = oe; }
  }
  // rows before every record: exclusive scan of the record lengths (the serial walk stops at the record in error: so does the scan)
  const uint32_t R_ok = first_bad < R ? first_bad : R;
  uint64_t carry = 0;
  uint32_t overflow = 0;
  for (uint32_t chunk = 0; chunk < R_ok; chunk += WAVE) {
    const uint32_t r = chunk + lane;
    const uint64_t c = r < R_ok ? (uint64_t)L.run_start[r] : 0;
    const uint64_t incl = carry + (uint64_t)wave_incl_scan_i64((int64_t)c, lane);
    if (r < R_ok) {
      if (incl > 0xfffffff0ull) overflow = 1;
      L.run_start[r] = (uint32_t)(incl - c);
    }
    carry = (uint64_t)__shfl((long long)incl, WAVE - 1);
  }
  const bool any_overflow = __ballot(overflow) != 0;
  __syncthreads();
  if (lane == 0) {
    L.run_start[R_ok] = (uint32_t)carry;
    L.n_runs = R_ok;
    L.n_tokens = T;
    L.total_rows = (uint32_t)carry;
    L.err = first_bad < R ? first_err : any_overflow ? (uint32_t)F_OVERFLOW : 0u;
  }
  __syncthreads();
}

// Stage + tokenise + record-walk one RLE column (uint or int values). All lanes must call it.
template <class WL, class P>
__device__ __forceinline__ void wv_load_column(WL& L, P col, uint32_t len, uint32_t lane) {
  __syncthreads();  // previous column's readers are done
  for (uint32_t i = lane; i < len; i += WAVE) L.bytes[i] = col[i];
  if (lane == 0) L.err = 0;
  __syncthreads();
  uint32_t tok_base = 0, carry_start = 0;
  for (uint32_t chunk = 0; chunk < len; chunk += WAVE) {
    uint32_t pos = chunk + lane;
    bool in = pos < len;
    uint32_t b = in ? L.bytes[pos] : 0x80;
    bool term = in && !(b & 0x80);
    unsigned long long mask = __ballot(term);
    if (term) {
      unsigned long long below = mask & ((1ull << lane) - 1);
      uint32_t start = below ? chunk + (63 - (uint32_t)__clzll(below)) + 1 : carry_start;
      uint32_t nb = pos - start + 1, idx = tok_base + (uint32_t)__popcll(below);
      uint64_t v = 0;
      if (nb <= 10)
        for (uint32_t k = 0; k < nb; k++) v |= (uint64_t)(L.bytes[start + k] & 0x7f) << (7 * k);
      L.tok[idx] = wv_pack_token(v, nb, b);
    }
    if (mask) carry_start = chunk + (63 - (uint32_t)__clzll(mask)) + 1;
    tok_base += (uint32_t)__popcll(mask);
  }
  __syncthreads();
  // Record walk. A record header names the next one (count > 1: two tokens on; count < 0: behind its -count values; count 0: two
  // on), so the headers are the orbit of token 0 under that map. Lane 0 walking it costs two or three dependent LDS reads per RECORD:
  // for the columns of a fat map change -- three hundred short records each, twelve columns -- that walk was nearly all of the 0.23 ms
  // such a change took. Columns of more than WV_SERIAL_TOKENS numbers are walked by the whole wavefront instead: every token computes
  // where it would point as a header, the orbit of token 0 is marked by pointer doubling (log2 rounds over the tokens, jumps read
  // before any is written), marked tokens are ranked by ballots and write their record. Same tables, same first error.
  // (a long column of FEW records -- a run of three hundred consecutive counters is one record of two numbers -- is walked by lane 0
  // faster than the doubling rounds take: lane 0 starts, and hands over when the column turns out to have more than WV_SERIAL_RECORDS)
  const bool may_hand_over = tok_base > WV_SERIAL_TOKENS && carry_start == len;
  if (lane == 0) {
    uint32_t e = carry_start != len ? (uint32_t)F_BAD_LEB : 0;  // buffer ended with incomplete number
    uint32_t t = 0, nr = 0, prev = 0;
    uint64_t rows = 0;
    L.n_runs = 0;
    while (t < tok_base && !e) {
      if (may_hand_over && nr >= WV_SERIAL_RECORDS) { L.n_runs = NONE32; break; }
      int64_t cnt;
      if (!wv_tok_sint(L, t, cnt)) { e = F_BAD_LEB; break; }
      uint32_t kind, first = t + 1;
      uint64_t count;
      if (cnt > 1) {
        if (t + 1 >= tok_base) { e = F_BAD_LEB; break; }
        kind = RK_REP; count = (uint64_t)cnt; t += 2;
      } else if (cnt == 1) {
        e = F_BAD_RLE; break;  // repetition count of 1
      } else if (cnt < 0) {
        if (prev == RK_LIT) { e = F_BAD_RLE; break; }  // successive literals
        count = (uint64_t)(-cnt);
        if (count > (uint64_t)(tok_base - t - 1)) { e = F_BAD_LEB; break; }
        kind = RK_LIT; t += 1 + (uint32_t)count;
      } else {
        uint64_t z;
        if (prev == RK_NUL) { e = F_BAD_RLE; break; }  // successive null runs
        if (t + 1 >= tok_base || !wv_tok_uint(L, t + 1, z)) { e = F_BAD_LEB; break; }
        if (z == 0) { e = F_BAD_RLE; break; }
        kind = RK_NUL; count = z; t += 2;
      }
      L.run_start[nr] = (uint32_t)rows;
      L.run_kind[nr] = (uint8_t)kind;
      L.run_tok[nr] = first;
      nr++;
      rows += count;
      if (rows > 0xfffffff0ull) { e = F_OVERFLOW; break; }
      prev = kind;
    }
    if (L.n_runs != NONE32) {
      L.run_start[nr] = (uint32_t)rows;
      L.n_runs = nr;
      L.n_tokens = tok_base;
      L.total_rows = (uint32_t)rows;
      L.err = e;
    }
  }
  __syncthreads();
  if (L.n_runs == NONE32) wv_records_parallel(L, tok_base, lane);   // (uniform: every lane reads the same LDS word)
}

// value of row i of the loaded column: returns the token index holding it, or NONE32 for null. *prev receives the token of
// row i - 1 when row i must differ from it for the encoding to be legal (first row of a record after a non-null record, or
// inside a literal), else NONE32.
template <class WL>
__device__ __forceinline__ uint32_t wv_row_token(const WL& L, uint32_t i, uint32_t* prev) {
  *prev = NONE32;
  if (i >= L.total_rows) return NONE32;  // past the end every value is null (encoding.js:821)
  uint32_t lo = 0, hi = L.n_runs;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (L.run_start[mid] <= i) lo = mid; else hi = mid;
  }
  uint32_t kind = L.run_kind[lo], start = L.run_start[lo], off = i - start;
  if (kind == RK_NUL) return NONE32;
  uint32_t first = L.run_tok[lo];
  if (off > 0) {
    if (kind == RK_LIT) *prev = first + off - 1;
  } else if (lo > 0) {
    uint32_t pk = L.run_kind[lo - 1];
    if (pk == RK_REP) *prev = L.run_tok[lo - 1];
    else if (pk == RK_LIT) *prev = L.run_tok[lo - 1] + (start - L.run_start[lo - 1] - 1);
  }
  return kind == RK_REP ? first : first + off;
}

// decoded value of row i (uint or signed), with the reference's adjacency rule (no equal neighbours across record
// boundaries or inside literals: encoding.js:826-829, 868-872)
template <bool SIGNED, class WL>
__device__ __forceinline__ bool wv_row_value(const WL& L, uint32_t i, bool& is_null, int64_t& v, uint32_t& err) {
  uint32_t tp;
  uint32_t t = wv_row_token(L, i, &tp);
  is_null = t == NONE32;
  v = 0;
  if (is_null) return true;
  bool ok;
  if (SIGNED) ok = wv_tok_sint(L, t, v);
  else { uint64_t u; ok = wv_tok_uint(L, t, u); v = (int64_t)u; }
  if (!ok) { err |= F_BAD_LEB; return false; }
  if (tp != NONE32) {
    int64_t pv = 0;
    bool okp;
    if (SIGNED) okp = wv_tok_sint(L, tp, pv);
    else { uint64_t u = 0; okp = wv_tok_uint(L, tp, u); pv = (int64_t)u; }
    if (okp && pv == v) err |= F_BAD_RLE;  // (an invalid predecessor raises its own flag when its row is decoded)
  }
  return true;
}

__device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t x, uint32_t lane) {
  for (int d = 1; d < WAVE; d <<= 1) {
    int64_t y = __shfl_up(x, (unsigned)d);
    if (lane >= (uint32_t)d) x += y;
  }
  return x;
}

// key strings of one change (UTF-8 RLE column: string bytes are not LEB tokens, so lane 0 walks it -- one step per RECORD, one
// per value only inside literals -- filling the run table in batches that all lanes then expand). P: LdsBytes when the
// column is staged (the usual case), else global memory.
template <class WL, class P>
__device__ __forceinline__ uint32_t wv_key_column(WL& L, P keycol, uint32_t key_len, uint32_t col_abs, uint32_t n, uint32_t base, uint32_t lane, OpCols o) {
CurT<P> c(keycol, 0, key_len);
uint32_t err = 0;
  int state = 0;                 // 0 none, 1 repetition, 2 literal, 3 nulls (encoding.js:865-887)
  uint32_t prev_kind = RK_NUL, prev_off = 0, prev_len = 0;  // last run of the previous batch (lane 0 uses it)
  int64_t lit_left = 0;          // values still to read from the current literal
  uint32_t rows_done = 0;        // rows already expanded
  for (;;) {
    if (lane == 0) {
      uint32_t nr = 0, e = 0;
      uint64_t rows = rows_done;
      while (nr < WL::RUNMAX && rows < n && !e && (lit_left > 0 || c.off < c.len)) {
        uint32_t kind, off = 0, len = 0;
        uint64_t count = 1;
        // inside a literal, the common case -- a string shorter than 128 bytes that lies inside the column -- in a loop of its own: one
        // dependent byte read and a dozen instructions per string (the general step below costs several times that, and a map change
        // of three hundred literal keys spent most of its decode time in it)
        while (lit_left > 0 && nr < WL::RUNMAX && rows < n) {
          const uint32_t at = c.off;
          if (at >= c.len) break;
          const uint32_t b = c.byte_at(at);
          if (b >= 0x80 || b > c.len - at - 1) break;
          L.run_start[nr] = (uint32_t)(rows - rows_done);
          L.run_kind[nr] = (uint8_t)RK_REP;
          L.run_tok[nr] = col_abs + at + 1;
          L.aux[nr] = b;
          c.off = at + 1 + b;
          nr++;
          rows++;
          lit_left--;
        }
        if (!(nr < WL::RUNMAX && rows < n && (lit_left > 0 || c.off < c.len))) break;
        if (lit_left > 0) {
          uint64_t l;
          if (!read_uleb(c, l)) { e = F_BAD_LEB; break; }
          off = c.off; len = (uint32_t)l;
          if (!skip_bytes(c, l)) { e = F_BAD_LEB; break; }
          lit_left--;
          kind = RK_REP;
        } else {
          int64_t cnt;
          if (!read_sleb(c, cnt)) { e = F_BAD_LEB; break; }
          if (cnt > 1) {
            uint64_t l;
            if (!read_uleb(c, l)) { e = F_BAD_LEB; break; }
            off = c.off; len = (uint32_t)l;
            if (!skip_bytes(c, l)) { e = F_BAD_LEB; break; }
            state = 1;
            kind = RK_REP; count = (uint64_t)cnt;
          } else if (cnt == 1) { e = F_BAD_RLE; break; }
          else if (cnt < 0) {
            if (state == 2) { e = F_BAD_RLE; break; }
            state = 2; lit_left = -cnt;
            continue;
          } else {
            uint64_t z;
            if (state == 3) { e = F_BAD_RLE; break; }
            if (!read_uleb(c, z)) { e = F_BAD_LEB; break; }
            if (z == 0) { e = F_BAD_RLE; break; }
            state = 3;
            kind = RK_NUL; count = z;
          }
        }
        L.run_start[nr] = (uint32_t)(rows - rows_done);
        L.run_kind[nr] = (uint8_t)kind;
        L.run_tok[nr] = col_abs + off;
        L.aux[nr] = len;
        nr++;
        rows += count;
        if (rows > n) rows = n;
      }
      bool exhausted = !(lit_left > 0 || c.off < c.len);
      if ((exhausted || e) && rows < n) {  // past the end of the column every value is null
        if (nr == WL::RUNMAX) nr--, rows = rows_done + L.run_start[nr];  // (cannot happen: loop stops at RUNMAX only with data left)
        L.run_start[nr] = (uint32_t)(rows - rows_done);
        L.run_kind[nr] = RK_NUL;
        L.run_tok[nr] = 0;
        L.aux[nr] = 0;
        nr++;
        rows = n;
      }
      L.run_start[nr] = (uint32_t)(rows - rows_done);
      L.n_runs = nr;
      L.total_rows = (uint32_t)(rows - rows_done);
      L.err = e;
    }
    __syncthreads();
    err |= L.err;
    uint32_t batch = L.total_rows, nr = L.n_runs;
    // a value never equals its predecessor (neither inside a literal nor across records, encoding.js:868-872): checked here,
    // one lane per run, instead of byte by byte inside lane 0's walk
    for (uint32_t r = lane; r < nr; r += WAVE) {
      uint32_t k = L.run_kind[r], pk = r ? L.run_kind[r - 1] : prev_kind;
      if (k == RK_NUL || pk == RK_NUL) continue;
      uint32_t off = L.run_tok[r] - col_abs, len = L.aux[r];
      uint32_t poff = r ? L.run_tok[r - 1] - col_abs : prev_off, plen = r ? L.aux[r - 1] : prev_len;
      if (len == plen && bytes_equal(keycol, poff, off, len, key_len)) err |= F_BAD_RLE;
    }
    if (lane == 0 && nr) { prev_kind = L.run_kind[nr - 1]; prev_off = L.run_tok[nr - 1] - col_abs; prev_len = L.aux[nr - 1]; }
    for (uint32_t i = lane; i < batch; i += WAVE) {
      uint32_t lo = 0, hi = nr;
      while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (L.run_start[mid] <= i) lo = mid; else hi = mid;
      }
      bool nul = L.run_kind[lo] == RK_NUL;
      o.key_off[base + rows_done + i] = nul ? 0 : L.run_tok[lo];
      o.key_len[base + rows_done + i] = nul ? NONE32 : L.aux[lo];
    }
    rows_done += batch;
    __syncthreads();
    if (rows_done >= n || batch == 0) break;
  }
  return err;
}

// column groups of k_decode_wave (see there)
enum : uint32_t { DG_ROWS = 1 /* action, op ids, insert, object */, DG_KEY_ID = 2 /* key element id, value */, DG_KEY_STR = 4, DG_PRED = 8, DG_ALL = 15 };

// (the small class is held to 80 registers = six wavefronts per SIMD: the gate and the wavefront record walk had taken it to 83 = five,
// and the headline decode from 60 to 66 us)
template <class WL>
__global__ __launch_bounds__(WAVE) AM355_WAVES_PER_EU(WL::COLMAX <= 256 ? 6 : 1) void k_decode_wave(const uint8_t* __restrict__ arena, const ChangeMeta* __restrict__ metas,
                                                       const ChangePlan* __restrict__ plans, uint32_t n_plans, ActorXlate x, OpCols o,
                                                       uint32_t* __restrict__ flags, DecodeGate gate) {
  __shared__ WL L;
  wave_priority_high();
  uint32_t pi = blockIdx.x, lane = threadIdx.x;
  // A batch of few, fat changes (256 changes of 6 KB in the map workload) leaves three quarters of the SIMDs idle with one wavefront
  // per change, and a wavefront's time is the serial record walks of its twelve columns one after the other. The columns fall into
  // groups that do not read each other's rows: with gridDim.y = 4 a change is decoded by four wavefronts (on four SIMDs, each staging
  // the change's 6 KB for itself), one group each; gridDim.y = 1: all of them (the small class; sharded replays, whose early exit
  // for foreign changes needs the object columns in the wavefront that decides).
  const uint32_t groups = gridDim.y > 1 ? (1u << blockIdx.y) : (uint32_t)DG_ALL;
  if (gate.totals) {
    // launched behind k_plan_apply with one wavefront per CHANGE, before the host has seen the totals: the class size comes from
    // the device copy, and the launch does nothing at all when the host is going to decide otherwise (same test as decode_gate_open)
    const PlanTotals t = *gate.totals;
    if (!decode_gate_open(t, gate.cap_ops, gate.cap_preds, gate.cap_distinct)) return;
    n_plans = gate.large ? t.n_large : t.n_small;
    if (gate.large) plans += gate.n_changes - t.n_large;
  }
  if (pi >= n_plans) return;
  const ChangePlan pl = plans[pi];
  const ChangeMeta* m = &metas[pl.change];
  const uint8_t* p = arena + m->base;
  const uint32_t n = m->n_ops, base = pl.op_base, abs0 = (uint32_t)m->base, n_preds_cap = m->n_preds;
  const uint32_t start_op = (uint32_t)m->start_op;
  uint32_t col_off[C_NUM], col_len[C_NUM];
  for (int k = 0; k < C_NUM; k++) { col_off[k] = m->col_off[k]; col_len[k] = m->col_len[k]; }
  uint32_t err = 0;
  // change-local actor index -> global rank, staged once per change
  for (uint32_t k = lane; k < pl.n_actors; k += WAVE) {
    uint32_t v = x.amap[pl.amap_base + k];
    L.rank[k] = x.slot_rank ? x.slot_rank[v] : v;
  }
  // the columns of a change are contiguous: when they fit, stage them all with one round of wide loads so the
  // per-column steps below never wait on global memory again
  uint32_t reg_lo = 0xffffffffu, reg_hi = 0;
  for (int k = 0; k < C_NUM; k++)
    if (col_len[k]) {
      reg_lo = col_off[k] < reg_lo ? col_off[k] : reg_lo;
      reg_hi = col_off[k] + col_len[k] > reg_hi ? col_off[k] + col_len[k] : reg_hi;
    }
  const bool staged = reg_hi > reg_lo && reg_hi - reg_lo <= WL::REGION;
  if (staged) stage_to_lds(L.region, p + reg_lo, reg_hi - reg_lo, lane);
  // columns are read through LDS-typed pointers when staged (ds_read), through global pointers otherwise; a pointer that
  // could be either would make every access a FLAT load
  auto load_col = [&](int k) {
    if (staged && col_len[k]) wv_load_column(L, (LdsBytes)(L.region + (col_off[k] - reg_lo)), col_len[k], lane);
    else wv_load_column(L, p + col_off[k], col_len[k], lane);
  };

  if (groups & DG_ROWS) {
  // ---- action (+ op ids: op i of a change is (startOp + i, author), new.js:708-709) ----
  load_col(C_ACTION);
  err |= L.err;
  for (uint32_t i = lane; i < n; i += WAVE) {
    bool nul; int64_t v;
    wv_row_value<false>(L, i, nul, v, err);
    if (nul) err |= F_UNSUPPORTED;
    if ((uint64_t)v >= NONE32) err |= F_OVERFLOW;
    o.action[base + i] = (uint32_t)v;
    o.id_ctr[base + i] = start_op + i;
    o.id_actor[base + i] = pl.author;
  }
  // ---- insert (boolean: alternating run lengths, first run counts false; encoding.js:1171-1183) ----
  {
    __syncthreads();
    uint32_t len = col_len[C_INSERT];
    if (staged && len) {
      LdsBytes col = (LdsBytes)(L.region + (col_off[C_INSERT] - reg_lo));
      for (uint32_t i = lane; i < len; i += WAVE) L.bytes[i] = col[i];
    } else {
      const uint8_t* col = p + col_off[C_INSERT];
      for (uint32_t i = lane; i < len; i += WAVE) L.bytes[i] = col[i];
    }
    __syncthreads();
    // tokens = run lengths
    uint32_t tok_base = 0, carry_start = 0;
    for (uint32_t chunk = 0; chunk < len; chunk += WAVE) {
      uint32_t pos = chunk + lane;
      bool in = pos < len;
      uint32_t b = in ? L.bytes[pos] : 0x80;
      bool term = in && !(b & 0x80);
      unsigned long long mask = __ballot(term);
      if (term) {
        unsigned long long below = mask & ((1ull << lane) - 1);
        uint32_t start = below ? chunk + (63 - (uint32_t)__clzll(below)) + 1 : carry_start;
        uint32_t nb = pos - start + 1, idx = tok_base + (uint32_t)__popcll(below);
        uint64_t v = 0;
        if (nb <= 10)
          for (uint32_t k = 0; k < nb; k++) v |= (uint64_t)(L.bytes[start + k] & 0x7f) << (7 * k);
        L.tok[idx] = wv_pack_token(v, nb, b);
      }
      if (mask) carry_start = chunk + (63 - (uint32_t)__clzll(mask)) + 1;
      tok_base += (uint32_t)__popcll(mask);
    }
    __syncthreads();
    if (lane == 0) {
      uint32_t e = carry_start != len ? (uint32_t)F_BAD_LEB : 0;
      uint64_t rows = 0;
      for (uint32_t t = 0; t < tok_base && !e; t++) {
        uint64_t c;
        if (!wv_tok_uint(L, t, c)) { e = F_BAD_LEB; break; }
        if (c == 0 && t > 0) { e = F_BAD_RLE; break; }  // zero-length runs are only legal as the first (false) run
        L.run_start[t] = (uint32_t)rows;
        rows += c;
        if (rows > 0xfffffff0ull) { e = F_OVERFLOW; break; }
      }
      L.run_start[tok_base] = (uint32_t)rows;
      L.n_runs = tok_base;
      L.total_rows = (uint32_t)rows;
      L.err = e;
    }
    __syncthreads();
    err |= L.err;
    for (uint32_t i = lane; i < n; i += WAVE) {
      uint32_t val = 0;
      if (i < L.total_rows) {
        uint32_t lo = 0, hi = L.n_runs;
        while (hi - lo > 1) {
          uint32_t mid = (lo + hi) >> 1;
          if (L.run_start[mid] <= i) lo = mid; else hi = mid;
        }
        // several runs may start at the same row when a zero-length first run exists: take the last one covering i
        val = lo & 1;  // run 0 = false, run 1 = true, ...
      }
      o.insert[base + i] = (uint8_t)val;
    }
  }
  // ---- object id ----
  load_col(C_OBJ_ACTOR);
  err |= L.err;
  for (uint32_t i = lane; i < n; i += WAVE) {
    bool nul; int64_t v;
    wv_row_value<false>(L, i, nul, v, err);
    uint32_t r = NONE32;
    if (!nul) { if ((uint64_t)v >= pl.n_actors) err |= F_BAD_ROW; else r = L.rank[v]; }
    o.obj_actor[base + i] = r;
  }
  load_col(C_OBJ_CTR);
  err |= L.err;
  for (uint32_t i = lane; i < n; i += WAVE) {
    bool nul; int64_t v;
    wv_row_value<false>(L, i, nul, v, err);
    if (nul != (o.obj_actor[base + i] == NONE32) && !(err & F_BAD_ROW)) err |= F_BAD_ROW;  // new.js:715-718
    if (!nul && (uint64_t)v >= NONE32) err |= F_OVERFLOW;
    o.obj_ctr[base + i] = nul ? 0 : (uint32_t)v;
  }
  }  // DG_ROWS
  // ---- objectId sharding: a change none of whose rows belongs to an object of this rank is decoded as far as the merge
  //      stage looks at foreign rows -- action, id, insert flag, object (k_resolve marks them K_FOREIGN from those and goes on; make
  //      rows still enter the object table) -- and no further: keys, values and pred lists of such a change are its owners' business,
  //      as is finding fault with them (a rank that rejects the batch makes every rank raise) ----
  if (x.shard_world > 1) {
    bool mine = false;
    for (uint32_t i = lane; i < n; i += WAVE) mine = mine || shard_owner(o.obj_actor[base + i], o.obj_ctr[base + i], x.shard_world) == x.shard_rank;
    if (!__ballot(mine)) {
      if (err) atomicOr(flags, err);
      return;
    }
  }
  if (groups & DG_KEY_ID) {
  // ---- key: element id (actor, delta-coded counter) ----
  load_col(C_KEY_ACTOR);
  err |= L.err;
  for (uint32_t i = lane; i < n; i += WAVE) {
    bool nul; int64_t v;
    wv_row_value<false>(L, i, nul, v, err);
    uint32_t r = NONE32;
    if (!nul) { if ((uint64_t)v >= pl.n_actors) err |= F_BAD_ROW; else r = L.rank[v]; }
    o.key_actor[base + i] = r;
  }
  load_col(C_KEY_CTR);
  err |= L.err;
  {
    int64_t carry = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += WAVE) {
      uint32_t i = i0 + lane;
      bool nul = true; int64_t d = 0;
      if (i < n) wv_row_value<true>(L, i, nul, d, err);
      int64_t abs = carry + wave_incl_scan_i64(nul ? 0 : d, lane);  // running value; nulls do not advance it
      carry = __shfl(abs, WAVE - 1);
      if (i < n) {
        bool a_null = o.key_actor[base + i] == NONE32;
        if (!nul && (abs > (int64_t)MAX_SAFE || abs < -(int64_t)MAX_SAFE)) err |= F_BAD_LEB;
        // new.js:719-723
        if ((nul && !a_null) || (!nul && abs == 0 && !a_null) || (!nul && abs > 0 && a_null)) err |= F_BAD_ROW;
        if (!nul && (abs < 0 || (uint64_t)abs >= NONE32)) err |= F_OVERFLOW;
        o.key_ctr[base + i] = nul ? NONE32 : (uint32_t)abs;
      }
    }
  }
  }  // DG_KEY_ID (continued below: the value column)
  // ---- key: string ----
  if (groups & DG_KEY_STR) {
    __syncthreads();
    // lane 0 compares neighbouring strings byte by byte: make sure it does so in LDS. If the whole column region did
    // not fit, the key column alone usually does (`region` is unused in that case: the other columns stage through `bytes`).
    const uint32_t col_abs = abs0 + col_off[C_KEY_STR];
    uint32_t key_at = staged ? col_off[C_KEY_STR] - reg_lo : 0;
    bool key_lds = staged && col_len[C_KEY_STR];
    if (!staged && col_len[C_KEY_STR] && col_len[C_KEY_STR] <= WL::REGION) {
      stage_to_lds(L.region, p + col_off[C_KEY_STR], col_len[C_KEY_STR], lane);
      key_lds = true;
      __syncthreads();
    }
    if (key_lds) err |= wv_key_column(L, (LdsBytes)(L.region + key_at), col_len[C_KEY_STR], col_abs, n, base, lane, o);
    else err |= wv_key_column(L, p + col_off[C_KEY_STR], col_len[C_KEY_STR], col_abs, n, base, lane, o);
  }
  if (groups & DG_KEY_ID) {
  // ---- value: (len << 4 | tag) per row, offsets into valRaw are an exclusive prefix sum of the lengths ----
  load_col(C_VAL_LEN);
  err |= L.err;
  {
    int64_t carry = 0;
    uint32_t raw_abs = abs0 + col_off[C_VAL_RAW], raw_len = col_len[C_VAL_RAW];
    for (uint32_t i0 = 0; i0 < n; i0 += WAVE) {
      uint32_t i = i0 + lane;
      bool nul = true; int64_t tl = 0;
      if (i < n) wv_row_value<false>(L, i, nul, tl, err);
      if (nul) tl = 0;
      if ((uint64_t)tl >= NONE32) { err |= F_OVERFLOW; tl = 0; }
      int64_t incl = carry + wave_incl_scan_i64(tl >> 4, lane);
      carry = __shfl(incl, WAVE - 1);
      if (i < n) {
        if ((uint64_t)incl > raw_len) err |= F_BAD_CHUNK;  // readRawBytes past the end of valRaw
        o.val_tl[base + i] = (uint32_t)tl;
        o.val_off[base + i] = raw_abs + (uint32_t)(incl - (tl >> 4));
      }
    }
  }
  }  // DG_KEY_ID
  if (groups & DG_PRED) {
  // ---- preds: group cardinality, then the two value columns consumed predNum[i] entries per row ----
  uint32_t total_preds = 0;
  load_col(C_PRED_NUM);
  err |= L.err;
  {
    int64_t carry = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += WAVE) {
      uint32_t i = i0 + lane;
      bool nul = true; int64_t k = 0;
      if (i < n) wv_row_value<false>(L, i, nul, k, err);
      if (nul) k = 0;
      int64_t incl = carry + wave_incl_scan_i64(k, lane);
      carry = __shfl(incl, WAVE - 1);
      if (i < n) {
        if ((uint64_t)incl > n_preds_cap) { err |= F_BAD_RLE; k = 0; incl = 0; }
        o.pred_num[base + i] = (uint32_t)k;
        o.pred_first[base + i] = pl.pred_base + (uint32_t)(incl - k);
      }
    }
    total_preds = (uint64_t)carry > n_preds_cap ? 0 : (uint32_t)carry;
  }
  load_col(C_PRED_ACTOR);
  err |= L.err;
  for (uint32_t j = lane; j < total_preds; j += WAVE) {
    bool nul; int64_t v;
    wv_row_value<false>(L, j, nul, v, err);
    uint32_t r = 0;
    if (nul) err |= F_UNSUPPORTED;
    else if ((uint64_t)v >= pl.n_actors) err |= F_BAD_ROW;
    else r = L.rank[v];
    o.pred_actor[pl.pred_base + j] = r;
  }
  load_col(C_PRED_CTR);
  err |= L.err;
  {
    int64_t carry = 0;
    for (uint32_t j0 = 0; j0 < total_preds; j0 += WAVE) {
      uint32_t j = j0 + lane;
      bool nul = true; int64_t d = 0;
      if (j < total_preds) wv_row_value<true>(L, j, nul, d, err);
      int64_t abs = carry + wave_incl_scan_i64(nul ? 0 : d, lane);
      carry = __shfl(abs, WAVE - 1);
      if (j < total_preds) {
        if (nul) err |= F_UNSUPPORTED;
        else if (abs < 0 || (uint64_t)abs >= NONE32) err |= F_OVERFLOW;
        o.pred_ctr[pl.pred_base + j] = (uint32_t)abs;
      }
    }
  }
  }  // DG_PRED
  if (err) atomicOr(flags, err);
}

int change_wave_class(const ChangeMeta& m) { return wave_class_of(m); }

void launch_parse_changes(const uint8_t* arena, const uint64_t* offsets, uint32_t n_changes, ChangeMeta* metas, uint32_t* n_entries, const ParseFills& fills,
                          hipStream_t st, bool fat) {
  if (!n_changes && !fills.n) return;
  if (fat) hipLaunchKernelGGL(k_parse_changes<true>, dim3(n_changes ? n_changes : 64), dim3(WAVE), 0, st, arena, offsets, n_changes, metas, n_entries, fills);
  else hipLaunchKernelGGL(k_parse_changes<false>, dim3(n_changes ? n_changes : 64), dim3(WAVE), 0, st, arena, offsets, n_changes, metas, n_entries, fills);
}

void launch_hash_changes(const uint8_t* arena, const uint64_t* offsets, uint32_t n, uint8_t* hashes, uint32_t* min_idx, uint32_t* hash_tab,
                         uint32_t tab_mask, uint32_t* flags, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_hash_changes, dim3((n + HASH_LANES - 1) / HASH_LANES), dim3(2 * HASH_LANES), 0, st, arena, offsets, n, hashes, min_idx, flags);
  AM355_LAUNCH_INDEPENDENT(k_hash_insert, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), st, (const uint8_t*)hashes, n, hash_tab, tab_mask, min_idx);
}

void launch_deps_resolve(const uint8_t* arena, const ChangeMeta* metas, const uint8_t* hashes, uint32_t n, const uint32_t* hash_tab, uint32_t tab_mask,
                         const uint32_t* min_idx, uint8_t* has_dependent, uint32_t* fast_flags, uint32_t* dep_idx, uint32_t* self_idx, hipStream_t st) {
  if (!n) return;
  AM355_LAUNCH_INDEPENDENT(k_deps_resolve, dim3(n), dim3(WAVE), st, arena, metas, hashes, n, hash_tab, tab_mask, min_idx, has_dependent, fast_flags, dep_idx,
                           self_idx);
}

void launch_actor_intern(const uint8_t* arena, ChangeMeta* metas, uint32_t n, const uint32_t* amap_base, uint32_t* amap, uint32_t amap_cap,
                         unsigned long long* slots, uint32_t slot_mask, uint32_t* first_idx, uint32_t* flags, uint32_t* fast_flags, uint32_t* distinct,
                         void* rank_ids, ChangeBrief* briefs, uint32_t* slot_rank, unsigned long long* block_sums, uint32_t* plan_words, hipStream_t st,
                         ChangeBrief* host_briefs) {
  if (n)
    hipLaunchKernelGGL(k_actor_intern, dim3(n), dim3(WAVE), 0, st, arena, metas, n, amap_base, amap, amap_cap, slots, slot_mask, first_idx, flags,
                       fast_flags, distinct, (RankId*)rank_ids);
  if (n) hipLaunchKernelGGL(k_actor_first, dim3(n), dim3(WAVE), 0, st, metas, n, amap_base, (const uint32_t*)amap, amap_cap, (const uint32_t*)first_idx, fast_flags);
  hipLaunchKernelGGL(k_actor_check, dim3((n + BLOCK - 1) / BLOCK + 1), dim3(BLOCK), 0, st, arena, metas, n, amap_base, (const uint32_t*)amap, amap_cap,
                     (const uint32_t*)first_idx, flags, fast_flags, briefs, (const uint32_t*)distinct, (const RankId*)rank_ids, slot_rank, block_sums, plan_words,
                     host_briefs);
}

size_t rank_ids_bytes() { return sizeof(RankId) * PLAN_RANK_MAX; }

size_t plan_block_sums_bytes(uint32_t n) { return sizeof(unsigned long long) * PLAN_SUMS * ((size_t)(n + BLOCK - 1) / BLOCK + 1); }

void launch_plan(const ChangeBrief* briefs, uint32_t n, const uint32_t* distinct, const uint32_t* slot_rank, uint32_t slot_mask, const unsigned long long* block_sums,
                 ChangePlan* plans, ChangePlan* plans_serial, const uint32_t* words, const uint32_t* plan_words, HostSignals* sig, uint32_t seq, hipStream_t st,
                 PlanTotals* dev_totals, uint32_t* host_s1) {
  uint32_t nb = (n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_plan_apply, dim3(nb ? nb : 1), dim3(BLOCK), 0, st, briefs, n, slot_rank, slot_mask, block_sums, plans, plans_serial, words, plan_words,
                     distinct, sig, seq, dev_totals, host_s1);
}

uint32_t distinct_capacity() { return DISTINCT_CAP; }

// rows and succ entries of a document's op columns (lane 0: values in `action`; lane 1: sum of `succNum`)
__global__ __launch_bounds__(WAVE) void k_doc_count(const uint8_t* __restrict__ arena, ChangeMeta* __restrict__ meta) {
  uint32_t lane = threadIdx.x;
  if (lane > 1) return;
  int slot = lane == 0 ? C_ACTION : C_PRED_NUM;
  uint64_t cnt, sum;
  bool ok = rle_count_sum(arena + meta->base + meta->col_off[slot], meta->col_len[slot], cnt, sum);
  if (!ok) { atomicOr(&meta->flags, (uint32_t)F_BAD_RLE); return; }
  if (lane == 0) meta->n_ops = (uint32_t)cnt; else meta->n_preds = (uint32_t)sum;
}

void launch_doc_count(const uint8_t* arena, ChangeMeta* meta, hipStream_t st) {
  AM355_LAUNCH_INDEPENDENT(k_doc_count, dim3(1), dim3(WAVE), st, arena, meta);
}

// key_off / key_len of every row from the run table (rows past the end of the column are null, encoding.js:821)
__global__ __launch_bounds__(BLOCK) void k_keystr_expand(const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ run_off,
                                                          const uint32_t* __restrict__ run_len, const uint32_t* __restrict__ n_runs_p, uint32_t n_rows,
                                                          uint32_t* __restrict__ key_off, uint32_t* __restrict__ key_len) {
  uint32_t row = gtid();
  if (row >= n_rows) return;
  uint32_t nr = *n_runs_p;
  uint32_t off = 0, len = NONE32;
  if (nr && row < run_start[nr]) {
    uint32_t lo = 0, hi = nr;
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (run_start[mid] <= row) lo = mid; else hi = mid;
    }
    off = run_off[lo];
    len = run_len[lo];
  }
  key_off[row] = off;
  key_len[row] = len;
}

void launch_keystr_expand(const uint32_t* run_start, const uint32_t* run_off, const uint32_t* run_len, const uint32_t* n_runs, uint32_t n_rows,
                          uint32_t* key_off, uint32_t* key_len, hipStream_t st) {
  if (!n_rows) return;
  AM355_LAUNCH_INDEPENDENT(k_keystr_expand, dim3((n_rows + BLOCK - 1) / BLOCK), dim3(BLOCK), st, run_start, run_off, run_len, n_runs, n_rows, key_off, key_len);
}

// Document op columns -> rows. First version: the lane-serial column decoders, one lane per column group (document
// columns run to megabytes, far beyond the wave decoder's LDS staging; a parallel big-column decoder is the next step).
void launch_decode_document(const uint8_t* arena, const ChangeMeta* meta, const ChangePlan* plan, const uint32_t* actor_rank, OpCols cols,
                            uint32_t* flags, hipStream_t st) {
  ActorXlate x{actor_rank, nullptr, 0, 1};
  AM355_LAUNCH_INDEPENDENT(k_decode_columns, dim3(1, T_NUM_DOC), dim3(WAVE), st, arena, meta, plan, 1u, x, cols, flags, 0);
}

// Four wavefronts per change (one per column group) while all of them are resident at once: the small class holds six wavefronts per
// SIMD -- 6144 on the device, 1536 changes --, the large class (24 KB of LDS per wavefront) a quarter of that. Measured (round 5, second
// session, profiles/r05_s2_ab_decode_split.txt): `c2_text_typing` (1001 changes) T_device 0.255 -> 0.237 ms with the split; the headline's
// 4097 changes decode 9 us faster with it but the replay is not (0.383 -> 0.390: sixteen thousand wavefronts take the SIMDs from the
// fills beside them): rounds 3-5 split up to 512 changes. (AM355_DECODE_SPLIT_MAX: the small class's bound, for A/B runs.)
static uint32_t decode_group_split(uint32_t n_waves, uint32_t shard_world, bool large_class = false) {
  static const uint32_t split_max = []() { const char* e = getenv("AM355_DECODE_SPLIT_MAX"); return e && atol(e) > 0 ? (uint32_t)atol(e) : 1536u; }();
  return (shard_world <= 1 && n_waves <= (large_class ? 512u : split_max)) ? 4u : 1u;
}

void launch_decode_columns(const uint8_t* arena, const ChangeMeta* metas, const ChangePlan* plans, uint32_t n_small, uint32_t n_large, uint32_t n_serial,
                           const uint32_t* amap, const uint32_t* slot_rank, OpCols cols, uint32_t* flags, hipStream_t st, hipStream_t aux, uint32_t shard_rank,
                           uint32_t shard_world) {
  // plans = [small wave class | large wave class | lane-serial]: the first two go to the wave-per-change run-level decoder
  // (two LDS footprints, see WaveLdsT), the rest (a column too long for LDS staging) to the lane-serial decoder.
  // Every launch is bound by the latency of one change, not by throughput, so the classes run side by side: the large
  // class on the auxiliary stream, which the caller has forked from `st` and joins afterwards. With few changes the LDS
  // footprint does not limit residency: one launch.
  ActorXlate x{amap, slot_rank, shard_rank, shard_world};
  if (n_large && n_small + n_large <= 1024) { n_large += n_small; n_small = 0; }
  hipStream_t s2 = (n_small && aux) ? aux : st;
  if (n_small) hipLaunchKernelGGL(k_decode_wave<WaveLdsSmall>, dim3(n_small, decode_group_split(n_small, shard_world)), dim3(WAVE), 0, st, arena, metas, plans, n_small, x, cols, flags, DecodeGate{});
  if (n_large) hipLaunchKernelGGL(k_decode_wave<WaveLdsLarge>, dim3(n_large, decode_group_split(n_large, shard_world, true)), dim3(WAVE), 0, s2, arena, metas, plans + n_small, n_large, x, cols, flags, DecodeGate{});
  if (n_serial)
    AM355_LAUNCH_INDEPENDENT(k_decode_columns, dim3((n_serial + WAVE - 1) / WAVE, T_NUM), dim3(WAVE), s2, arena, metas, plans + n_small + n_large, n_serial,
                             x, cols, flags, 0);
}

// the same three launches from the device-built plans (k_plan): small class at the front of `plans`, large class at its back
void launch_decode_planned(const uint8_t* arena, const ChangeMeta* metas, const ChangePlan* plans, const ChangePlan* plans_serial, uint32_t n_changes,
                           uint32_t n_small, uint32_t n_large, uint32_t n_serial, const uint32_t* amap, const uint32_t* slot_rank, OpCols cols, uint32_t* flags,
                           hipStream_t st, hipStream_t aux, uint32_t shard_rank, uint32_t shard_world) {
  ActorXlate x{amap, slot_rank, shard_rank, shard_world};
  hipStream_t s2 = (n_small && aux) ? aux : st;
  if (n_small) hipLaunchKernelGGL(k_decode_wave<WaveLdsSmall>, dim3(n_small, decode_group_split(n_small, shard_world)), dim3(WAVE), 0, st, arena, metas, plans, n_small, x, cols, flags, DecodeGate{});
  if (n_large) hipLaunchKernelGGL(k_decode_wave<WaveLdsLarge>, dim3(n_large, decode_group_split(n_large, shard_world, true)), dim3(WAVE), 0, s2, arena, metas, plans + (n_changes - n_large), n_large, x, cols, flags, DecodeGate{});
  if (n_serial)
    AM355_LAUNCH_INDEPENDENT(k_decode_columns, dim3((n_serial + WAVE - 1) / WAVE, T_NUM), dim3(WAVE), s2, arena, metas, plans_serial, n_serial, x, cols, flags, 0);
}

// Four wavefronts per change (one per column group, k_decode_wave) when the launch would otherwise leave SIMDs idle: up to 512
// changes -- 2048 wavefronts on 1024 SIMDs. A batch of thousands of changes fills the device with one wavefront per change, and four
// would only stage every change four times. (Sharded replays keep one: the early exit for foreign changes needs the object columns.)
// The two wavefront-per-change decoder classes enqueued BEFORE the totals are known (behind k_plan_apply, which leaves them in
// `totals`): one wavefront per change each, the small class on `st`, the large class on `aux`; rows go to `cols`, carved for
// cap_ops / cap_preds. Both launches do nothing unless decode_gate_open() holds, which the host evaluates on the same totals.
void launch_decode_speculative(const uint8_t* arena, const ChangeMeta* metas, const ChangePlan* plans, uint32_t n_changes, const PlanTotals* totals, uint32_t cap_ops,
                               uint32_t cap_preds, uint32_t cap_distinct, const uint32_t* amap, const uint32_t* slot_rank, OpCols cols, uint32_t* flags, hipStream_t st,
                               hipStream_t aux, bool with_large, uint32_t shard_rank, uint32_t shard_world) {
  if (!n_changes) return;
  ActorXlate x{amap, slot_rank, shard_rank, shard_world};
  hipLaunchKernelGGL(k_decode_wave<WaveLdsSmall>, dim3(n_changes, decode_group_split(n_changes, shard_world)), dim3(WAVE), 0, st, arena, metas, plans, 0u, x, cols, flags,
                     DecodeGate{totals, cap_ops, cap_preds, cap_distinct, n_changes, 0u});
  // (!with_large: the caller expects no change of the large class -- none in the context's previous batch -- and launches that class
  // itself should there be one after all)
  if (with_large)
    hipLaunchKernelGGL(k_decode_wave<WaveLdsLarge>, dim3(n_changes, decode_group_split(n_changes, shard_world, true)), dim3(WAVE), 0, aux, arena, metas, plans, 0u, x, cols, flags,
                       DecodeGate{totals, cap_ops, cap_preds, cap_distinct, n_changes, 1u});
}

}  // namespace am355
