// Host-side byte utilities (see am355_host.h). Plain C++: compiled for the host only.
#include "am355_host.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>

#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#endif

namespace am355 {
namespace {

const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

void blocks_scalar(uint32_t h[8], const uint8_t* p, size_t nblocks) {
  for (; nblocks; nblocks--, p += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
      uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
}

#if defined(__x86_64__)
// x86 SHA extensions: sha256rnds2 does two rounds on the (ABEF, CDGH) state halves, sha256msg1 / msg2 the message schedule;
// group g of four rounds consumes W[4g..4g+3] and produces W[4g+16..4g+19] in its place.
__attribute__((target("sha,sse4.1,ssse3"))) void blocks_shani(uint32_t h[8], const uint8_t* p, size_t nblocks) {
  const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
  __m128i tmp = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&h[0]), 0xB1);  // CDAB
  __m128i s1 = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&h[4]), 0x1B);   // EFGH
  __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                                         // ABEF
  s1 = _mm_blend_epi16(s1, tmp, 0xF0);                                              // CDGH
  for (; nblocks; nblocks--, p += 64) {
    const __m128i save0 = s0, save1 = s1;
    __m128i m[4];
    for (int k = 0; k < 4; k++) m[k] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * k)), bswap);
#pragma GCC unroll 16
    for (int g = 0; g < 16; g++) {
      __m128i msg = _mm_add_epi32(m[g & 3], _mm_loadu_si128((const __m128i*)&K256[4 * g]));
      s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
      msg = _mm_shuffle_epi32(msg, 0x0E);
      s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
      if (g < 12) {
        __m128i t = _mm_sha256msg1_epu32(m[g & 3], m[(g + 1) & 3]);
        t = _mm_add_epi32(t, _mm_alignr_epi8(m[(g + 3) & 3], m[(g + 2) & 3], 4));
        m[g & 3] = _mm_sha256msg2_epu32(t, m[(g + 3) & 3]);
      }
    }
    s0 = _mm_add_epi32(s0, save0);
    s1 = _mm_add_epi32(s1, save1);
  }
  tmp = _mm_shuffle_epi32(s0, 0x1B);        // FEBA
  s1 = _mm_shuffle_epi32(s1, 0xB1);         // DCHG
  s0 = _mm_blend_epi16(tmp, s1, 0xF0);      // DCBA
  s1 = _mm_alignr_epi8(s1, tmp, 8);         // HGFE
  _mm_storeu_si128((__m128i*)&h[0], s0);
  _mm_storeu_si128((__m128i*)&h[4], s1);
}

bool cpu_has_sha() {
  static const bool has = []() {
    unsigned a, b, c, d;
    if (!__get_cpuid(1, &a, &b, &c, &d) || !(c & (1u << 9)) || !(c & (1u << 19))) return false;  // SSSE3, SSE4.1
    if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
    return (b & (1u << 29)) != 0;  // SHA
  }();
  return has;
}
#endif

void blocks(uint32_t h[8], const uint8_t* p, size_t nblocks) {
#if defined(__x86_64__)
  if (cpu_has_sha()) { blocks_shani(h, p, nblocks); return; }
#endif
  blocks_scalar(h, p, nblocks);
}

}  // namespace

void sha256_initial(uint32_t h[8]) {
  static const uint32_t init[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(h, init, sizeof init);
}
void sha256_blocks(uint32_t h[8], const uint8_t* p, size_t nblocks) { blocks(h, p, nblocks); }
void sha256_finish(uint32_t h[8], const uint8_t* tail_bytes, size_t rem, uint64_t total_len, uint8_t out[32]) {
  uint8_t tail[128] = {0};
  size_t tl = rem < 56 ? 64 : 128;
  if (rem) memcpy(tail, tail_bytes, rem);
  tail[rem] = 0x80;
  uint64_t bits = total_len * 8;
  for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
  blocks(h, tail, tl / 64);
  for (int k = 0; k < 8; k++) { out[4 * k] = h[k] >> 24; out[4 * k + 1] = h[k] >> 16; out[4 * k + 2] = h[k] >> 8; out[4 * k + 3] = h[k]; }
}
void sha256_digest(const uint8_t* p, size_t len, uint8_t out[32]) {
  uint32_t h[8];
  sha256_initial(h);
  size_t full = len / 64;
  blocks(h, p, full);
  sha256_finish(h, p + 64 * full, len - 64 * full, len, out);
}

// (one z_stream per host thread, re-armed with inflateReset: inflateInit2 allocates and clears ~40 KB of state, which for a batch of
// thousands of 3 KB changes cost three times the decoding itself)
namespace {
struct ThreadInflater {
  z_stream zs;
  bool live = false;
  ~ThreadInflater() { if (live) inflateEnd(&zs); }
  bool arm() {
    if (live) return inflateReset2(&zs, -15) == Z_OK;
    memset(&zs, 0, sizeof zs);
    live = inflateInit2(&zs, -15) == Z_OK;
    return live;
  }
};
}  // namespace

// libdeflate (whole-buffer DEFLATE, about twice zlib's inflate rate on the document columns) when the system has it: looked up
// once with dlopen -- the image ships libdeflate.so.0 without headers --, zlib otherwise and for everything libdeflate does not
// settle: a stream it calls bad is decided by zlib (pako is a port of zlib: its verdict is the reference's), and one whose size
// three growing guesses do not cover is inflated by zlib's streaming loop below.
namespace {
struct LibDeflate {
  void* (*alloc)() = nullptr;
  int (*decompress_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;
  void (*release)(void*) = nullptr;
  LibDeflate() {
    if (getenv("AM355_NO_LIBDEFLATE")) return;
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
    decompress_ex = (int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*))dlsym(h, "libdeflate_deflate_decompress_ex");
    release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
    if (!alloc || !decompress_ex || !release) alloc = nullptr;
  }
};
const LibDeflate& libdeflate() {
  static const LibDeflate l;
  return l;
}
struct ThreadDecompressor {
  void* d = nullptr;
  ~ThreadDecompressor() { if (d) libdeflate().release(d); }
};
}  // namespace

int inflate_raw(const uint8_t* in, size_t in_len, std::vector<uint8_t>& out, size_t cap) {
  if (libdeflate().alloc) {
    static thread_local ThreadDecompressor td;
    if (!td.d) td.d = libdeflate().alloc();
    if (td.d) {
      size_t guess = std::min<size_t>(in_len * 4 + 4096, cap);
      for (int attempt = 0; attempt < 3; attempt++) {
        out.resize(guess);
        size_t used = 0, made = 0;
        int rc = libdeflate().decompress_ex(td.d, in, in_len, out.data(), out.size(), &used, &made);
        if (rc == 0) { out.resize(made); return 0; }
        if (rc != 3) break;         // LIBDEFLATE_BAD_DATA: zlib decides below
        if (guess >= cap) break;    // LIBDEFLATE_INSUFFICIENT_SPACE at the limit: zlib reports the overflow
        guess = std::min<size_t>(guess * 4, cap);
      }
    }
  }
  static thread_local ThreadInflater ti;
  if (!ti.arm()) return 3;
  z_stream& zs = ti.zs;
  zs.next_in = nullptr;  // (inflateReset keeps the caller's buffer fields: a previous stream may have ended with input left over)
  zs.avail_in = 0;
  zs.next_out = nullptr;
  zs.avail_out = 0;
  out.resize(std::min<size_t>(std::max<size_t>(in_len * 4, 1024), cap));
  size_t in_off = 0, produced = 0;
  int result = 1;
  for (;;) {
    if (zs.avail_in == 0 && in_off < in_len) {
      size_t take = std::min<size_t>(in_len - in_off, 1u << 30);
      zs.next_in = (Bytef*)(in + in_off);
      zs.avail_in = (uInt)take;
      in_off += take;
    }
    if (produced == out.size()) {
      if (out.size() >= cap) { result = 2; break; }
      out.resize(std::min<size_t>(out.size() * 2, cap));
    }
    size_t room = std::min<size_t>(out.size() - produced, 1u << 30);
    zs.next_out = out.data() + produced;
    zs.avail_out = (uInt)room;
    int rc = inflate(&zs, Z_NO_FLUSH);
    produced += room - zs.avail_out;
    if (rc == Z_STREAM_END) { result = 0; break; }
    if (rc == Z_OK) continue;
    if (rc == Z_BUF_ERROR && zs.avail_out == 0) continue;  // output full: grow and go on
    result = rc == Z_MEM_ERROR ? 3 : 1;                     // Z_DATA_ERROR, or Z_BUF_ERROR with the input exhausted = truncated
    break;
  }
  out.resize(result == 0 ? produced : 0);
  return result;
}

}  // namespace am355
