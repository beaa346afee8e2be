// Parallel raw-DEFLATE decode of one long stream (see am355_pinflate.h). Host code, no device work.
// The format is RFC 1951; the validity rules for code-length sets are the ones zlib's inftrees.c enforces (pako is a port of
// it, columnar.js:1062-1067 calls pako.inflateRaw): over-subscribed sets are errors, incomplete sets are errors unless the set
// is a single one-bit code, a block without an end-of-block code is an error.
#include "am355_pinflate.h"

#include <string.h>
#include <algorithm>
#include <functional>
#include <thread>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

namespace am355 {
namespace {

constexpr int LIT_ROOT = 11, DIST_ROOT = 9;
constexpr int LIT_CAP = (1 << LIT_ROOT) + 288 * 16, DIST_CAP = (1 << DIST_ROOT) + 32 * 64;
constexpr uint32_t LIT_MASK = (1u << LIT_ROOT) - 1, DIST_MASK = (1u << DIST_ROOT) - 1;
// table entry: value << 16 | extra bits (or sub-table bits) << 12 | kind << 8 | bits to consume
enum : uint32_t { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
inline constexpr uint32_t mk(uint32_t value, uint32_t kind, uint32_t extra, uint32_t nbits) { return (value << 16) | (extra << 12) | (kind << 8) | nbits; }
inline uint32_t e_kind(uint32_t e) { return (e >> 8) & 15; }
inline uint32_t e_extra(uint32_t e) { return (e >> 12) & 15; }
inline uint32_t e_bits(uint32_t e) { return e & 255; }
inline uint32_t e_value(uint32_t e) { return e >> 16; }

struct Tables {
  uint32_t lit[LIT_CAP];
  uint32_t dist[DIST_CAP];
};

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t lit_entry(int sym, uint32_t nbits) {
  if (sym < 256) return mk((uint32_t)sym, K_LIT, 0, nbits);
  if (sym == 256) return mk(0, K_EOB, 0, nbits);
  if (sym < 286) return mk(LEN_BASE[sym - 257], K_LEN, LEN_EXTRA[sym - 257], nbits);
  return mk(0, K_BAD, 0, nbits);
}
inline uint32_t dist_entry(int sym, uint32_t nbits) {
  if (sym < 30) return mk(DIST_BASE[sym], K_LEN, DIST_EXTRA[sym], nbits);
  return mk(0, K_BAD, 0, nbits);
}

inline uint32_t reverse_bits(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1); code >>= 1; }
  return r;
}

// Canonical Huffman decoding table over LSB-first bit strings: `root` index bits, longer codes through sub-tables.
// false: over-subscribed, or incomplete and not the single one-bit code (inftrees.c: "incomplete set").
bool build_table(const uint8_t* lens, int n, int root, uint32_t* table, bool is_dist) {
  int count[16] = {0};
  for (int i = 0; i < n; i++) count[lens[i]]++;
  int max = 15;
  while (max > 0 && !count[max]) max--;
  const uint32_t bad = mk(0, K_BAD, 0, 1);
  const int root_size = 1 << root;
  for (int i = 0; i < root_size; i++) table[i] = bad;
  if (max == 0) return true;  // no code at all: legal to declare, an error to use
  int left = 1;
  for (int len = 1; len <= 15; len++) {
    left <<= 1;
    left -= count[len];
    if (left < 0) return false;
  }
  if (left > 0 && max != 1) return false;
  uint32_t next_code[16];
  {
    uint32_t code = 0;
    count[0] = 0;
    for (int bits = 1; bits <= 15; bits++) { code = (code + (uint32_t)count[bits - 1]) << 1; next_code[bits] = code; }
  }
  uint8_t sub_bits[1 << LIT_ROOT];
  uint32_t rev[288];
  bool any_long = false;
  if (max > root) memset(sub_bits, 0, (size_t)root_size);
  for (int sym = 0; sym < n; sym++) {
    int len = lens[sym];
    if (!len) continue;
    uint32_t r = reverse_bits(next_code[len]++, len);
    rev[sym] = r;
    if (len <= root) {
      uint32_t e = is_dist ? dist_entry(sym, (uint32_t)len) : lit_entry(sym, (uint32_t)len);
      for (uint32_t i = r; i < (uint32_t)root_size; i += 1u << len) table[i] = e;
    } else {
      uint32_t prefix = r & ((1u << root) - 1);
      if (len - root > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(len - root);
      any_long = true;
    }
  }
  if (!any_long) return true;
  uint32_t next_off = (uint32_t)root_size;
  for (int prefix = 0; prefix < root_size; prefix++) {
    if (!sub_bits[prefix]) continue;
    uint32_t size = 1u << sub_bits[prefix];
    table[prefix] = mk(next_off, K_SUB, sub_bits[prefix], (uint32_t)root);
    for (uint32_t i = 0; i < size; i++) table[next_off + i] = bad;
    next_off += size;
  }
  for (int sym = 0; sym < n; sym++) {
    int len = lens[sym];
    if (len <= root) continue;
    uint32_t r = rev[sym], prefix = r & ((1u << root) - 1);
    uint32_t off = e_value(table[prefix]), sb = e_extra(table[prefix]);
    uint32_t e = is_dist ? dist_entry(sym, (uint32_t)(len - root)) : lit_entry(sym, (uint32_t)(len - root));
    for (uint32_t i = r >> root; i < (1u << sb); i += 1u << (len - root)) table[off + i] = e;
  }
  return true;
}

const Tables& fixed_tables() {
  static const Tables* t = []() {
    Tables* x = new Tables;
    uint8_t lens[288];
    for (int i = 0; i < 144; i++) lens[i] = 8;
    for (int i = 144; i < 256; i++) lens[i] = 9;
    for (int i = 256; i < 280; i++) lens[i] = 7;
    for (int i = 280; i < 288; i++) lens[i] = 8;
    build_table(lens, 288, LIT_ROOT, x->lit, false);
    uint8_t dl[32];
    for (int i = 0; i < 32; i++) dl[i] = 5;
    build_table(dl, 32, DIST_ROOT, x->dist, true);
    return x;
  }();
  return *t;
}

// LSB-first bit reader with a 64-bit buffer. Past the end of the input it reads zero bytes and raises `over` once more than
// sixteen of them were taken (a truncated stream; every caller checks the bit position at block boundaries as well).
struct BitIn {
  const uint8_t* in;
  size_t len;
  size_t pos = 0;
  uint64_t buf = 0;
  unsigned cnt = 0;
  bool over = false;
  BitIn(const uint8_t* p, size_t n) : in(p), len(n) {}
  inline void refill() {
    if (__builtin_expect(pos + 8 <= len, 1)) {
      uint64_t w;
      memcpy(&w, in + pos, 8);
      buf |= w << cnt;
      pos += (63 - cnt) >> 3;
      cnt |= 56;
    } else {
      while (cnt <= 56) {
        uint64_t byte = pos < len ? in[pos] : 0;
        buf |= byte << cnt;
        pos++;
        cnt += 8;
      }
      if (pos > len + 16) over = true;
    }
  }
  void seek(size_t bit) {
    pos = bit >> 3;
    buf = 0;
    cnt = 0;
    refill();
    unsigned s = (unsigned)(bit & 7);
    buf >>= s;
    cnt -= s;
  }
  inline size_t bitpos() const { return pos * 8 - cnt; }
  inline uint32_t take(unsigned n) {
    uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
    buf >>= n;
    cnt -= n;
    return v;
  }
};

// header of a dynamic block (after the three block-type bits): code-length code, then the literal/length and distance sets
bool read_dynamic(BitIn& b, Tables& T) {
  b.refill();
  unsigned hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  uint8_t pl[19] = {0};
  for (unsigned i = 0; i < hclen; i++) {
    if (b.cnt < 3) b.refill();
    pl[order[i]] = (uint8_t)b.take(3);
  }
  // the code-length code must be complete (inftrees.c, type CODES); all-zero is an error as well (no end-of-block can follow)
  int count[8] = {0};
  for (int i = 0; i < 19; i++) count[pl[i]]++;
  if (count[0] == 19) return false;
  int left = 1;
  for (int len = 1; len <= 7; len++) {
    left <<= 1;
    left -= count[len];
    if (left < 0) return false;
  }
  if (left > 0) return false;
  uint8_t ptab[128];  // sym << 3 | len
  {
    uint32_t next_code[8], code = 0;
    count[0] = 0;
    for (int bits = 1; bits <= 7; bits++) { code = (code + (uint32_t)count[bits - 1]) << 1; next_code[bits] = code; }
    for (int sym = 0; sym < 19; sym++) {
      int len = pl[sym];
      if (!len) continue;
      uint32_t r = reverse_bits(next_code[len]++, len);
      for (uint32_t i = r; i < 128; i += 1u << len) ptab[i] = (uint8_t)((sym << 3) | len);
    }
  }
  uint8_t lens[320];
  const unsigned n = hlit + hdist;
  unsigned i = 0;
  while (i < n) {
    if (b.cnt < 14) b.refill();
    uint8_t e = ptab[b.buf & 127];
    b.buf >>= (e & 7);
    b.cnt -= (e & 7);
    unsigned sym = e >> 3;
    if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
    unsigned rep;
    uint8_t val = 0;
    if (sym == 16) {
      if (i == 0) return false;
      val = lens[i - 1];
      rep = 3 + b.take(2);
    } else if (sym == 17) {
      rep = 3 + b.take(3);
    } else {
      rep = 11 + b.take(7);
    }
    if (i + rep > n) return false;
    while (rep--) lens[i++] = val;
  }
  if (b.over) return false;
  if (lens[256] == 0) return false;  // "invalid code -- missing end-of-block"
  if (!build_table(lens, (int)hlit, LIT_ROOT, T.lit, false)) return false;
  if (!build_table(lens + hlit, (int)hdist, DIST_ROOT, T.dist, true)) return false;
  return true;
}

// growing symbol buffer of one chunk
struct SymOut {
  PInflateChunk* ch;
  uint16_t* out;    // next symbol
  uint16_t* limit;  // grow when out passes it (keeps room for one loop iteration: three literals + one match + copy overshoot)
  size_t cap_syms = ~(size_t)0;               // the job's output cap: the chunk alone must not pass it ...
  std::atomic<size_t>* job_made = nullptr;    // ... nor all chunks of the job together (symbols made so far, reported at every grow)
  static constexpr size_t SLACK = 280;
  bool reserve(size_t cap_syms_) {
    if (cap_syms_ <= ch->sym_cap) return true;
    uint16_t* q = (uint16_t*)realloc(ch->sym, cap_syms_ * sizeof(uint16_t));
    if (!q) return false;
    ch->sym = q;
    ch->sym_cap = cap_syms_;
    return true;
  }
  size_t reported = 0;   // symbols of this chunk already counted in *job_made
  bool grow() {
    size_t at = (size_t)(out - ch->sym);
    // the cap is enforced INSIDE the decode (it used to be looked at at block boundaries only, and per chunk: one long block of a
    // crafted column could ask for ~2 KB of symbols per input byte, N chunks for N x cap, before anybody compared -- ADVICE r5).
    // Together the chunks may make half as many symbols again as the stream may have: a chunk that started from a false block
    // header decodes noise until the data stops making sense, and that is no reason to give a sound stream up.
    const size_t made = at > PINFLATE_WINDOW ? at - PINFLATE_WINDOW : 0;
    if (made > cap_syms) return false;
    if (job_made) {
      const size_t add = made - reported;
      reported = made;
      if (job_made->fetch_add(add, std::memory_order_relaxed) + add > cap_syms + cap_syms / 2 + ((size_t)1 << 20)) return false;
    }
    if (!reserve(ch->sym_cap + ch->sym_cap / 2 + 4096)) return false;
    out = ch->sym + at;
    limit = ch->sym + ch->sym_cap - SLACK;
    return true;
  }
};

// Symbols of one block body up to its end-of-block code. DRY: nothing is stored, the symbols are only counted (search).
// 0 = end of block reached, -1 = invalid data / truncated, -2 = out of memory, -3 = dry run beyond max_syms
template <bool DRY>
int decode_body(BitIn& b, const Tables& T, SymOut* o, size_t max_syms) {
  size_t dry_count = 0;
  uint16_t* out = DRY ? nullptr : o->out;
  for (;;) {
    if (DRY) {
      if (dry_count > max_syms) return -3;
    } else if (__builtin_expect(out > o->limit, 0)) {
      o->out = out;
      if (!o->grow()) return -2;
      out = o->out;
    }
    b.refill();
    if (__builtin_expect(b.over, 0)) return -1;
    uint32_t e;
#define AM355_LIT_LOOKUP()                                                         \
  e = T.lit[b.buf & LIT_MASK];                                                     \
  if (__builtin_expect(e_kind(e) == K_SUB, 0)) {                                   \
    b.buf >>= LIT_ROOT;                                                            \
    b.cnt -= LIT_ROOT;                                                             \
    e = T.lit[e_value(e) + (uint32_t)(b.buf & ((1u << e_extra(e)) - 1))];          \
  }                                                                                \
  b.buf >>= e_bits(e);                                                             \
  b.cnt -= e_bits(e);
    AM355_LIT_LOOKUP();
    if ((e & 0xf00) == 0) {
      if (DRY) dry_count++; else *out++ = (uint16_t)e_value(e);
      AM355_LIT_LOOKUP();
      if ((e & 0xf00) == 0) {
        if (DRY) dry_count++; else *out++ = (uint16_t)e_value(e);
        AM355_LIT_LOOKUP();
        if ((e & 0xf00) == 0) {
          if (DRY) dry_count++; else *out++ = (uint16_t)e_value(e);
          continue;
        }
      }
    }
#undef AM355_LIT_LOOKUP
    const uint32_t kind = e_kind(e);
    if (kind == K_EOB) {
      if (!DRY) o->out = out;
      return 0;
    }
    if (kind != K_LEN) return -1;
    uint32_t length = e_value(e) + b.take(e_extra(e));  // (at most 45 + 5 bits since the refill)
    b.refill();
    uint32_t d = T.dist[b.buf & DIST_MASK];
    if (__builtin_expect(e_kind(d) == K_SUB, 0)) {
      b.buf >>= DIST_ROOT;
      b.cnt -= DIST_ROOT;
      d = T.dist[e_value(d) + (uint32_t)(b.buf & ((1u << e_extra(d)) - 1))];
    }
    b.buf >>= e_bits(d);
    b.cnt -= e_bits(d);
    if (e_kind(d) != K_LEN) return -1;
    uint32_t dist = e_value(d) + b.take(e_extra(d));
    if (DRY) { dry_count += length; continue; }
    // the marker prefix in front of the chunk's output makes every distance up to 32768 a plain copy
    uint16_t* s = out - dist;
    uint16_t* dst = out;
    uint16_t* end = out + length;
    if (dist >= 4) {
      do { memcpy(dst, s, 8); dst += 4; s += 4; } while (dst < end);
    } else if (dist == 1) {
      uint16_t v = *s;
      do { *dst++ = v; } while (dst < end);
    } else {
      do { *dst++ = *s++; } while (dst < end);
    }
    out = end;
  }
}

inline void cpu_pause() {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

}  // namespace

void PInflateJob::prepare(const uint8_t* in_, size_t in_len_, size_t cap_, size_t chunk_bytes) {
  in = in_;
  in_len = in_len_;
  cap = cap_;
  ok = false;
  out_len = 0;
  last_byte = 0;
  chain.clear();
  resolve_failed.store(0);
  made_syms.store(0);
  if (chunk_bytes < 512) chunk_bytes = 512;
  n_chunks = (unsigned)std::max<size_t>(1, in_len / chunk_bytes);
  while (chunks.size() < n_chunks) chunks.emplace_back(new PInflateChunk);
  for (unsigned k = 0; k < n_chunks; k++) {
    PInflateChunk& ch = *chunks[k];
    ch.nominal_bit = (in_len * k / n_chunks) * 8;
    ch.limit_bit = k + 1 < n_chunks ? (in_len * (k + 1) / n_chunks) * 8 : in_len * 8;
    ch.start_bit.store(k == 0 ? 0 : PENDING, std::memory_order_relaxed);
    ch.n_out = 0;
    ch.next = 0;
    ch.status = 0;
    ch.out_off = 0;
  }
  std::atomic_thread_fence(std::memory_order_release);
}

void PInflateJob::search(unsigned k) {
  PInflateChunk& ch = *chunks[k];
  if (k == 0) return;  // the stream's first block starts at bit 0
  const size_t total_bits = in_len * 8;
  std::unique_ptr<Tables> T(new Tables);
  uint64_t found = NOT_FOUND;
  for (size_t byte = ch.nominal_bit >> 3; byte * 8 < ch.limit_bit && found == NOT_FOUND; byte++) {
    uint64_t w = 0;
    if (byte + 8 <= in_len) memcpy(&w, in + byte, 8);
    else if (byte < in_len) memcpy(&w, in + byte, in_len - byte);
    else break;
    for (unsigned s = 0; s < 8; s++) {
      uint64_t v = w >> s;
      if ((v & 7) != 4) continue;                                     // not the last block, dynamic codes
      if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;     // at most 286 literal/length and 30 distance codes
      size_t bit = byte * 8 + s;
      if (bit < ch.nominal_bit || bit >= ch.limit_bit) continue;
      {
        // the code-length code must be complete: 4 + HCLEN three-bit lengths from bit 17 on, Kraft sum exactly 1 -- from one more
        // 64-bit load, before the bit reader is set up (one candidate in a hundred passes)
        const size_t pos2 = bit + 17;
        if ((pos2 >> 3) + 8 <= in_len) {
          uint64_t q;
          memcpy(&q, in + (pos2 >> 3), 8);
          q >>= (pos2 & 7);
          static const uint8_t kraft[8] = {0, 64, 32, 16, 8, 4, 2, 1};
          uint32_t sum = 0;
          for (unsigned n = (unsigned)((v >> 13) & 15) + 4; n; n--, q >>= 3) sum += kraft[q & 7];
          if (sum != 128) continue;
        }
      }
      BitIn b(in, in_len);
      b.seek(bit + 3);
      if (!read_dynamic(b, *T) || b.bitpos() > total_bits) continue;
      // (no trial decode of the block: three complete prefix codes accept any bit string, so it would only tell at the block's
      // end -- a third of a millisecond per chunk -- what the chain of chunks tells anyway: a start nobody lands on is skipped)
      found = bit;
      break;
    }
  }
  ch.start_bit.store(found, std::memory_order_release);
}

void PInflateJob::decode(unsigned k) {
  PInflateChunk& ch = *chunks[k];
  uint64_t start;
  while ((start = ch.start_bit.load(std::memory_order_acquire)) == PENDING) cpu_pause();
  if (start == NOT_FOUND) return;
  ch.status = 2;
  const size_t total_bits = in_len * 8;
  SymOut o{&ch, nullptr, nullptr};
  o.cap_syms = cap;
  o.job_made = &made_syms;
  {
    size_t comp = (ch.limit_bit - ch.nominal_bit) / 8;
    if (!o.reserve(PINFLATE_WINDOW + comp * 4 + 4096)) return;
  }
  for (uint32_t w = 0; w < PINFLATE_WINDOW; w++) ch.sym[w] = (uint16_t)(256 + w);
  o.out = ch.sym + PINFLATE_WINDOW;
  o.limit = ch.sym + ch.sym_cap - SymOut::SLACK;
  BitIn b(in, in_len);
  b.seek((size_t)start);
  std::unique_ptr<Tables> T(new Tables);
  unsigned j = k + 1;  // the next chunk whose start this one may still land on
  for (;;) {
    const size_t bp = b.bitpos();
    if (bp > total_bits || b.over) return;
    // block boundary: is it the start another chunk decodes from?
    bool stop = false;
    while (j < n_chunks && bp >= chunks[j]->nominal_bit) {
      uint64_t s;
      while ((s = chunks[j]->start_bit.load(std::memory_order_acquire)) == PENDING) cpu_pause();
      if (s == NOT_FOUND || s < bp) { j++; continue; }  // nothing found there, or a position this chunk passed over: not a block start
      if (s == bp) stop = true;
      break;
    }
    if (stop) { ch.next = j; break; }
    if (bp + 3 > total_bits) return;
    b.refill();
    const uint32_t final_block = b.take(1), type = b.take(2);
    if (type == 3) return;
    if (type == 0) {
      size_t byte = (b.bitpos() + 7) >> 3;
      if (byte + 4 > in_len) return;
      uint32_t len = in[byte] | (in[byte + 1] << 8), nlen = in[byte + 2] | (in[byte + 3] << 8);
      if (len != (~nlen & 0xffffu) || byte + 4 + len > in_len) return;
      while ((size_t)(o.limit - o.out) < len || o.out > o.limit)
        if (!o.grow()) return;
      for (uint32_t i = 0; i < len; i++) o.out[i] = in[byte + 4 + i];
      o.out += len;
      b.seek((byte + 4 + len) * 8);
    } else {
      const Tables* use = &fixed_tables();
      if (type == 2) {
        if (!read_dynamic(b, *T)) return;
        use = T.get();
      }
      if (decode_body<false>(b, *use, &o, 0) != 0) return;
    }
    if ((size_t)(o.out - ch.sym) - PINFLATE_WINDOW > cap) return;
    if (final_block) {
      if (b.bitpos() > total_bits) return;
      ch.next = n_chunks;
      break;
    }
  }
  ch.n_out = (size_t)(o.out - ch.sym) - PINFLATE_WINDOW;
  ch.status = 1;
}

bool PInflateJob::link() {
  ok = false;
  chain.clear();
  size_t total = 0;
  unsigned i = 0;
  for (;;) {
    PInflateChunk& ch = *chunks[i];
    if (ch.status != 1) return false;
    ch.out_off = total;
    total += ch.n_out;
    if (total > cap) return false;
    chain.push_back(i);
    if (ch.next >= n_chunks) break;
    if (ch.next <= i) return false;
    i = ch.next;
  }
  // windows: the last 32 KiB of (window of the chunk in front ++ its bytes); the marker prefix of the chunk in front stands
  // for its own window, so this is the resolved tail of its prefixed symbol buffer. Kept as a table over SYMBOLS (0..255 the byte
  // itself, 256 + w the window): resolving is one load per symbol, no branch on "is it a marker"
  for (size_t ci = 0; ci < chain.size(); ci++) {
    PInflateChunk& ch = *chunks[chain[ci]];
    ch.lut.resize(256 + PINFLATE_WINDOW);
    uint8_t* lut = ch.lut.data();
    for (int v = 0; v < 256; v++) lut[v] = (uint8_t)v;
    uint8_t* w = lut + 256;
    if (ci == 0) { memset(w, 0, PINFLATE_WINDOW); continue; }
    const PInflateChunk& prev = *chunks[chain[ci - 1]];
    const uint16_t* tail = prev.sym + prev.n_out;  // = prefix + n_out - WINDOW
    const uint8_t* pl = prev.lut.data();
    for (uint32_t x = 0; x < PINFLATE_WINDOW; x++) w[x] = pl[tail[x]];
  }
  out_len = total;
  if (total) {
    // last byte of the stream (the caller checks that a column ends on the last byte of a number before the bytes are resolved)
    const PInflateChunk* lc = nullptr;
    for (size_t ci = chain.size(); ci-- > 0;)
      if (chunks[chain[ci]]->n_out) { lc = chunks[chain[ci]].get(); break; }
    last_byte = lc->lut[lc->sym[PINFLATE_WINDOW + lc->n_out - 1]];
  }
  ok = true;
  return true;
}

void PInflateJob::resolve(unsigned ci, unsigned r, uint8_t* dst) {
  const PInflateChunk& ch = *chunks[chain[ci]];
  const size_t b0 = (size_t)r * PIECE, b1 = std::min(ch.n_out, b0 + PIECE);
  const uint16_t* src = ch.sym + PINFLATE_WINDOW;
  uint8_t* d = dst + ch.out_off;
  const uint8_t* lut = ch.lut.data();
  // a marker that points in front of the stream's first byte is zlib's "invalid distance too far back" (every marker of the
  // first chunk; in a later chunk only while less than a window of output lies in front of it): symbols in [256, too_far_end)
  const uint32_t too_far_end = 256 + (ch.out_off >= PINFLATE_WINDOW ? 0 : (uint32_t)(PINFLATE_WINDOW - ch.out_off));
  uint32_t too_far = 0;
  size_t x = b0;
#if defined(__x86_64__)
  const __m128i hi = _mm_set1_epi16((short)0xff00);
  for (; x + 16 <= b1; x += 16) {
    __m128i a = _mm_loadu_si128((const __m128i*)(src + x)), c = _mm_loadu_si128((const __m128i*)(src + x + 8));
    if (_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_and_si128(_mm_or_si128(a, c), hi), _mm_setzero_si128())) == 0xffff) {
      _mm_storeu_si128((__m128i*)(d + x), _mm_packus_epi16(a, c));
    } else {
      for (size_t y = x; y < x + 16; y++) {
        const uint32_t s = src[y];
        too_far |= (uint32_t)(s - 256 < too_far_end - 256);
        d[y] = lut[s];
      }
    }
  }
#endif
  for (; x < b1; x++) {
    const uint32_t s = src[x];
    too_far |= (uint32_t)(s - 256 < too_far_end - 256);
    d[x] = lut[s];
  }
  if (too_far) resolve_failed.store(1, std::memory_order_relaxed);
}

bool PInflateJob::has_too_far_marker() const {
  for (size_t ci = 0; ci < chain.size(); ci++) {
    const PInflateChunk& ch = *chunks[chain[ci]];
    if (ch.out_off >= PINFLATE_WINDOW) break;   // (a window of output lies in front of this chunk and of every later one)
    const uint32_t too_far_end = 256 + (uint32_t)(PINFLATE_WINDOW - ch.out_off);
    const uint16_t* src = ch.sym + PINFLATE_WINDOW;
    for (size_t x = 0; x < ch.n_out; x++)
      if ((uint32_t)(src[x] - 256) < too_far_end - 256) return true;
  }
  return false;
}

void PInflateJob::trim(size_t keep_chunk_bytes, size_t keep_job_bytes) {
  size_t kept = 0;
  for (auto& ch : chunks) {
    const size_t bytes = ch->sym_cap * sizeof(uint16_t);
    if (bytes > keep_chunk_bytes || kept + bytes > keep_job_bytes) {
      free(ch->sym);
      ch->sym = nullptr;
      ch->sym_cap = 0;
    } else {
      kept += bytes;
    }
    std::vector<uint8_t>().swap(ch->lut);
  }
}

int inflate_raw_parallel(const uint8_t* in, size_t in_len, std::vector<uint8_t>& out, size_t cap, size_t chunk_bytes, unsigned n_threads) {
  PInflateJob job;
  job.prepare(in, in_len, cap, chunk_bytes);
  if (n_threads < 1) n_threads = 1;
  auto run = [&](unsigned n_tasks, const std::function<void(unsigned)>& fn) {
    std::atomic<unsigned> next{0};
    auto body = [&]() {
      for (;;) {
        unsigned t = next.fetch_add(1);
        if (t >= n_tasks) break;
        fn(t);
      }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < n_threads; i++) th.emplace_back(body);
    body();
    for (auto& t : th) t.join();
  };
  const unsigned n = job.n_chunks;
  // (search tasks first: a decode task spins for searches, all of which have been drawn by then)
  run(2 * n, [&](unsigned t) { if (t < n) job.search(t); else job.decode(t - n); });
  if (!job.link()) return 1;
  out.resize(job.out_len);
  std::vector<std::pair<unsigned, unsigned>> pieces;
  for (unsigned ci = 0; ci < job.chain.size(); ci++)
    for (unsigned r = 0; r < job.n_pieces(ci); r++) pieces.emplace_back(ci, r);
  run((unsigned)pieces.size(), [&](unsigned t) { job.resolve(pieces[t].first, pieces[t].second, out.data()); });
  if (job.resolve_failed.load()) return 1;
  return 0;
}

}  // namespace am355
