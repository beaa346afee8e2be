// See am355_history.h. Host side of the history reconstruction after Backend.load.
//
// Step by step (reference lines in the header):
//   1. change metadata columns -> one record per change (actor, seq, maxOp, time, message, deps by index, extra bytes)
//   2. every op id the document mentions -- ids of rows and ids in succ lists -- gets a SLOT: ids are numbered per actor in counter
//      order through a bitmap + popcount directory, which is exactly the order of ops inside a change and of changes inside an actor.
//      An id in a succ list that no row carries is a deletion (groupChangeOps rebuilds a `del` op for it).
//   3. preds: the inverse of the succ lists (CSR by slot), sorted by (counter, actor) when a change is encoded
//   4. changes -> slot ranges: count of ids in (maxOp of the actor's previous change, maxOp]; startOp = maxOp - count + 1
//   5. every change is encoded independently (columns + header without the dependency hashes), in parallel
//   6. hashes chain in document order (dependencies come first in a document); containers are assembled (+ DEFLATE) in parallel
#include "am355_history.h"

#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../loggen/wire.hpp"
#include "am355_host.h"

namespace am355 {
namespace {

constexpr uint32_t NONE = 0xffffffffu;
using amlog::Bytes;
using amlog::OptInt;
using amlog::OptStr;
using amlog::put_sleb;
using amlog::put_uleb;

// ---- column readers for the change metadata (encoding.js:558-783 RLE, :932-1010 delta) ----
struct Reader {
  const uint8_t* p = nullptr;
  size_t len = 0, off = 0;
  bool uleb(uint64_t& out) {
    if (off < len && !(p[off] & 0x80)) { out = p[off++]; return true; }   // (one-byte numbers: nearly all of a metadata column)
    uint64_t v = 0;
    int shift = 0;
    while (off < len && shift < 64) {
      uint8_t b = p[off++];
      v |= (uint64_t)(b & 0x7f) << shift;
      shift += 7;
      if (!(b & 0x80)) { out = v; return true; }
    }
    return false;
  }
  bool sleb(int64_t& out) {
    if (off < len && !(p[off] & 0x80)) { const uint8_t b = p[off++]; out = (b & 0x40) ? (int64_t)b - 128 : (int64_t)b; return true; }
    uint64_t v = 0;
    int shift = 0;
    while (off < len && shift < 64) {
      uint8_t b = p[off++];
      v |= (uint64_t)(b & 0x7f) << shift;
      shift += 7;
      if (!(b & 0x80)) {
        if ((b & 0x40) && shift < 64) v |= ~0ull << shift;
        out = (int64_t)v;
        return true;
      }
    }
    return false;
  }
};

// kind: 0 unsigned numbers, 1 signed numbers, 2 strings
struct RleReader {
  Reader r;
  int kind;
  int state = 0;  // 1 repetition, 2 literal, 3 nulls
  int64_t count = 0, last = 0;
  std::string last_s;
  RleReader(const std::vector<uint8_t>* col, int k) : kind(k) {
    if (col) { r.p = col->data(); r.len = col->size(); }
  }
  bool done() const { return count == 0 && r.off >= r.len; }
  bool raw(int64_t& v, std::string& s) {
    if (kind == 2) {
      uint64_t n;
      if (!r.uleb(n) || n > r.len - r.off) return false;
      s.assign((const char*)r.p + r.off, (size_t)n);
      r.off += (size_t)n;
      return true;
    }
    if (kind == 1) return r.sleb(v);
    uint64_t u;
    if (!r.uleb(u)) return false;
    v = (int64_t)u;
    return true;
  }
  // consumes one run without materialising its values; adds its length to n
  bool skip_run(uint64_t& n) {
    int64_t c;
    if (!r.sleb(c)) return false;
    int64_t v;
    std::string s;
    if (c > 1) { if (!raw(v, s)) return false; n += (uint64_t)c; }
    else if (c == 1) return false;
    else if (c < 0) {
      if (c == INT64_MIN) return false;
      for (int64_t i = 0; i < -c; i++) if (!raw(v, s)) return false;
      n += (uint64_t)-c;
    } else { uint64_t z; if (!r.uleb(z) || z == 0) return false; n += z; }
    return true;
  }
  // numeric columns: the same as next() without the string (the dependency index column alone holds a value per dependency edge)
  inline bool next_num(bool& is_null, int64_t& v) {
    if (count == 0) {
      if (r.off >= r.len) { is_null = true; return true; }   // a column that has run out yields nulls
      int64_t n;
      if (!r.sleb(n)) return false;
      if (n > 1) { if (!raw_num(last)) return false; state = 1; count = n; }
      else if (n == 1) return false;
      else if (n < 0) { state = 2; count = -n; }
      else { uint64_t z; if (!r.uleb(z) || z == 0) return false; state = 3; count = (int64_t)z; }
    }
    count--;
    if (state == 2 && !raw_num(last)) return false;
    is_null = state == 3;
    v = last;
    return true;
  }
  inline bool raw_num(int64_t& v) {
    if (kind == 1) return r.sleb(v);
    uint64_t u;
    if (!r.uleb(u)) return false;
    v = (int64_t)u;
    return true;
  }
  // a column that has run out yields nulls (encoding.js:639-642)
  bool next(bool& is_null, int64_t& v, std::string& s) {
    if (done()) { is_null = true; return true; }
    if (count == 0) {
      int64_t n;
      if (!r.sleb(n)) return false;
      if (n > 1) { if (!raw(last, last_s)) return false; state = 1; count = n; }
      else if (n == 1) return false;
      else if (n < 0) { state = 2; count = -n; }
      else { uint64_t z; if (!r.uleb(z) || z == 0) return false; state = 3; count = (int64_t)z; }
    }
    count--;
    if (state == 2 && !raw(last, last_s)) return false;
    is_null = state == 3;
    v = last;
    s = last_s;
    return true;
  }
};

using ChangeRec = HistoryChange;

inline void atomic_or(uint64_t* w, uint64_t bit) { __atomic_fetch_or(w, bit, __ATOMIC_RELAXED); }
inline uint64_t atomic_or_old(uint64_t* w, uint64_t bit) { return __atomic_fetch_or(w, bit, __ATOMIC_RELAXED); }
inline void atomic_min(uint32_t* p, uint32_t v) {
  uint32_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

// valid UTF-8 that TextDecoder -> TextEncoder reproduces (a leading U+FEFF is dropped by the decoder, encoding.js:9-17)
bool valid_utf8(const uint8_t* s, size_t n) {
  if (n >= 3 && s[0] == 0xef && s[1] == 0xbb && s[2] == 0xbf) return false;
  size_t i = 0;
  while (i < n) {
    uint8_t c = s[i];
    if (c < 0x80) { i++; continue; }
    int extra;
    uint32_t cp;
    if ((c & 0xe0) == 0xc0) { extra = 1; cp = c & 0x1f; }
    else if ((c & 0xf0) == 0xe0) { extra = 2; cp = c & 0x0f; }
    else if ((c & 0xf8) == 0xf0) { extra = 3; cp = c & 0x07; }
    else return false;
    if (i + extra >= n) return false;
    for (int k = 1; k <= extra; k++) {
      if ((s[i + k] & 0xc0) != 0x80) return false;
      cp = cp << 6 | (s[i + k] & 0x3f);
    }
    if ((extra == 1 && cp < 0x80) || (extra == 2 && cp < 0x800) || (extra == 3 && (cp < 0x10000 || cp > 0x10ffff)) || (cp >= 0xd800 && cp <= 0xdfff)) return false;
    i += extra + 1;
  }
  return true;
}

// A value the reference's decodeValue -> encodeValue round trip (columnar.js:259-329) reproduces byte for byte?
// 0 yes; HISTORY_INVALID: the reference throws on it; HISTORY_UNSUPPORTED: the reference writes something else (non-minimal numbers,
// byte arrays -- encodeValue writes the whole underlying buffer of a decoded byte array --, unknown type tags).
int value_round_trips(uint32_t tl, const uint8_t* bytes) {
  uint32_t tag = tl & 15, len = tl >> 4;
  switch (tag) {
    case 0: case 1: case 2: return len ? HISTORY_UNSUPPORTED : 0;
    case 3: {
      Reader r; r.p = bytes; r.len = len;
      uint64_t v;
      if (!r.uleb(v) || r.off != len || v >= (1ull << 53)) return HISTORY_INVALID;
      Bytes b; put_uleb(b, v);
      return b.size() == len ? 0 : HISTORY_UNSUPPORTED;
    }
    case 4: case 8: case 9: {
      Reader r; r.p = bytes; r.len = len;
      int64_t v;
      if (!r.sleb(v) || r.off != len || v >= (1ll << 53) || v <= -(1ll << 53)) return HISTORY_INVALID;
      Bytes b; put_sleb(b, v);
      return b.size() == len ? 0 : HISTORY_UNSUPPORTED;
    }
    case 5: return len == 8 ? 0 : HISTORY_INVALID;
    case 6: return valid_utf8(bytes, len) ? 0 : HISTORY_UNSUPPORTED;
    default: return HISTORY_UNSUPPORTED;
  }
}

}  // namespace

int history_metadata(const HistoryInput& in, HistoryMeta& meta, std::string& err) {
  auto bad = [&](int rc, const char* msg) { err = msg; return rc; };
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_lap = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  history: %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
    t_lap = now;
  };
  const std::vector<std::string>& actors = *in.actors;
  const uint32_t NA = (uint32_t)actors.size(), N = in.n_rows, P = in.n_succ;

  // the reference reads rows until every column is exhausted (columnar.js:577-590): the key column must hold N values or none
  // (and the value bytes must be consumed exactly: summed on the device, checked by the caller)
  {
    RleReader keys(nullptr, 2);
    keys.r.p = in.key_column; keys.r.len = in.key_column_len;
    uint64_t n_keys = 0;
    while (!keys.done()) {
      if (!keys.skip_run(n_keys)) return bad(HISTORY_INVALID, "malformed key column");
      if (n_keys > (uint64_t)N) break;
    }
    if (n_keys != 0 && n_keys != N) return bad(HISTORY_UNSUPPORTED, "key column and action column differ in length: the JS path decides");
  }
  lap("column lengths");
  // ---- 1. change metadata ----
  std::vector<ChangeRec>& chg = meta.chg;
  std::vector<uint32_t>& dep_index = meta.dep_index;
  chg.clear();
  dep_index.clear();
  {
    const std::vector<uint8_t>* col[9] = {};
    static const uint32_t ids[9] = {0x01, 0x03, 0x13, 0x23, 0x35, 0x40, 0x43, 0x56, 0x57};
    for (auto& c : *in.change_columns) {
      int k = -1;
      for (int i = 0; i < 9; i++) if (ids[i] == c.first) k = i;
      if (k < 0) { if (!c.second.empty()) return bad(HISTORY_UNSUPPORTED, "document has change columns this engine does not model"); continue; }
      col[k] = &c.second;
    }
    RleReader r_actor(col[0], 0), r_seq(col[1], 1), r_max(col[2], 1), r_time(col[3], 1), r_msg(col[4], 2), r_dnum(col[5], 0), r_xlen(col[7], 0);
    int64_t seq_abs = 0, max_abs = 0, time_abs = 0;
    uint64_t n_deps = 0;
    size_t xoff = 0;
    const size_t xtotal = col[8] ? col[8]->size() : 0;
    std::vector<uint32_t> last_of(NA, NONE);
    while (!(r_actor.done() && r_seq.done() && r_max.done() && r_time.done() && r_msg.done() && r_dnum.done() && r_xlen.done())) {
      ChangeRec c;
      bool nul;
      int64_t v;
      std::string s;
      if (!r_actor.next_num(nul, v)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul || v < 0 || (uint64_t)v >= in.doc_actor_rank->size()) return bad(HISTORY_INVALID, "bad actor index in change metadata");
      c.actor = (*in.doc_actor_rank)[(size_t)v];
      if (!r_seq.next_num(nul, v)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul) return bad(HISTORY_INVALID, "change without seq");
      seq_abs += v; c.seq = (uint64_t)seq_abs;
      if (!r_max.next_num(nul, v)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul) return bad(HISTORY_INVALID, "change without maxOp");
      max_abs += v;
      if (max_abs < 0 || max_abs > 0x7fffffff) return bad(HISTORY_UNSUPPORTED, "maxOp beyond 2^31");
      c.max_op = (uint64_t)max_abs;
      if (!r_time.next_num(nul, v)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul) return bad(HISTORY_INVALID, "change without time");
      time_abs += v; c.time = time_abs;
      if (!r_msg.next(nul, v, s)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (!nul) {
        if (!valid_utf8((const uint8_t*)s.data(), s.size())) return bad(HISTORY_UNSUPPORTED, "message is not valid UTF-8");
        c.message = s;
      }
      if (!r_dnum.next_num(nul, v)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul || v < 0 || v > 0x7fffffff) return bad(HISTORY_INVALID, "bad dependency count");
      // (the dependency INDEXES -- one value per edge of the hash graph, 63 per change in a log of 64 synced actors: the one long
      // column of the metadata -- are decoded by history_dependencies, which the caller runs beside the device stages)
      if (n_deps + (uint64_t)v > 0x7fffffffull) return bad(HISTORY_UNSUPPORTED, "too many dependency edges");
      c.dep_first = (uint32_t)n_deps;
      c.dep_num = (uint32_t)v;
      n_deps += (uint64_t)v;
      if (!r_xlen.next_num(nul, v)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul || (v & 15) != 7) return bad(HISTORY_INVALID, "Bad datatype for extra bytes");
      size_t xl = (size_t)(v >> 4);
      if (xl > xtotal - xoff) return bad(HISTORY_INVALID, "extra bytes column too short");
      if (xl) c.extra.assign((const char*)col[8]->data() + xoff, xl);
      xoff += xl;
      // seq continuity and maxOp monotonic per actor (columnar.js:881-886)
      uint32_t prev = last_of[c.actor];
      if (c.seq != (prev == NONE ? 1 : chg[prev].seq + 1)) return bad(HISTORY_INVALID, "unexpected seq");
      if (prev != NONE && chg[prev].max_op > c.max_op) return bad(HISTORY_INVALID, "maxOp must increase monotonically per actor");
      c.prev_same_actor = prev;
      last_of[c.actor] = (uint32_t)chg.size();
      chg.push_back(std::move(c));
      if (chg.size() > 0x7ffffff0u) return bad(HISTORY_UNSUPPORTED, "too many changes");
    }
    meta.n_deps = (uint32_t)n_deps;
    lap("change metadata");
    // ---- 2. (device, am355_hist.hip) slots: one bit per (actor, counter) up to the actor's last maxOp, in 32-bit words ----
    meta.act_max.assign(NA, 0);
    for (uint32_t a = 0; a < NA; a++) meta.act_max[a] = last_of[a] == NONE ? 0 : (uint32_t)chg[last_of[a]].max_op;
    meta.word_base.assign((size_t)NA + 1, 0);
    uint64_t words = 0;
    for (uint32_t a = 0; a < NA; a++) {
      meta.word_base[a] = (uint32_t)words;
      words += ((uint64_t)meta.act_max[a] + 2 + 31) / 32;  // bits 0 .. max + 1
      if (words > (1ull << 27)) return bad(HISTORY_UNSUPPORTED, "actors x operation counters beyond the id index (2^32 ids)");
    }
    meta.word_base[NA] = (uint32_t)words;
  }
  (void)P; (void)N;
  return HISTORY_OK;
}

// Steps 5b - 6 on the host threads: headers and column pieces into place, the hash chain, containers. Every change is written ONCE,
// straight into the arena the caller receives (sizes first, then a prefix sum, then the bytes); with `deflate` the arena of plain
// containers is built the same way and compressed change by change into the one that is returned.
namespace {
inline size_t uleb_len(uint64_t v) { size_t n = 1; while (v >= 0x80) { v >>= 7; n++; } return n; }
inline size_t sleb_len(int64_t v) {
  size_t n = 1;
  for (;; n++) {
    const uint8_t b = (uint8_t)(v & 0x7f);
    v >>= 7;
    if ((v == 0 && !(b & 0x40)) || (v == -1 && (b & 0x40))) return n;
  }
}
inline uint8_t* w_uleb(uint8_t* p, uint64_t v) { while (v >= 0x80) { *p++ = (uint8_t)(v | 0x80); v >>= 7; } *p++ = (uint8_t)v; return p; }
inline uint8_t* w_sleb(uint8_t* p, int64_t v) {
  for (;;) {
    const uint8_t b = (uint8_t)(v & 0x7f);
    v >>= 7;
    if ((v == 0 && !(b & 0x40)) || (v == -1 && (b & 0x40))) { *p++ = b; return p; }
    *p++ = b | 0x80;
  }
}
inline uint8_t* w_bytes(uint8_t* p, const void* src, size_t n) { if (n) memcpy(p, src, n); return p + n; }
}  // namespace

// The dependency index column (depsIndex, delta-coded: columnar.js:945-981 reads it change by change): meta.dep_index[c.dep_first ..
// + c.dep_num) of every change. One sequential decode of one value per dependency edge -- independent of everything the device stages
// need, so doc_changes_impl runs it on a thread of its own beside them.
int history_dependencies(const HistoryInput& in, HistoryMeta& meta, std::string& err) {
  auto bad = [&](int rc, const char* msg) { err = msg; return rc; };
  const std::vector<uint8_t>* col = nullptr;
  for (auto& c : *in.change_columns) if (c.first == 0x43) col = &c.second;
  RleReader r(col, 1);
  meta.dep_index.resize(meta.n_deps);
  uint32_t* out = meta.dep_index.data();
  int64_t abs = 0;
  for (size_t k = 0; k < meta.chg.size(); k++) {
    const ChangeRec& c = meta.chg[k];
    for (uint32_t d = 0; d < c.dep_num; d++) {
      bool nul;
      int64_t dv;
      if (!r.next_num(nul, dv) || nul) return bad(HISTORY_INVALID, "malformed dependency index column");
      abs += dv;
      if (abs < 0 || (uint64_t)abs >= k) return bad(HISTORY_INVALID, "dependency index does not name an earlier change");
      out[c.dep_first + d] = (uint32_t)abs;
    }
  }
  if (!r.done()) return bad(HISTORY_INVALID, "dependency index column has trailing values");
  return HISTORY_OK;
}

int history_finish(const HistoryInput& in, HistoryMeta& meta, const HistoryPieces& pc, bool deflate, const ParallelFor& par, HistoryOutput& out, std::string& err) {
  auto bad = [&](int rc, const char* msg) { err = msg; return rc; };
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_lap = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  history: %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
    t_lap = now;
  };
  const std::vector<std::string>& actors = *in.actors;
  const uint32_t NA = (uint32_t)actors.size();
  std::vector<ChangeRec>& chg = meta.chg;
  const std::vector<uint32_t>& dep_index = meta.dep_index;
  const uint32_t NC = (uint32_t)chg.size();
  const unsigned T = 256;
  const unsigned TE = NC < T ? std::max(NC, 1u) : T;
  static const uint32_t col_id[HISTORY_NCOL] = {0x01, 0x02, 0x11, 0x13, 0x15, 0x34, 0x42, 0x56, 0x57, 0x70, 0x71, 0x73};  // (chldActor 0x61 / chldCtr 0x63: all null = empty = omitted)
  auto col_range = [&](uint32_t k, int q, uint32_t& off, uint32_t& len) {
    const uint32_t* o = pc.col_off + (size_t)q * 2 * (NC + 1);
    off = o[2 * k];
    len = o[2 * k + 1] - o[2 * k];
  };
  // the actors change k mentions besides its author, in id order = rank order (columnar.js:154-157)
  auto others_of = [&](uint32_t k, std::vector<uint32_t>& others) {
    others.clear();
    const uint32_t* bits = pc.abits + (size_t)k * pc.aw;
    for (uint32_t w = 0; w < pc.aw; w++)
      for (uint32_t m = bits[w]; m; m &= m - 1) {
        const uint32_t a = w * 32 + (uint32_t)__builtin_ctz(m);
        if (a != chg[k].actor && a < NA) others.push_back(a);
      }
  };
  // ---- sizes: container = magic 4 | checksum 4 | type 1 | LEB(body) | body; body = deps (count, hashes) | rest (header, columns, extra) ----
  std::vector<uint64_t> body_len(NC), plain_off((size_t)NC + 1, 0);
  par(TE, [&](unsigned t) {
    std::vector<uint32_t> others;
    for (uint32_t k = t; k < NC; k += TE) {
      ChangeRec& c = chg[k];
      c.n_ops = pc.chg_nops[k];
      c.start_op = c.max_op - c.n_ops + 1;
      others_of(k, others);
      const std::string& author = actors[c.actor];
      uint64_t n = uleb_len(c.dep_num) + 32ull * c.dep_num;
      n += uleb_len(author.size()) + author.size() + uleb_len(c.seq) + uleb_len(c.start_op) + sleb_len(c.time) + uleb_len(c.message.size()) + c.message.size();
      n += uleb_len(others.size());
      for (uint32_t a : others) n += uleb_len(actors[a].size()) + actors[a].size();
      size_t ncols = 0;
      for (int q = 0; q < HISTORY_NCOL; q++) {
        uint32_t off, len;
        col_range(k, q, off, len);
        if (len) { ncols++; n += uleb_len(col_id[q]) + uleb_len(len) + len; }
      }
      n += uleb_len(ncols) + c.extra.size();
      body_len[k] = n;
    }
  });
  for (uint32_t k = 0; k < NC; k++) plain_off[k + 1] = plain_off[k] + 9 + uleb_len(body_len[k]) + body_len[k];
  std::vector<uint8_t>& plain = out.arena;   // (the result itself unless the changes are to be compressed)
  plain.resize(plain_off[NC]);
  // ---- 5b. every change: header (the dependency hashes are filled in by the chain below) + its stretch of each encoded column ----
  std::vector<uint32_t> deps_at(NC);   // offset of the dependency hashes inside the arena
  par(TE, [&](unsigned t) {
    static const uint8_t MAGIC[4] = {0x85, 0x6f, 0x4a, 0x83};
    std::vector<uint32_t> others;
    for (uint32_t k = t; k < NC; k += TE) {
      const ChangeRec& c = chg[k];
      others_of(k, others);
      uint8_t* p = plain.data() + plain_off[k];
      p = w_bytes(p, MAGIC, 4);
      p += 4;   // checksum: the first four bytes of the hash
      *p++ = 1;
      p = w_uleb(p, body_len[k]);
      p = w_uleb(p, c.dep_num);
      deps_at[k] = (uint32_t)(p - (plain.data() + plain_off[k]));
      p += 32ull * c.dep_num;
      const std::string& author = actors[c.actor];
      p = w_uleb(p, author.size()); p = w_bytes(p, author.data(), author.size());
      p = w_uleb(p, c.seq);
      p = w_uleb(p, c.start_op);
      p = w_sleb(p, c.time);
      p = w_uleb(p, c.message.size()); p = w_bytes(p, c.message.data(), c.message.size());
      p = w_uleb(p, others.size());
      for (uint32_t a : others) { p = w_uleb(p, actors[a].size()); p = w_bytes(p, actors[a].data(), actors[a].size()); }
      size_t ncols = 0;
      for (int q = 0; q < HISTORY_NCOL; q++) { uint32_t off, len; col_range(k, q, off, len); ncols += len ? 1 : 0; }
      p = w_uleb(p, ncols);
      for (int q = 0; q < HISTORY_NCOL; q++) { uint32_t off, len; col_range(k, q, off, len); if (len) { p = w_uleb(p, col_id[q]); p = w_uleb(p, len); } }
      for (int q = 0; q < HISTORY_NCOL; q++) { uint32_t off, len; col_range(k, q, off, len); p = w_bytes(p, pc.col_bytes[q] + off, len); }
      p = w_bytes(p, c.extra.data(), c.extra.size());
    }
  });
  lap("headers + column pieces");
  // ---- 6. hash chain in document order; heads = hashes nobody depends on ----
  out.hashes.assign((size_t)NC * 32, 0);
  std::vector<uint8_t> is_dep(NC, 0);
  {
    // a change can be hashed once its dependencies are: level = 1 + the highest level among them; the changes of one level are
    // independent (64 per level in a 64-actor round structure, one in a single-author history, which then runs inline)
    std::vector<uint32_t> level(NC, 0), level_first(2, 0), by_level(NC);
    uint32_t n_levels = NC ? 1 : 0;
    for (uint32_t k = 0; k < NC; k++) {
      const ChangeRec& c = chg[k];
      uint32_t l = 0;
      for (uint32_t d = 0; d < c.dep_num; d++) {
        uint32_t j = dep_index[c.dep_first + d];
        if (j >= k) return bad(HISTORY_INVALID, "dependency on a later change");
        l = std::max(l, level[j] + 1);
        is_dep[j] = 1;
      }
      level[k] = l;
      n_levels = std::max(n_levels, l + 1);
    }
    level_first.assign((size_t)n_levels + 1, 0);
    for (uint32_t k = 0; k < NC; k++) level_first[level[k] + 1]++;
    for (uint32_t l = 0; l < n_levels; l++) level_first[l + 1] += level_first[l];
    {
      std::vector<uint32_t> at(level_first.begin(), level_first.end() - 1);
      for (uint32_t k = 0; k < NC; k++) by_level[at[level[k]]++] = k;
    }
    auto hash_change = [&](uint32_t k, std::vector<const uint8_t*>& deps) {
      const ChangeRec& c = chg[k];
      deps.clear();
      for (uint32_t d = 0; d < c.dep_num; d++) deps.push_back(&out.hashes[(size_t)dep_index[c.dep_first + d] * 32]);
      std::sort(deps.begin(), deps.end(), [](const uint8_t* x, const uint8_t* y) { return memcmp(x, y, 32) < 0; });
      uint8_t* base = plain.data() + plain_off[k];
      uint8_t* p = base + deps_at[k];
      for (const uint8_t* h : deps) { memcpy(p, h, 32); p += 32; }
      uint8_t* digest = &out.hashes[(size_t)k * 32];
      sha256_digest(base + 8, (size_t)(plain_off[k + 1] - plain_off[k] - 8), digest);   // over [chunk type][LEB length][body]
      memcpy(base + 4, digest, 4);
    };
    // ONE run of the pool for all levels: a task takes changes of the current level until none is left, waits for the ones others
    // took, and goes on to the next level. (A task only ever waits for changes some RUNNING task holds, so the pool may execute the
    // tasks on fewer threads than there are tasks. A run of the pool per level cost a wake-up per level: 5.5 ms for the 64 levels of
    // the headline log against 0.4 ms of hashing.)
    std::vector<uint32_t> taken((size_t)n_levels + 1, 0), finished((size_t)n_levels + 1, 0);
    uint32_t widest = 0;
    for (uint32_t l = 0; l < n_levels; l++) widest = std::max(widest, level_first[l + 1] - level_first[l]);
    const unsigned tasks = std::max(1u, std::min<uint32_t>(widest, 32));
    par(tasks, [&](unsigned) {
      std::vector<const uint8_t*> deps;
      for (uint32_t l = 0; l < n_levels; l++) {
        const uint32_t lo = level_first[l], cnt = level_first[l + 1] - lo;
        for (;;) {
          const uint32_t i = __atomic_fetch_add(&taken[l], 1u, __ATOMIC_RELAXED);
          if (i >= cnt) break;
          hash_change(by_level[lo + i], deps);
          __atomic_fetch_add(&finished[l], 1u, __ATOMIC_RELEASE);
        }
        while (__atomic_load_n(&finished[l], __ATOMIC_ACQUIRE) < cnt) {
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
      }
    });
  }
  {
    std::vector<const uint8_t*> heads;
    for (uint32_t k = 0; k < NC; k++) if (!is_dep[k]) heads.push_back(&out.hashes[(size_t)k * 32]);
    std::sort(heads.begin(), heads.end(), [](const uint8_t* x, const uint8_t* y) { return memcmp(x, y, 32) < 0; });
    heads.erase(std::unique(heads.begin(), heads.end(), [](const uint8_t* x, const uint8_t* y) { return memcmp(x, y, 32) == 0; }), heads.end());
    bool same = heads.size() == in.n_heads;
    for (size_t i = 0; same && i < heads.size(); i++) same = memcmp(heads[i], in.heads + 32 * i, 32) == 0;
    if (!same) return bad(HISTORY_INVALID, "Mismatched heads hashes");
  }
  lap("hash chain + heads");
  out.offsets.assign(plain_off.begin(), plain_off.end());
  if (!deflate) return HISTORY_OK;
  // ---- DEFLATE of the chunk data of changes of >= 256 bytes (columnar.js:798-811), in parallel; then one arena again ----
  std::vector<Bytes> packed(NC);
  std::vector<int> pack_rc(NC, 0);
  par(TE, [&](unsigned t) {
    for (uint32_t k = t; k < NC; k += TE) {
      const uint8_t* full = plain.data() + plain_off[k] + 8;   // [type][length][body]
      const size_t full_len = (size_t)(plain_off[k + 1] - plain_off[k] - 8);
      if (8 + full_len < 256) continue;   // stays as it is
      size_t hdr = 1;  // chunk data = everything after [type][length]
      while (full[hdr] & 0x80) hdr++;
      hdr++;
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { pack_rc[k] = 1; continue; }
      Bytes z(deflateBound(&zs, (uLong)(full_len - hdr)) + 64);
      zs.next_in = const_cast<uint8_t*>(full + hdr); zs.avail_in = (uInt)(full_len - hdr);
      zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
      int zr = ::deflate(&zs, Z_FINISH);
      size_t zn = zs.total_out;
      deflateEnd(&zs);
      if (zr != Z_STREAM_END) { pack_rc[k] = 1; continue; }
      Bytes& o = packed[k];
      o.reserve(16 + zn);
      o.insert(o.end(), plain.data() + plain_off[k], plain.data() + plain_off[k] + 8);
      o.push_back(2);
      put_uleb(o, zn);
      o.insert(o.end(), z.begin(), z.begin() + zn);
    }
  });
  lap("containers");
  for (int rc : pack_rc) if (rc) return bad(HISTORY_UNSUPPORTED, "deflate failed");
  for (uint32_t k = 0; k < NC; k++) out.offsets[k + 1] = out.offsets[k] + (packed[k].empty() ? plain_off[k + 1] - plain_off[k] : packed[k].size());
  std::vector<uint8_t> arena(out.offsets[NC]);
  par(TE, [&](unsigned t) {
    for (uint32_t k = t; k < NC; k += TE) {
      if (packed[k].empty()) memcpy(&arena[out.offsets[k]], plain.data() + plain_off[k], (size_t)(plain_off[k + 1] - plain_off[k]));
      else memcpy(&arena[out.offsets[k]], packed[k].data(), packed[k].size());
    }
  });
  out.arena.swap(arena);
  lap("concatenate");
  return HISTORY_OK;
}

}  // namespace am355
