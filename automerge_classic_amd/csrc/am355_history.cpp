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

struct ChangeRec {
  uint32_t actor = 0;  // rank
  uint64_t seq = 0, max_op = 0, start_op = 0;
  int64_t time = 0;
  std::string message, extra;
  uint32_t dep_first = 0, dep_num = 0;
  uint32_t op_base = 0, n_ops = 0;  // slot range
  uint32_t prev_same_actor = NONE;
};

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

struct Built {      // one change after step 5
  Bytes rest;       // everything after the dependency hashes: actor .. columns .. extra bytes
  int rc = 0;
  const char* why = nullptr;
};

}  // namespace

int reconstruct_history(const HistoryInput& in, bool deflate, const ParallelFor& par, HistoryOutput& out, std::string& err) {
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

  // the reference reads rows until every column is exhausted (columnar.js:577-590): the key column must hold N values or none,
  // and the value bytes must be consumed exactly (a longer valRaw column makes extra rows there too)
  {
    RleReader keys(nullptr, 2);
    keys.r.p = in.key_column; keys.r.len = in.key_column_len;
    uint64_t n_keys = 0;
    while (!keys.done()) {
      if (!keys.skip_run(n_keys)) return bad(HISTORY_INVALID, "malformed key column");
      if (n_keys > (uint64_t)N) break;
    }
    if (n_keys != 0 && n_keys != N) return bad(HISTORY_UNSUPPORTED, "key column and action column differ in length: the JS path decides");
    uint64_t val_bytes = 0;
    for (uint32_t r = 0; r < N; r++) val_bytes += in.val_tl[r] >> 4;
    if (val_bytes != in.val_raw_len) return bad(HISTORY_UNSUPPORTED, "value bytes do not cover the valRaw column: the JS path decides");
  }
  lap("column lengths");
  // ---- 1. change metadata ----
  std::vector<ChangeRec> chg;
  std::vector<uint32_t> dep_index;
  {
    const std::vector<uint8_t>* col[9] = {};
    static const uint32_t ids[9] = {0x01, 0x03, 0x13, 0x23, 0x35, 0x40, 0x43, 0x56, 0x57};
    for (auto& c : *in.change_columns) {
      int k = -1;
      for (int i = 0; i < 9; i++) if (ids[i] == c.first) k = i;
      if (k < 0) { if (!c.second.empty()) return bad(HISTORY_UNSUPPORTED, "document has change columns this engine does not model"); continue; }
      col[k] = &c.second;
    }
    RleReader r_actor(col[0], 0), r_seq(col[1], 1), r_max(col[2], 1), r_time(col[3], 1), r_msg(col[4], 2), r_dnum(col[5], 0), r_didx(col[6], 1), r_xlen(col[7], 0);
    int64_t seq_abs = 0, max_abs = 0, time_abs = 0, didx_abs = 0;
    size_t xoff = 0;
    const size_t xtotal = col[8] ? col[8]->size() : 0;
    std::vector<uint32_t> last_of(NA, NONE);
    while (!(r_actor.done() && r_seq.done() && r_max.done() && r_time.done() && r_msg.done() && r_dnum.done() && r_xlen.done())) {
      ChangeRec c;
      bool nul;
      int64_t v;
      std::string s;
      if (!r_actor.next(nul, v, s)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul || v < 0 || (uint64_t)v >= in.doc_actor_rank->size()) return bad(HISTORY_INVALID, "bad actor index in change metadata");
      c.actor = (*in.doc_actor_rank)[(size_t)v];
      if (!r_seq.next(nul, v, s)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul) return bad(HISTORY_INVALID, "change without seq");
      seq_abs += v; c.seq = (uint64_t)seq_abs;
      if (!r_max.next(nul, v, s)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul) return bad(HISTORY_INVALID, "change without maxOp");
      max_abs += v;
      if (max_abs < 0 || max_abs > 0x7fffffff) return bad(HISTORY_UNSUPPORTED, "maxOp beyond 2^31");
      c.max_op = (uint64_t)max_abs;
      if (!r_time.next(nul, v, s)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul) return bad(HISTORY_INVALID, "change without time");
      time_abs += v; c.time = time_abs;
      if (!r_msg.next(nul, v, s)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (!nul) {
        if (!valid_utf8((const uint8_t*)s.data(), s.size())) return bad(HISTORY_UNSUPPORTED, "message is not valid UTF-8");
        c.message = s;
      }
      if (!r_dnum.next(nul, v, s)) return bad(HISTORY_INVALID, "malformed change metadata columns");
      if (nul || v < 0 || v > 0x7fffffff) return bad(HISTORY_INVALID, "bad dependency count");
      c.dep_first = (uint32_t)dep_index.size();
      c.dep_num = (uint32_t)v;
      for (uint32_t d = 0; d < c.dep_num; d++) {
        bool dn;
        int64_t dv;
        if (!r_didx.next(dn, dv, s) || dn) return bad(HISTORY_INVALID, "malformed dependency index column");
        didx_abs += dv;
        if (didx_abs < 0 || (uint64_t)didx_abs >= chg.size()) return bad(HISTORY_INVALID, "dependency index does not name an earlier change");
        dep_index.push_back((uint32_t)didx_abs);
      }
      if (!r_xlen.next(nul, v, s)) return bad(HISTORY_INVALID, "malformed change metadata columns");
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
    if (!r_didx.done()) return bad(HISTORY_INVALID, "dependency index column has trailing values");
    lap("change metadata");
    // ---- 2. slots: one bit per (actor, counter) up to the actor's last maxOp ----
    std::vector<uint64_t> act_max(NA, 0);
    for (uint32_t a = 0; a < NA; a++) act_max[a] = last_of[a] == NONE ? 0 : chg[last_of[a]].max_op;
    std::vector<uint64_t> word_base(NA + 1, 0);
    for (uint32_t a = 0; a < NA; a++) word_base[a + 1] = word_base[a] + (act_max[a] + 2 + 63) / 64;  // bits 0 .. max + 1
    const uint64_t W = word_base[NA];
    if (W > (1ull << 26)) return bad(HISTORY_UNSUPPORTED, "actors x operation counters beyond the id index (2^32 ids)");
    std::vector<uint64_t> all_bits((size_t)W + 1, 0), row_bits((size_t)W + 1, 0);
    const unsigned T = 256;
    std::vector<int> task_rc(T, 0);
    auto span = [&](uint32_t n, unsigned t, uint32_t& lo, uint32_t& hi) {
      lo = (uint32_t)((uint64_t)n * t / T);
      hi = (uint32_t)((uint64_t)n * (t + 1) / T);
    };
    par(T, [&](unsigned t) {
      uint32_t lo, hi;
      span(N, t, lo, hi);
      for (uint32_t r = lo; r < hi; r++) {
        uint32_t a = in.id_actor[r], ctr = in.id_ctr[r];
        if (a >= NA || ctr == 0 || ctr > act_max[a]) { task_rc[t] = 1; continue; }
        uint64_t w = word_base[a] + ctr / 64, bit = 1ull << (ctr % 64);
        if (atomic_or_old(&row_bits[w], bit) & bit) task_rc[t] = 2;
        atomic_or(&all_bits[w], bit);
      }
      span(P, t, lo, hi);
      for (uint32_t e = lo; e < hi; e++) {
        uint32_t a = in.succ_actor[e], ctr = in.succ_ctr[e];
        if (a >= NA || ctr == 0 || ctr > act_max[a]) { task_rc[t] = 1; continue; }
        atomic_or(&all_bits[word_base[a] + ctr / 64], 1ull << (ctr % 64));
      }
    });
    for (int rc : task_rc) {
      if (rc == 1) return bad(HISTORY_INVALID, "operation id outside of the range its actor's changes allow");
      if (rc == 2) return bad(HISTORY_INVALID, "two rows carry one operation id");
    }
    std::vector<uint32_t> word_rank((size_t)W + 1, 0);
    {
      uint64_t acc = 0;
      for (uint64_t w = 0; w < W; w++) { word_rank[w] = (uint32_t)acc; acc += (uint64_t)__builtin_popcountll(all_bits[w]); }
      if (acc >= 0xfffffff0ull) return bad(HISTORY_UNSUPPORTED, "more than 2^32 operations");
      word_rank[W] = (uint32_t)acc;
    }
    lap("id bitmaps + rank directory");
    const uint32_t M = word_rank[W];
    // number of ids of actor a with counter < ctr, plus the actor's slot base = the slot of (a, ctr) when that id exists
    auto slot_of = [&](uint32_t a, uint64_t ctr) -> uint32_t {
      uint64_t w = word_base[a] + ctr / 64;
      return word_rank[w] + (uint32_t)__builtin_popcountll(all_bits[w] & ((1ull << (ctr % 64)) - 1));
    };
    // ---- 3. rows, deletion witnesses and preds by slot ----
    std::vector<uint32_t> slot_row(M, NONE), slot_ref(M, NONE), pred_first((size_t)M + 1, 0);
    par(T, [&](unsigned t) {
      uint32_t lo, hi;
      span(N, t, lo, hi);
      for (uint32_t r = lo; r < hi; r++) {
        slot_row[slot_of(in.id_actor[r], in.id_ctr[r])] = r;
        uint32_t f = in.succ_first[r], n = in.succ_num[r];
        if ((uint64_t)f + n > P) { task_rc[t] = 1; continue; }
        for (uint32_t e = f; e < f + n; e++) {
          uint32_t s = slot_of(in.succ_actor[e], in.succ_ctr[e]);
          atomic_min(&slot_ref[s], r);
          __atomic_fetch_add(&pred_first[s + 1], 1u, __ATOMIC_RELAXED);
        }
      }
    });
    for (int rc : task_rc) if (rc) return bad(HISTORY_INVALID, "succ lists exceed the succ columns");
    for (uint32_t s = 0; s < M; s++) pred_first[s + 1] += pred_first[s];
    std::vector<uint32_t> pred_row(pred_first[M]), cursor(M, 0);
    par(T, [&](unsigned t) {
      uint32_t lo, hi;
      span(N, t, lo, hi);
      for (uint32_t r = lo; r < hi; r++) {
        uint32_t f = in.succ_first[r], n = in.succ_num[r];
        for (uint32_t e = f; e < f + n; e++) {
          uint32_t s = slot_of(in.succ_actor[e], in.succ_ctr[e]);
          pred_row[pred_first[s] + __atomic_fetch_add(&cursor[s], 1u, __ATOMIC_RELAXED)] = r;
        }
      }
    });
    lap("rows and preds by slot");
    // ---- 4. changes -> slot ranges ----
    for (ChangeRec& c : chg) {
      uint64_t prev_max = c.prev_same_actor == NONE ? 0 : chg[c.prev_same_actor].max_op;
      c.op_base = slot_of(c.actor, prev_max + 1);
      c.n_ops = slot_of(c.actor, c.max_op + 1) - c.op_base;
      if (c.n_ops > c.max_op) return bad(HISTORY_INVALID, "more operations than maxOp allows");
      c.start_op = c.max_op - c.n_ops + 1;
      // ids must be startOp .. maxOp without a gap (columnar.js:935-939)
      if (c.n_ops && slot_of(c.actor, c.start_op) != c.op_base) return bad(HISTORY_INVALID, "operation ids of a change are not consecutive");
    }
    // ---- 5. encode every change (without its dependency hashes) ----
    const uint32_t NC = (uint32_t)chg.size();
    std::vector<Built> built(NC);
    // (changes vary from a few ops to millions: tasks take them round-robin)
    const unsigned TE = NC < T ? std::max(NC, 1u) : T;
    par(TE, [&](unsigned t) {
      std::vector<uint32_t> local(NA, NONE), touched;
      std::vector<OptInt> objActor, objCtr, keyActor, keyCtr, action, valLen, predNum, predActor, predCtr;
      std::vector<OptStr> keyStr;
      std::vector<uint8_t> insert;
      std::vector<std::pair<uint32_t, uint32_t>> preds;  // (ctr, actor rank)
      Bytes valRaw;
      for (uint32_t k = t; k < NC; k += TE) {
        const ChangeRec& c = chg[k];
        Built& b = built[k];
        auto fail = [&](int rc, const char* why) { b.rc = rc; b.why = why; };
        // referenced actors: author first, the others in id order = rank order (columnar.js:154-157)
        touched.clear();
        auto touch = [&](uint32_t a) { if (local[a] == NONE) { local[a] = 0; touched.push_back(a); } };
        touch(c.actor);
        const uint32_t n = c.n_ops;
        struct OpSrc { uint32_t row; bool del; };
        for (uint32_t i = 0; i < n && !b.rc; i++) {
          uint32_t s = c.op_base + i, r = slot_row[s];
          uint32_t q = r != NONE ? r : slot_ref[s];
          if (q == NONE) { fail(HISTORY_INVALID, "operation without a row"); break; }
          if (in.obj_actor[q] != NONE) { if (in.obj_actor[q] >= NA) { fail(HISTORY_INVALID, "bad actor"); break; } touch(in.obj_actor[q]); }
          if (in.key_len[q] == NONE) {
            if (r == NONE && in.insert[q]) touch(in.id_actor[q]);
            else if (in.key_ctr[q] != 0 && in.key_ctr[q] != NONE) { if (in.key_actor[q] >= NA) { fail(HISTORY_INVALID, "bad actor"); break; } touch(in.key_actor[q]); }
          }
          for (uint32_t e = pred_first[s]; e < pred_first[s + 1]; e++) touch(in.id_actor[pred_row[e]]);
        }
        if (b.rc) { for (uint32_t a : touched) local[a] = NONE; continue; }
        std::sort(touched.begin() + 1, touched.end());
        for (uint32_t i = 0; i < touched.size(); i++) local[touched[i]] = i;
        objActor.assign(n, OptInt::none()); objCtr.assign(n, OptInt::none()); keyActor.assign(n, OptInt::none()); keyCtr.assign(n, OptInt::none());
        action.resize(n); valLen.resize(n); predNum.resize(n); predActor.clear(); predCtr.clear();
        keyStr.assign(n, OptStr{true, std::string()});
        insert.assign(n, 0);
        valRaw.clear();
        for (uint32_t i = 0; i < n && !b.rc; i++) {
          uint32_t s = c.op_base + i, r = slot_row[s];
          bool del = r == NONE;
          uint32_t q = del ? slot_ref[s] : r;
          if (in.obj_actor[q] != NONE) { objActor[i] = OptInt::of(local[in.obj_actor[q]]); objCtr[i] = OptInt::of(in.obj_ctr[q]); }
          if (in.key_len[q] != NONE) {
            if (in.key_len[q] == 0) { fail(HISTORY_UNSUPPORTED, "empty map key"); break; }
            if ((uint64_t)in.key_off[q] + in.key_len[q] > in.arena_len) { fail(HISTORY_INVALID, "key outside the arena"); break; }
            if (!valid_utf8(in.arena + in.key_off[q], in.key_len[q])) { fail(HISTORY_UNSUPPORTED, "key is not valid UTF-8"); break; }
            keyStr[i] = OptStr{false, std::string((const char*)in.arena + in.key_off[q], in.key_len[q])};
          } else if (del && in.insert[q]) {  // deleting the element the witness row inserted
            keyActor[i] = OptInt::of(local[in.id_actor[q]]); keyCtr[i] = OptInt::of(in.id_ctr[q]);
          } else if (in.key_ctr[q] == 0) {
            if (del || !in.insert[q]) { fail(HISTORY_INVALID, "operation on _head that is not an insertion"); break; }
            keyCtr[i] = OptInt::of(0);
          } else if (in.key_ctr[q] != NONE) {
            keyActor[i] = OptInt::of(local[in.key_actor[q]]); keyCtr[i] = OptInt::of(in.key_ctr[q]);
          } else { fail(HISTORY_INVALID, "operation without a key"); break; }
          uint32_t act = del ? 3u : in.action[q];
          if (!del && act == 3) { fail(HISTORY_INVALID, "document should not contain del operations"); break; }
          if (act >= 7) { fail(HISTORY_UNSUPPORTED, "link or unknown action"); break; }
          insert[i] = del ? 0 : in.insert[q];
          action[i] = OptInt::of(act);
          uint32_t tl = 0;
          if (!del && (act == 1 || act == 5)) {
            tl = in.val_tl[q];
            uint32_t len = tl >> 4;
            if (len && (uint64_t)in.val_off[q] + len > in.arena_len) { fail(HISTORY_INVALID, "value outside the arena"); break; }
            int vr = value_round_trips(tl, in.arena + in.val_off[q]);
            if (vr) { fail(vr, "value the reference does not re-encode byte for byte"); break; }
            valRaw.insert(valRaw.end(), in.arena + in.val_off[q], in.arena + in.val_off[q] + len);
          }
          valLen[i] = OptInt::of(tl);
          preds.clear();
          for (uint32_t e = pred_first[s]; e < pred_first[s + 1]; e++) preds.emplace_back(in.id_ctr[pred_row[e]], in.id_actor[pred_row[e]]);
          std::sort(preds.begin(), preds.end());
          predNum[i] = OptInt::of((int64_t)preds.size());
          for (auto& p : preds) { predActor.push_back(OptInt::of(local[p.second])); predCtr.push_back(OptInt::of(p.first)); }
        }
        if (!b.rc) {
          struct Col { uint32_t id; Bytes data; };
          Col cols[12];
          cols[0].id = 0x01; amlog::rle_uint(cols[0].data, objActor);
          cols[1].id = 0x02; amlog::rle_uint(cols[1].data, objCtr);
          cols[2].id = 0x11; amlog::rle_uint(cols[2].data, keyActor);
          cols[3].id = 0x13; amlog::delta_encode(cols[3].data, keyCtr);
          cols[4].id = 0x15; amlog::rle_utf8(cols[4].data, keyStr);
          cols[5].id = 0x34; amlog::bool_encode(cols[5].data, insert);
          cols[6].id = 0x42; amlog::rle_uint(cols[6].data, action);
          cols[7].id = 0x56; amlog::rle_uint(cols[7].data, valLen);
          cols[8].id = 0x57; cols[8].data.swap(valRaw);
          cols[9].id = 0x70; amlog::rle_uint(cols[9].data, predNum);   // (chldActor 0x61 / chldCtr 0x63: all null = empty = omitted)
          cols[10].id = 0x71; amlog::rle_uint(cols[10].data, predActor);
          cols[11].id = 0x73; amlog::delta_encode(cols[11].data, predCtr);
          Bytes& o = b.rest;
          const std::string& author = actors[c.actor];
          put_uleb(o, author.size()); o.insert(o.end(), author.begin(), author.end());
          put_uleb(o, c.seq);
          put_uleb(o, c.start_op);
          put_sleb(o, c.time);
          put_uleb(o, c.message.size()); o.insert(o.end(), c.message.begin(), c.message.end());
          put_uleb(o, touched.size() - 1);
          for (size_t i = 1; i < touched.size(); i++) { const std::string& id = actors[touched[i]]; put_uleb(o, id.size()); o.insert(o.end(), id.begin(), id.end()); }
          size_t ncols = 0;
          for (auto& col : cols) ncols += col.data.empty() ? 0 : 1;
          put_uleb(o, ncols);
          for (auto& col : cols) if (!col.data.empty()) { put_uleb(o, col.id); put_uleb(o, col.data.size()); }
          for (auto& col : cols) o.insert(o.end(), col.data.begin(), col.data.end());
          o.insert(o.end(), c.extra.begin(), c.extra.end());
          valRaw.swap(cols[8].data);
        }
        for (uint32_t a : touched) local[a] = NONE;
      }
    });
    lap("encode changes");
    {  // the first failure in document order is the one a sequential reader meets
      for (uint32_t k = 0; k < NC; k++) if (built[k].rc == HISTORY_INVALID) return bad(HISTORY_INVALID, built[k].why);
      for (uint32_t k = 0; k < NC; k++) if (built[k].rc) return bad(built[k].rc, built[k].why);
    }
    // ---- 6. hash chain in document order; heads = hashes nobody depends on ----
    out.hashes.assign((size_t)NC * 32, 0);
    std::vector<Bytes> plain(NC);  // [chunk type 1][LEB length][dependency count][hashes][rest]: what the hash covers
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
      auto hash_change = [&](uint32_t k, std::vector<const uint8_t*>& deps, Bytes& dp) {
        const ChangeRec& c = chg[k];
        deps.clear();
        for (uint32_t d = 0; d < c.dep_num; d++) deps.push_back(&out.hashes[(size_t)dep_index[c.dep_first + d] * 32]);
        std::sort(deps.begin(), deps.end(), [](const uint8_t* x, const uint8_t* y) { return memcmp(x, y, 32) < 0; });
        dp.clear();
        put_uleb(dp, deps.size());
        for (const uint8_t* h : deps) dp.insert(dp.end(), h, h + 32);
        Bytes& full = plain[k];
        Bytes& rest = built[k].rest;
        full.reserve(12 + dp.size() + rest.size());
        full.push_back(1);
        put_uleb(full, dp.size() + rest.size());
        full.insert(full.end(), dp.begin(), dp.end());
        full.insert(full.end(), rest.begin(), rest.end());
        Bytes().swap(rest);
        sha256_digest(full.data(), full.size(), &out.hashes[(size_t)k * 32]);
      };
      std::vector<const uint8_t*> deps0;
      Bytes dp0;
      for (uint32_t l = 0; l < n_levels; l++) {
        const uint32_t lo = level_first[l], hi = level_first[l + 1], cnt = hi - lo;
        if (cnt < 8) {
          for (uint32_t i = lo; i < hi; i++) hash_change(by_level[i], deps0, dp0);
        } else {
          const unsigned tasks = std::min<uint32_t>(cnt, 32);
          par(tasks, [&](unsigned t) {
            std::vector<const uint8_t*> deps;
            Bytes dp;
            for (uint32_t i = lo + t; i < hi; i += tasks) hash_change(by_level[i], deps, dp);
          });
        }
      }
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
    // ---- containers (+ DEFLATE of the chunk data of changes of >= 256 bytes), in parallel ----
    std::vector<Bytes> packed(NC);
    std::vector<int> pack_rc(NC, 0);
    par(TE, [&](unsigned t) {
      static const uint8_t MAGIC[4] = {0x85, 0x6f, 0x4a, 0x83};
      for (uint32_t k = t; k < NC; k += TE) {
        Bytes& o = packed[k];
        const Bytes& full = plain[k];
        o.reserve(8 + full.size());
        o.insert(o.end(), MAGIC, MAGIC + 4);
        o.insert(o.end(), &out.hashes[(size_t)k * 32], &out.hashes[(size_t)k * 32] + 4);
        if (deflate && 8 + full.size() >= 256) {
          size_t hdr = 1;  // chunk data = everything after [type][length]
          while (full[hdr] & 0x80) hdr++;
          hdr++;
          z_stream zs;
          memset(&zs, 0, sizeof zs);
          if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { pack_rc[k] = 1; continue; }
          Bytes z(deflateBound(&zs, (uLong)(full.size() - hdr)) + 64);
          zs.next_in = const_cast<uint8_t*>(full.data() + hdr); zs.avail_in = (uInt)(full.size() - hdr);
          zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
          int zr = ::deflate(&zs, Z_FINISH);
          size_t zn = zs.total_out;
          deflateEnd(&zs);
          if (zr != Z_STREAM_END) { pack_rc[k] = 1; continue; }
          o.push_back(2);
          put_uleb(o, zn);
          o.insert(o.end(), z.begin(), z.begin() + zn);
        } else {
          o.insert(o.end(), full.begin(), full.end());
        }
        Bytes().swap(plain[k]);
      }
    });
    lap("containers");
    for (int rc : pack_rc) if (rc) return bad(HISTORY_UNSUPPORTED, "deflate failed");
    out.offsets.assign((size_t)NC + 1, 0);
    for (uint32_t k = 0; k < NC; k++) out.offsets[k + 1] = out.offsets[k] + packed[k].size();
    out.arena.resize(out.offsets[NC]);
    par(TE, [&](unsigned t) {
      for (uint32_t k = t; k < NC; k += TE) if (!packed[k].empty()) memcpy(&out.arena[out.offsets[k]], packed[k].data(), packed[k].size());
    });
  }
  lap("concatenate");
  return HISTORY_OK;
}

}  // namespace am355
