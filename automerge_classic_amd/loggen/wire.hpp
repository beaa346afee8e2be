// Host-side writer for Automerge's binary change format (the wire format the replay engine consumes).
//
// This is the log *generator* side: it produces byte-identical output to the reference encoder
// (reference: backend/columnar.js:710-739 encodeChange, :370-444 encodeOps, :659-686 encodeContainer;
// backend/encoding.js:558-783 RLEEncoder, :932-948 DeltaEncoder, :1061-1135 BooleanEncoder) so that
// synthetic change logs are exactly what a JS frontend would have shipped. It is written batch-style
// (whole column -> run detection -> bytes) rather than as the reference's streaming state machines.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

namespace amlog {

using Bytes = std::vector<uint8_t>;

// ---- SHA-256 (FIPS 180-4) -------------------------------------------------------------------------
struct Sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t total = 0;
  uint32_t fill = 0;
  Sha256();
  void update(const uint8_t* p, size_t n);
  void digest(uint8_t out[32]);
 private:
  void block(const uint8_t* p);
};

// ---- LEB128 ---------------------------------------------------------------------------------------
inline void put_uleb(Bytes& b, uint64_t v) {
  do {
    uint8_t x = v & 0x7f;
    v >>= 7;
    if (v) x |= 0x80;
    b.push_back(x);
  } while (v);
}
inline void put_sleb(Bytes& b, int64_t v) {
  for (;;) {
    uint8_t x = v & 0x7f;
    int64_t rest = v >> 7;  // arithmetic shift
    bool done = (rest == 0 && !(x & 0x40)) || (rest == -1 && (x & 0x40));
    if (!done) x |= 0x80;
    b.push_back(x);
    if (done) break;
    v = rest;
  }
}
inline size_t sleb_len(int64_t v) { Bytes t; put_sleb(t, v); return t.size(); }

// ---- column value model ----------------------------------------------------------------------------
// A nullable integer column value.
struct OptInt {
  bool null;
  int64_t v;
  static OptInt none() { return {true, 0}; }
  static OptInt of(int64_t x) { return {false, x}; }
  bool operator==(const OptInt& o) const { return null == o.null && (null || v == o.v); }
};
struct OptStr {
  bool null;
  std::string s;
  bool operator==(const OptStr& o) const { return null == o.null && (null || s == o.s); }
};

// RLE over nullable values: runs of nulls -> (0, n); runs of >=2 equal values -> (n, v);
// maximal stretches of lone values -> (-k, v1..vk). A column holding only nulls encodes to nothing.
template <class T, class PutValue>
void rle_encode(Bytes& out, const std::vector<T>& vals, PutValue put) {
  size_t n = vals.size(), i = 0;
  bool wrote = false;
  std::vector<const T*> lit;
  auto flush_lit = [&]() {
    if (lit.empty()) return;
    put_sleb(out, -(int64_t)lit.size());
    for (auto* p : lit) put(out, *p);
    lit.clear();
    wrote = true;
  };
  while (i < n) {
    size_t j = i + 1;
    while (j < n && vals[j] == vals[i]) j++;
    size_t run = j - i;
    if (vals[i].null) {
      flush_lit();
      if (j == n && !wrote) break;  // only nulls seen so far and nothing follows: omit
      put_sleb(out, 0);
      put_uleb(out, run);
      wrote = true;
    } else if (run >= 2) {
      flush_lit();
      put_sleb(out, (int64_t)run);
      put(out, vals[i]);
      wrote = true;
    } else {
      lit.push_back(&vals[i]);
    }
    i = j;
  }
  flush_lit();
}

inline void rle_uint(Bytes& out, const std::vector<OptInt>& v) {
  rle_encode(out, v, [](Bytes& b, const OptInt& x) { put_uleb(b, (uint64_t)x.v); });
}
inline void rle_int(Bytes& out, const std::vector<OptInt>& v) {
  rle_encode(out, v, [](Bytes& b, const OptInt& x) { put_sleb(b, x.v); });
}
inline void rle_utf8(Bytes& out, const std::vector<OptStr>& v) {
  rle_encode(out, v, [](Bytes& b, const OptStr& x) {
    put_uleb(b, x.s.size());
    b.insert(b.end(), x.s.begin(), x.s.end());
  });
}
// Delta: RLE-int over successive differences of the non-null values (running value starts at 0).
inline void delta_encode(Bytes& out, const std::vector<OptInt>& v) {
  std::vector<OptInt> d(v.size());
  int64_t abs = 0;
  for (size_t i = 0; i < v.size(); i++) {
    if (v[i].null) d[i] = OptInt::none();
    else { d[i] = OptInt::of(v[i].v - abs); abs = v[i].v; }
  }
  rle_int(out, d);
}
// Boolean: alternating run lengths, first run counts `false`.
inline void bool_encode(Bytes& out, const std::vector<uint8_t>& v) {
  if (v.empty()) return;
  uint8_t cur = 0;
  uint64_t cnt = 0;
  for (uint8_t x : v) {
    if ((x != 0) == (cur != 0)) cnt++;
    else { put_uleb(out, cnt); cur = x; cnt = 1; }
  }
  if (cnt > 0) put_uleb(out, cnt);
}

// ---- operations -------------------------------------------------------------------------------------
enum Action : uint32_t { MAKE_MAP = 0, SET = 1, MAKE_LIST = 2, DEL = 3, MAKE_TEXT = 4, INC = 5, MAKE_TABLE = 6, LINK = 7 };
enum ValType : uint32_t { V_NULL = 0, V_FALSE = 1, V_TRUE = 2, V_UINT = 3, V_INT = 4, V_F64 = 5, V_UTF8 = 6, V_BYTES = 7, V_COUNTER = 8, V_TIMESTAMP = 9 };

// Operation id: (counter, index into the generator's global actor table). ctr==0 means "none".
struct Id {
  uint64_t ctr = 0;
  uint32_t actor = 0;
};

struct Op {
  Id obj;              // ctr==0 -> _root
  bool has_key = false;
  std::string key;     // map key, when has_key
  Id elem;             // list element reference; ctr==0 with insert -> _head
  bool insert = false;
  uint32_t action = SET;
  uint32_t vtype = V_NULL;
  int64_t ival = 0;    // V_UINT / V_INT / V_COUNTER / V_TIMESTAMP
  double fval = 0;     // V_F64
  std::string sval;    // V_UTF8 / V_BYTES
  std::vector<Id> pred;
};

struct Change {
  uint32_t actor = 0;  // index into the global actor table
  uint64_t seq = 1, start_op = 1;
  int64_t time = 0;
  std::string message;
  std::vector<std::array<uint8_t, 32>> deps;
  std::vector<Op> ops;
};

struct Encoded {
  Bytes bytes;                 // container as shipped (possibly chunk type 2 = DEFLATE)
  std::array<uint8_t, 32> hash;
  size_t raw_len;              // length of the uncompressed (chunk type 1) container
};

// actors: global table of raw actor-id bytes. deflate: mimic encodeChange's DEFLATE of changes >= 256 bytes.
Encoded encode_change(const Change& c, const std::vector<Bytes>& actors, bool deflate);

}  // namespace amlog
