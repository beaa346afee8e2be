// See wire.hpp. Binary change writer used by the synthetic log generator.
#include <array>
#include "wire.hpp"
#include <zlib.h>
#include <stdexcept>

namespace amlog {

// ---- SHA-256 ----------------------------------------------------------------------------------------
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

Sha256::Sha256() {
  static const uint32_t init[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(h, init, sizeof(h));
}

void Sha256::block(const uint8_t* p) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
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

void Sha256::update(const uint8_t* p, size_t n) {
  total += n;
  while (n) {
    if (fill == 0 && n >= 64) { block(p); p += 64; n -= 64; continue; }
    size_t take = std::min<size_t>(64 - fill, n);
    memcpy(buf + fill, p, take);
    fill += take; p += take; n -= take;
    if (fill == 64) { block(buf); fill = 0; }
  }
}

void Sha256::digest(uint8_t out[32]) {
  uint64_t bits = total * 8;
  uint8_t pad[72] = {0x80};
  size_t padlen = (fill < 56) ? 56 - fill : 120 - fill;
  uint8_t len[8];
  for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
  update(pad, padlen);
  update(len, 8);
  for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
}

// ---- change encoding ----------------------------------------------------------------------------------
static int cmp_bytes(const Bytes& a, const Bytes& b) {
  size_t n = std::min(a.size(), b.size());
  int c = n ? memcmp(a.data(), b.data(), n) : 0;
  if (c) return c;
  return a.size() < b.size() ? -1 : (a.size() > b.size() ? 1 : 0);
}

Encoded encode_change(const Change& c, const std::vector<Bytes>& actors, bool deflate) {
  // Local actor table: author first, every other referenced actor in lexicographic order
  // (reference columnar.js:154-157).
  std::vector<uint32_t> used;
  auto note = [&](const Id& id) { if (id.ctr) used.push_back(id.actor); };
  for (const Op& op : c.ops) {
    note(op.obj);
    note(op.elem);
    for (const Id& p : op.pred) note(p);
  }
  std::sort(used.begin(), used.end());
  used.erase(std::unique(used.begin(), used.end()), used.end());
  used.erase(std::remove(used.begin(), used.end(), c.actor), used.end());
  std::sort(used.begin(), used.end(), [&](uint32_t x, uint32_t y) { return cmp_bytes(actors[x], actors[y]) < 0; });
  std::vector<uint32_t> local(actors.size(), UINT32_MAX);
  local[c.actor] = 0;
  for (size_t i = 0; i < used.size(); i++) local[used[i]] = (uint32_t)i + 1;

  size_t n = c.ops.size();
  std::vector<OptInt> objActor(n), objCtr(n), keyActor(n), keyCtr(n), action(n), valLen(n), predNum(n), predActor, predCtr;
  std::vector<OptStr> keyStr(n);
  std::vector<uint8_t> insert(n);
  Bytes valRaw;
  for (size_t i = 0; i < n; i++) {
    const Op& op = c.ops[i];
    if (op.obj.ctr) { objActor[i] = OptInt::of(local[op.obj.actor]); objCtr[i] = OptInt::of((int64_t)op.obj.ctr); }
    else { objActor[i] = OptInt::none(); objCtr[i] = OptInt::none(); }
    if (op.has_key) {
      keyActor[i] = OptInt::none(); keyCtr[i] = OptInt::none(); keyStr[i] = {false, op.key};
    } else if (op.elem.ctr == 0) {
      if (!op.insert) throw std::runtime_error("non-insert op must name a list element");
      keyActor[i] = OptInt::none(); keyCtr[i] = OptInt::of(0); keyStr[i] = {true, ""};
    } else {
      keyActor[i] = OptInt::of(local[op.elem.actor]); keyCtr[i] = OptInt::of((int64_t)op.elem.ctr); keyStr[i] = {true, ""};
    }
    insert[i] = op.insert;
    action[i] = OptInt::of(op.action);
    // value (reference columnar.js:258-292): only set/inc carry one
    uint32_t vt = (op.action == SET || op.action == INC) ? op.vtype : V_NULL;
    size_t before = valRaw.size();
    switch (vt) {
      case V_NULL: case V_FALSE: case V_TRUE: break;
      case V_UINT: put_uleb(valRaw, (uint64_t)op.ival); break;
      case V_INT: case V_COUNTER: case V_TIMESTAMP: put_sleb(valRaw, op.ival); break;
      case V_F64: { uint8_t b[8]; memcpy(b, &op.fval, 8); valRaw.insert(valRaw.end(), b, b + 8); break; }
      default: valRaw.insert(valRaw.end(), op.sval.begin(), op.sval.end()); break;
    }
    valLen[i] = OptInt::of((int64_t)(((valRaw.size() - before) << 4) | vt));
    // preds sorted by (counter, actor id) (reference columnar.js:426)
    std::vector<Id> preds = op.pred;
    std::sort(preds.begin(), preds.end(), [&](const Id& a, const Id& b) {
      if (a.ctr != b.ctr) return a.ctr < b.ctr;
      return cmp_bytes(actors[a.actor], actors[b.actor]) < 0;
    });
    predNum[i] = OptInt::of((int64_t)preds.size());
    for (const Id& p : preds) { predActor.push_back(OptInt::of(local[p.actor])); predCtr.push_back(OptInt::of((int64_t)p.ctr)); }
  }

  struct Col { uint32_t id; Bytes data; };
  std::vector<Col> cols(13);
  cols[0].id = 0x01; rle_uint(cols[0].data, objActor);
  cols[1].id = 0x02; rle_uint(cols[1].data, objCtr);
  cols[2].id = 0x11; rle_uint(cols[2].data, keyActor);
  cols[3].id = 0x13; delta_encode(cols[3].data, keyCtr);
  cols[4].id = 0x15; rle_utf8(cols[4].data, keyStr);
  cols[5].id = 0x34; bool_encode(cols[5].data, insert);
  cols[6].id = 0x42; rle_uint(cols[6].data, action);
  cols[7].id = 0x56; rle_uint(cols[7].data, valLen);
  cols[8].id = 0x57; cols[8].data = valRaw;
  cols[9].id = 0x61;   // chldActor: always null for the actions generated here
  cols[10].id = 0x70; rle_uint(cols[10].data, predNum);
  cols[11].id = 0x71; rle_uint(cols[11].data, predActor);
  cols[12].id = 0x73; delta_encode(cols[12].data, predCtr);

  Bytes body;
  put_uleb(body, c.deps.size());
  auto deps = c.deps;
  std::sort(deps.begin(), deps.end());
  for (auto& d : deps) body.insert(body.end(), d.begin(), d.end());
  put_uleb(body, actors[c.actor].size());
  body.insert(body.end(), actors[c.actor].begin(), actors[c.actor].end());
  put_uleb(body, c.seq);
  put_uleb(body, c.start_op);
  put_sleb(body, c.time);
  put_uleb(body, c.message.size());
  body.insert(body.end(), c.message.begin(), c.message.end());
  put_uleb(body, used.size());
  for (uint32_t a : used) { put_uleb(body, actors[a].size()); body.insert(body.end(), actors[a].begin(), actors[a].end()); }
  size_t ncols = 0;
  for (auto& col : cols) if (!col.data.empty()) ncols++;
  put_uleb(body, ncols);
  for (auto& col : cols) if (!col.data.empty()) { put_uleb(body, col.id); put_uleb(body, col.data.size()); }
  for (auto& col : cols) body.insert(body.end(), col.data.begin(), col.data.end());

  Bytes head;  // chunk type + length: the hashed prefix
  head.push_back(1);
  put_uleb(head, body.size());
  Sha256 sh;
  sh.update(head.data(), head.size());
  sh.update(body.data(), body.size());
  Encoded e;
  sh.digest(e.hash.data());
  static const uint8_t MAGIC[4] = {0x85, 0x6f, 0x4a, 0x83};
  e.raw_len = 8 + head.size() + body.size();
  e.bytes.insert(e.bytes.end(), MAGIC, MAGIC + 4);
  e.bytes.insert(e.bytes.end(), e.hash.begin(), e.hash.begin() + 4);
  if (deflate && e.raw_len >= 256) {
    // chunk type 2: raw DEFLATE of the chunk data; checksum/hash remain those of the type-1 form
    // (reference columnar.js:798-811)
    uLongf cap = compressBound(body.size()) + 64;
    Bytes z(cap);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2");
    zs.next_in = body.data(); zs.avail_in = (uInt)body.size();
    zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
    if (::deflate(&zs, Z_FINISH) != Z_STREAM_END) throw std::runtime_error("deflate");
    z.resize(zs.total_out);
    deflateEnd(&zs);
    e.bytes.push_back(2);
    put_uleb(e.bytes, z.size());
    e.bytes.insert(e.bytes.end(), z.begin(), z.end());
  } else {
    e.bytes.insert(e.bytes.end(), head.begin(), head.end());
    e.bytes.insert(e.bytes.end(), body.begin(), body.end());
  }
  return e;
}

}  // namespace amlog
