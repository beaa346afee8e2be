// Synthetic saved-document generator (BASELINE config 5: "Automerge.load from columnar binary, mixed map/list/text doc").
//
// Writes a document chunk (chunk type 0) in the reference's format (backend/columnar.js:983-1004 encodeDocumentHeader,
// :56-94 column specs, :1052-1057 per-column DEFLATE) directly from a synthesised FINAL state -- rows in canonical order
// with their succ lists -- instead of replaying a history: root map -> Text objects (typing runs by many actors, some
// elements deleted), nested maps (keys with overwritten / conflicting values, counters with increments) and lists of
// primitives (some elements updated). The reference's own Backend.load() accepts these documents (checked at small
// sizes by tests/golden/synthetic_doc_*.json).
#include <array>
#include "wire.hpp"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <map>

using namespace amlog;

extern "C" {
typedef struct {
  uint32_t n_actors;
  uint32_t n_texts, text_len;      // Text objects and elements per object
  uint32_t n_maps, keys_per_map;   // first-level maps under root; each also holds n_submaps nested maps
  uint32_t n_submaps;
  uint32_t n_lists, list_len;
  uint32_t deflate;                // DEFLATE columns >= 256 bytes like the reference
  uint32_t pad;
  uint64_t seed;
} amlog_doc_params;
int amlog_generate_document(const amlog_doc_params* p, uint8_t** out, uint64_t* out_len, uint64_t* n_rows);
void amlog_free_bytes(uint8_t* p);
}

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  uint64_t below(uint64_t n) { return next() % n; }
  double unit() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct Row {
  Id obj;            // ctr 0 = root
  bool has_key = false;
  std::string key;
  Id elem;           // list rows: element id (ctr 0 = _head for inserts)
  Id id;
  bool insert = false;
  uint32_t action = SET;
  uint32_t vtype = V_NULL;
  int64_t ival = 0;
  std::string sval;
  std::vector<Id> succ;
};

struct Builder {
  Rng rng;
  std::vector<Bytes> actors;
  uint64_t ctr = 0;  // global Lamport counter: every new op id takes the next value, so parents precede children
  // rows grouped per object; objects are emitted in (ctr, actor) order of their make op, root first
  std::map<std::pair<uint64_t, std::string>, std::vector<Row>> objects;  // key: (make ctr, actor bytes as string) ; root = (0,"")
  explicit Builder(uint64_t seed) : rng(seed) {}

  Id fresh(uint32_t actor) { return Id{++ctr, actor}; }
  uint32_t any_actor() { return (uint32_t)rng.below(actors.size()); }
  std::pair<uint64_t, std::string> okey(Id obj) {
    if (!obj.ctr) return {0, std::string()};
    return {obj.ctr, std::string(actors[obj.actor].begin(), actors[obj.actor].end())};
  }
  std::vector<Row>& rows_of(Id obj) { return objects[okey(obj)]; }
  bool id_less(const Id& a, const Id& b) const {
    if (a.ctr != b.ctr) return a.ctr < b.ctr;
    return actors[a.actor] < actors[b.actor];
  }
  void sort_succ(Row& r) { std::sort(r.succ.begin(), r.succ.end(), [&](const Id& a, const Id& b) { return id_less(a, b); }); }

  Id make(Id parent, const std::string& key, uint32_t action) {
    Row r;
    r.obj = parent; r.has_key = true; r.key = key; r.id = fresh(any_actor()); r.action = action;
    rows_of(parent).push_back(r);
    rows_of(r.id);  // the (possibly empty) object exists
    return r.id;
  }

  // map keys with one value, an overwritten value, two conflicting values, or a counter with increments
  void fill_map(Id obj, uint32_t n_keys, const char* prefix) {
    for (uint32_t k = 0; k < n_keys; k++) {
      char name[32];
      snprintf(name, sizeof name, "%s%05u", prefix, k);
      double u = rng.unit();
      auto set_row = [&](uint32_t vtype, int64_t v, const char* s) {
        Row r;
        r.obj = obj; r.has_key = true; r.key = name; r.id = fresh(any_actor()); r.action = SET; r.vtype = vtype; r.ival = v;
        if (s) r.sval = s;
        return r;
      };
      std::vector<Row>& rows = rows_of(obj);
      if (u < 0.55) {
        rows.push_back(set_row(V_INT, (int64_t)rng.below(100000) - 50000, nullptr));
      } else if (u < 0.75) {  // overwritten once
        Row a = set_row(V_UTF8, 0, "old"), b = set_row(V_UTF8, 0, "new value");
        a.succ.push_back(b.id);
        rows.push_back(a); rows.push_back(b);
      } else if (u < 0.9) {   // two concurrent values: a conflict
        rows.push_back(set_row(V_UINT, (int64_t)rng.below(1000), nullptr));
        rows.push_back(set_row(V_TRUE, 0, nullptr));
      } else {                // counter with 1..3 increments
        Row c = set_row(V_COUNTER, (int64_t)rng.below(10), nullptr);
        uint32_t n_inc = 1 + (uint32_t)rng.below(3);
        std::vector<Row> incs;
        for (uint32_t i = 0; i < n_inc; i++) {
          Row inc = set_row(V_INT, 1 + (int64_t)rng.below(5), nullptr);
          inc.action = INC;
          c.succ.push_back(inc.id);
          incs.push_back(inc);
        }
        sort_succ(c);
        rows.push_back(c);
        for (auto& r : incs) rows.push_back(r);
      }
    }
  }

  // a list/text object built from typing runs that all hang off _head (children of the head in descending id order)
  void fill_list(Id obj, uint32_t n_elems, bool text) {
    struct Run { std::vector<Row> rows; Id head; };
    std::vector<Run> runs;
    uint32_t made = 0;
    while (made < n_elems) {
      uint32_t len = 1 + (uint32_t)rng.below(40);
      if (len > n_elems - made) len = n_elems - made;
      uint32_t actor = any_actor();
      Run run;
      Id prev{0, 0};
      for (uint32_t i = 0; i < len; i++) {
        Row r;
        r.obj = obj; r.elem = prev; r.insert = true; r.id = fresh(actor); r.action = SET;
        if (text) { r.vtype = V_UTF8; r.sval.assign(1, "abcdefghijklmnopqrstuvwxyz "[rng.below(27)]); }
        else { r.vtype = V_INT; r.ival = (int64_t)rng.below(1000); }
        if (i == 0) run.head = r.id;
        prev = r.id;
        run.rows.push_back(r);
      }
      runs.push_back(std::move(run));
      made += len;
    }
    // deletions and (lists only) updates get ids after every insert, like edits made later
    for (auto& run : runs)
      for (size_t i = 0; i < run.rows.size(); i++) {
        double u = rng.unit();
        if (u < 0.2) run.rows[i].succ.push_back(fresh(any_actor()));  // deleted: the del op is only a succ entry
      }
    std::vector<Row>& out = rows_of(obj);
    std::sort(runs.begin(), runs.end(), [&](const Run& a, const Run& b) { return id_less(b.head, a.head); });
    for (auto& run : runs)
      for (auto& r : run.rows) {
        Row upd;
        bool has_upd = false;
        if (!text && r.succ.empty() && rng.unit() < 0.15) {  // element assigned a new value later
          upd.obj = obj; upd.elem = r.id; upd.insert = false; upd.id = fresh(any_actor()); upd.action = SET; upd.vtype = V_UTF8; upd.sval = "upd";
          r.succ.push_back(upd.id);
          has_upd = true;
        }
        out.push_back(r);
        if (has_upd) out.push_back(upd);
      }
  }
};

void deflate_if_large(Bytes& data, uint32_t& id, bool enable) {
  if (!enable || data.size() < 256) return;
  uLongf cap = compressBound(data.size()) + 64;
  Bytes z(cap);
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
  zs.next_in = data.data(); zs.avail_in = (uInt)data.size(); zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
  ::deflate(&zs, Z_FINISH);
  z.resize(zs.total_out);
  deflateEnd(&zs);
  data.swap(z);
  id |= 8;
}

}  // namespace

extern "C" int amlog_generate_document(const amlog_doc_params* p, uint8_t** out, uint64_t* out_len, uint64_t* n_rows) {
  Builder b(p->seed);
  uint32_t A = p->n_actors ? p->n_actors : 16;
  for (uint32_t i = 0; i < A; i++) {
    Bytes a(16);
    for (int k = 0; k < 16; k += 8) { uint64_t r = b.rng.next(); memcpy(&a[k], &r, 8); }
    a[0] = (uint8_t)(((0xa + (a[0] >> 4) % 6) << 4) | (a[0] & 0x0f));
    b.actors.push_back(a);
  }
  Id root{0, 0};
  b.rows_of(root);
  char name[32];
  // containers first (small ids), contents afterwards: a child's ops always follow its make op
  std::vector<Id> texts, maps, lists;
  std::vector<std::vector<Id>> submaps;
  for (uint32_t i = 0; i < p->n_texts; i++) { snprintf(name, sizeof name, "t%04u", i); texts.push_back(b.make(root, name, MAKE_TEXT)); }
  for (uint32_t i = 0; i < p->n_maps; i++) { snprintf(name, sizeof name, "m%04u", i); maps.push_back(b.make(root, name, MAKE_MAP)); }
  for (uint32_t i = 0; i < p->n_lists; i++) { snprintf(name, sizeof name, "l%04u", i); lists.push_back(b.make(root, name, MAKE_LIST)); }
  for (Id m : maps) {
    std::vector<Id> subs;
    for (uint32_t j = 0; j < p->n_submaps; j++) { snprintf(name, sizeof name, "sub%02u", j); subs.push_back(b.make(m, name, j % 2 ? MAKE_TABLE : MAKE_MAP)); }
    submaps.push_back(subs);
  }
  for (Id t : texts) b.fill_list(t, p->text_len, true);
  for (size_t i = 0; i < maps.size(); i++) {
    b.fill_map(maps[i], p->keys_per_map, "k");
    for (Id s : submaps[i]) b.fill_map(s, p->keys_per_map / 4 + 1, "n");
  }
  for (Id l : lists) b.fill_list(l, p->list_len, false);
  b.fill_map(root, 8, "zz");  // a few plain keys on the root as well

  // ---- columns (documents: columnar.js:80-84 DOC_OPS_COLUMNS) ----
  std::vector<OptInt> objActor, objCtr, keyActor, keyCtr, idActor, idCtr, action, valLen, succNum, succActor, succCtr;
  std::vector<OptStr> keyStr;
  std::vector<uint8_t> insert;
  Bytes valRaw;
  uint64_t rows = 0;
  for (auto& kv : b.objects) {
    std::vector<Row>& rs = kv.second;
    bool is_map = rs.empty() || rs[0].has_key;
    if (is_map)  // keys ascending (byte order = UTF-16 order for these ASCII keys), ops of one key by ascending id
      std::stable_sort(rs.begin(), rs.end(), [&](const Row& x, const Row& y) { return x.key != y.key ? x.key < y.key : b.id_less(x.id, y.id); });
    for (Row& r : rs) {
      rows++;
      if (r.obj.ctr) { objActor.push_back(OptInt::of(r.obj.actor)); objCtr.push_back(OptInt::of((int64_t)r.obj.ctr)); }
      else { objActor.push_back(OptInt::none()); objCtr.push_back(OptInt::none()); }
      if (r.has_key) { keyActor.push_back(OptInt::none()); keyCtr.push_back(OptInt::none()); keyStr.push_back({false, r.key}); }
      else if (!r.elem.ctr) { keyActor.push_back(OptInt::none()); keyCtr.push_back(OptInt::of(0)); keyStr.push_back({true, ""}); }
      else { keyActor.push_back(OptInt::of(r.elem.actor)); keyCtr.push_back(OptInt::of((int64_t)r.elem.ctr)); keyStr.push_back({true, ""}); }
      idActor.push_back(OptInt::of(r.id.actor));
      idCtr.push_back(OptInt::of((int64_t)r.id.ctr));
      insert.push_back(r.insert);
      action.push_back(OptInt::of(r.action));
      uint32_t vt = (r.action == SET || r.action == INC) ? r.vtype : V_NULL;
      size_t before = valRaw.size();
      switch (vt) {
        case V_NULL: case V_FALSE: case V_TRUE: break;
        case V_UINT: put_uleb(valRaw, (uint64_t)r.ival); break;
        case V_INT: case V_COUNTER: case V_TIMESTAMP: put_sleb(valRaw, r.ival); break;
        default: valRaw.insert(valRaw.end(), r.sval.begin(), r.sval.end()); break;
      }
      valLen.push_back(OptInt::of((int64_t)(((valRaw.size() - before) << 4) | vt)));
      b.sort_succ(r);
      succNum.push_back(OptInt::of((int64_t)r.succ.size()));
      for (const Id& s : r.succ) { succActor.push_back(OptInt::of(s.actor)); succCtr.push_back(OptInt::of((int64_t)s.ctr)); }
    }
  }
  struct Col { uint32_t id; Bytes data; };
  std::vector<Col> ops(14);
  ops[0].id = 0x01; rle_uint(ops[0].data, objActor);   ops[1].id = 0x02; rle_uint(ops[1].data, objCtr);
  ops[2].id = 0x11; rle_uint(ops[2].data, keyActor);   ops[3].id = 0x13; delta_encode(ops[3].data, keyCtr);
  ops[4].id = 0x15; rle_utf8(ops[4].data, keyStr);     ops[5].id = 0x21; rle_uint(ops[5].data, idActor);
  ops[6].id = 0x23; delta_encode(ops[6].data, idCtr);  ops[7].id = 0x34; bool_encode(ops[7].data, insert);
  ops[8].id = 0x42; rle_uint(ops[8].data, action);     ops[9].id = 0x56; rle_uint(ops[9].data, valLen);
  ops[10].id = 0x57; ops[10].data = valRaw;            ops[11].id = 0x80; rle_uint(ops[11].data, succNum);
  ops[12].id = 0x81; rle_uint(ops[12].data, succActor); ops[13].id = 0x83; delta_encode(ops[13].data, succCtr);

  // ---- change metadata (columnar.js:86-94): one synthetic change per actor, chained by index ----
  std::vector<OptInt> cActor, cSeq, cMaxOp, cTime, cDepsNum, cDepsIndex, cExtraLen;
  std::vector<OptStr> cMsg;
  for (uint32_t a = 0; a < A; a++) {
    cActor.push_back(OptInt::of(a)); cSeq.push_back(OptInt::of(1)); cMaxOp.push_back(OptInt::of((int64_t)b.ctr)); cTime.push_back(OptInt::of(0));
    cMsg.push_back({true, ""});
    cDepsNum.push_back(OptInt::of(a ? 1 : 0));
    if (a) cDepsIndex.push_back(OptInt::of(a - 1));
    cExtraLen.push_back(OptInt::of(V_BYTES));
  }
  std::vector<Col> chg(8);
  chg[0].id = 0x01; rle_uint(chg[0].data, cActor);     chg[1].id = 0x03; delta_encode(chg[1].data, cSeq);
  chg[2].id = 0x13; delta_encode(chg[2].data, cMaxOp); chg[3].id = 0x23; delta_encode(chg[3].data, cTime);
  chg[4].id = 0x35; rle_utf8(chg[4].data, cMsg);       chg[5].id = 0x40; rle_uint(chg[5].data, cDepsNum);
  chg[6].id = 0x43; delta_encode(chg[6].data, cDepsIndex); chg[7].id = 0x56; rle_uint(chg[7].data, cExtraLen);

  for (auto& c : chg) deflate_if_large(c.data, c.id, p->deflate != 0);
  for (auto& c : ops) deflate_if_large(c.data, c.id, p->deflate != 0);

  Bytes body;
  put_uleb(body, A);
  for (auto& a : b.actors) { put_uleb(body, a.size()); body.insert(body.end(), a.begin(), a.end()); }
  put_uleb(body, 1);  // one head: an arbitrary 32-byte value (load does not recompute hashes)
  for (int k = 0; k < 32; k++) body.push_back((uint8_t)(0x11 * (k % 15 + 1)));
  auto dir = [&](std::vector<Col>& cols) {
    size_t n = 0;
    for (auto& c : cols) if (!c.data.empty()) n++;
    put_uleb(body, n);
    for (auto& c : cols) if (!c.data.empty()) { put_uleb(body, c.id); put_uleb(body, c.data.size()); }
  };
  dir(chg);
  dir(ops);
  for (auto& c : chg) body.insert(body.end(), c.data.begin(), c.data.end());
  for (auto& c : ops) body.insert(body.end(), c.data.begin(), c.data.end());
  put_uleb(body, A - 1);  // headsIndexes
  Bytes head;
  head.push_back(0);
  put_uleb(head, body.size());
  Sha256 sh;
  sh.update(head.data(), head.size());
  sh.update(body.data(), body.size());
  uint8_t digest[32];
  sh.digest(digest);
  static const uint8_t MAGIC[4] = {0x85, 0x6f, 0x4a, 0x83};
  Bytes doc(MAGIC, MAGIC + 4);
  doc.insert(doc.end(), digest, digest + 4);
  doc.insert(doc.end(), head.begin(), head.end());
  doc.insert(doc.end(), body.begin(), body.end());
  *out = (uint8_t*)malloc(doc.size());
  if (!*out) return -3;
  memcpy(*out, doc.data(), doc.size());
  *out_len = doc.size();
  *n_rows = rows;
  return 0;
}

extern "C" void amlog_free_bytes(uint8_t* p) { free(p); }
