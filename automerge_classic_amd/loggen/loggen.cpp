// Synthetic change-log generator for the bulk-replay benchmark configurations.
//
// Emits binary Automerge changes (the exact wire format a JS frontend + reference `encodeChange`
// would produce; see wire.hpp) for three workload families:
//   kind 0  text_typing      one actor typing into one Text object (BASELINE config 2)
//   kind 1  map_lww          many actors concurrently overwriting root-map keys in synced rounds (config 3)
//   kind 2  text_concurrent  many actors inserting runs / deleting elements in Text objects in synced
//                            rounds; n_objects = 1 (single object, config 4 headline) or >1 (shardable)
// PRNG: splitmix64. C ABI so Python (ctypes) and the N-API addon can both call it.
#include <array>
#include "wire.hpp"
#include <cstdio>
#include <cstdlib>
#include <unordered_set>

using namespace amlog;

extern "C" {
typedef struct {
  uint32_t kind;
  uint32_t n_actors;
  uint32_t n_rounds;        // kind 1, 2
  uint32_t ins_per_change;  // kind 2
  uint32_t del_per_change;  // kind 2
  uint32_t n_objects;       // kind 2
  uint32_t n_keys;          // kind 1
  uint32_t ops_per_change;  // kind 0
  uint64_t n_ops;           // kind 0: number of character inserts
  uint64_t seed;
  uint32_t deflate;         // 1: DEFLATE changes >= 256 bytes like the reference encoder
  uint32_t reserved;
} amlog_params;

typedef struct {
  uint8_t* arena;      // all changes back to back
  uint64_t* offsets;   // n_changes + 1 byte offsets into arena
  uint32_t n_changes;
  uint32_t n_actors;
  uint64_t n_ops;      // total op rows (deletes included)
  uint64_t raw_bytes;  // total size of the changes in uncompressed (chunk type 1) form
} amlog_log;

int amlog_generate(const amlog_params* p, amlog_log* out);
void amlog_free(amlog_log* log);
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
  uint64_t below(uint64_t n) { return next() % n; }  // modulo bias is irrelevant for workload synthesis
  double unit() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct Sink {
  std::vector<uint8_t> arena;
  std::vector<uint64_t> offsets{0};
  uint64_t n_ops = 0, raw_bytes = 0;
  std::array<uint8_t, 32> add(const Change& c, const std::vector<Bytes>& actors, bool deflate) {
    Encoded e = encode_change(c, actors, deflate);
    arena.insert(arena.end(), e.bytes.begin(), e.bytes.end());
    offsets.push_back(arena.size());
    n_ops += c.ops.size();
    raw_bytes += e.raw_len;
    return e.hash;
  }
};

std::vector<Bytes> make_actors(Rng& rng, uint32_t n) {
  // 16 random bytes each (32 hex digits); first nibble forced to a letter so an id is never all-decimal.
  std::vector<Bytes> actors(n);
  for (auto& a : actors) {
    a.resize(16);
    for (int i = 0; i < 16; i += 8) { uint64_t r = rng.next(); memcpy(&a[i], &r, 8); }
    a[0] = (uint8_t)(((0xa + (a[0] >> 4) % 6) << 4) | (a[0] & 0x0f));
  }
  return actors;
}

char rand_char(Rng& rng) {
  static const char alphabet[] = "abcdefghijklmnopqrstuvwxyz etaoin";
  return alphabet[rng.below(sizeof(alphabet) - 1)];
}

Op insert_op(Id obj, Id ref, char ch) {
  Op op;
  op.obj = obj; op.elem = ref; op.insert = true; op.action = SET; op.vtype = V_UTF8; op.sval.assign(1, ch);
  return op;
}

// ---- kind 0: single-actor typing --------------------------------------------------------------------
void gen_text_typing(const amlog_params& p, Sink& sink, std::vector<Bytes>& actors) {
  Rng rng(p.seed);
  actors = make_actors(rng, 1);
  uint32_t K = p.ops_per_change ? p.ops_per_change : 100;
  Change c;
  c.actor = 0; c.seq = 1; c.start_op = 1;
  Op mk; mk.has_key = true; mk.key = "text"; mk.action = MAKE_TEXT;
  c.ops.push_back(mk);
  auto head = sink.add(c, actors, p.deflate);
  Id text{1, 0};
  uint64_t next_ctr = 2, done = 0, seq = 2;
  Id cursor{0, 0};  // _head
  while (done < p.n_ops) {
    Change ch;
    ch.actor = 0; ch.seq = seq++; ch.start_op = next_ctr; ch.deps.push_back(head);
    uint64_t take = std::min<uint64_t>(K, p.n_ops - done);
    for (uint64_t i = 0; i < take; i++) {
      if (rng.unit() >= 0.95) {
        // jump to a uniformly random position: head or any existing element (all are visible)
        uint64_t n_elems = next_ctr - 2;
        uint64_t r = rng.below(n_elems + 1);
        cursor = r == 0 ? Id{0, 0} : Id{r + 1, 0};
      }
      ch.ops.push_back(insert_op(text, cursor, rand_char(rng)));
      cursor = Id{next_ctr++, 0};
    }
    done += take;
    head = sink.add(ch, actors, p.deflate);
  }
}

// ---- kind 1: concurrent map overwrites -----------------------------------------------------------------
void gen_map_lww(const amlog_params& p, Sink& sink, std::vector<Bytes>& actors) {
  Rng rng(p.seed);
  uint32_t A = p.n_actors ? p.n_actors : 32, R = p.n_rounds ? p.n_rounds : 8, NK = p.n_keys ? p.n_keys : 10000;
  actors = make_actors(rng, A);
  uint32_t per = (NK + A - 1) / A;
  std::vector<std::string> keys(NK);
  for (uint32_t k = 0; k < NK; k++) { char b[16]; snprintf(b, sizeof b, "k%05u", k); keys[k] = b; }
  std::vector<std::vector<Id>> visible(NK), fresh(NK);
  std::vector<std::array<uint8_t, 32>> heads;
  uint64_t max_op = 0;
  std::vector<uint32_t> perm(NK);
  for (uint32_t k = 0; k < NK; k++) perm[k] = k;
  for (uint32_t r = 0; r < R; r++) {
    std::vector<std::array<uint8_t, 32>> new_heads;
    std::vector<uint32_t> touched;
    for (uint32_t a = 0; a < A; a++) {
      Change c;
      c.actor = a; c.seq = r + 1; c.start_op = max_op + 1; c.deps = heads;
      // `per` distinct uniformly random keys: partial Fisher-Yates over a persistent permutation
      for (uint32_t i = 0; i < per && i < NK; i++) {
        uint32_t j = i + (uint32_t)rng.below(NK - i);
        std::swap(perm[i], perm[j]);
        uint32_t k = perm[i];
        Op op;
        op.has_key = true; op.key = keys[k]; op.action = SET; op.vtype = V_INT;
        op.ival = (int64_t)rng.below(1000000) - 500000;
        op.pred = visible[k];
        c.ops.push_back(op);
        if (fresh[k].empty()) touched.push_back(k);
        fresh[k].push_back(Id{c.start_op + i, a});
      }
      new_heads.push_back(sink.add(c, actors, p.deflate));
    }
    for (uint32_t k : touched) { visible[k].swap(fresh[k]); fresh[k].clear(); }
    max_op += std::min(per, NK);
    heads.swap(new_heads);
  }
}

// ---- kind 2: concurrent text editing in synced rounds -------------------------------------------------
void gen_text_concurrent(const amlog_params& p, Sink& sink, std::vector<Bytes>& actors) {
  Rng rng(p.seed);
  uint32_t A = p.n_actors ? p.n_actors : 64, R = p.n_rounds ? p.n_rounds : 64;
  uint32_t n_ins = p.ins_per_change ? p.ins_per_change : 200, n_del = p.del_per_change, NO = p.n_objects ? p.n_objects : 1;
  actors = make_actors(rng, A);
  // setup change by actor 0: one makeText per object at root keys "text" (single) or "t00".."tNN"
  Change setup;
  setup.actor = 0; setup.seq = 1; setup.start_op = 1;
  for (uint32_t o = 0; o < NO; o++) {
    Op mk; mk.has_key = true; mk.action = MAKE_TEXT;
    if (NO == 1) mk.key = "text"; else { char b[16]; snprintf(b, sizeof b, "t%02u", o); mk.key = b; }
    setup.ops.push_back(mk);
  }
  std::vector<std::array<uint8_t, 32>> heads{sink.add(setup, actors, p.deflate)};
  uint64_t max_op = NO;
  std::vector<std::vector<Id>> visible(NO);
  for (uint32_t r = 0; r < R; r++) {
    std::vector<std::array<uint8_t, 32>> new_heads;
    std::vector<std::vector<Id>> added(NO);
    std::vector<std::unordered_set<uint64_t>> removed(NO);  // indexes into visible[o]
    uint64_t longest = 0;
    for (uint32_t a = 0; a < A; a++) {
      uint32_t o = NO == 1 ? 0 : (uint32_t)rng.below(NO);
      Id obj{(uint64_t)o + 1, 0};
      const std::vector<Id>& vis = visible[o];
      Change c;
      c.actor = a; c.seq = r + 1 + (a == 0 ? 1 : 0); c.start_op = max_op + 1; c.deps = heads;
      uint64_t ctr = c.start_op;
      uint64_t pos = rng.below(vis.size() + 1);
      Id ref = pos == 0 ? Id{0, 0} : vis[pos - 1];
      for (uint32_t i = 0; i < n_ins; i++) {
        c.ops.push_back(insert_op(obj, ref, rand_char(rng)));
        ref = Id{ctr++, a};
        added[o].push_back(ref);
      }
      uint32_t ndel = (uint32_t)std::min<uint64_t>(n_del, vis.size());
      std::unordered_set<uint64_t> mine;
      while (mine.size() < ndel) {
        uint64_t idx = rng.below(vis.size());
        if (!mine.insert(idx).second) continue;
        Op op;
        op.obj = obj; op.elem = vis[idx]; op.action = DEL; op.pred.push_back(vis[idx]);
        c.ops.push_back(op);
        removed[o].insert(idx);
        ctr++;
      }
      longest = std::max<uint64_t>(longest, c.ops.size());
      new_heads.push_back(sink.add(c, actors, p.deflate));
    }
    for (uint32_t o = 0; o < NO; o++) {
      std::vector<Id> next;
      next.reserve(visible[o].size() + added[o].size());
      for (uint64_t i = 0; i < visible[o].size(); i++) if (!removed[o].count(i)) next.push_back(visible[o][i]);
      next.insert(next.end(), added[o].begin(), added[o].end());
      visible[o].swap(next);
    }
    max_op += longest;  // every actor has seen every op of the round once synced
    heads.swap(new_heads);
  }
}

}  // namespace

extern "C" int amlog_generate(const amlog_params* p, amlog_log* out) {
  try {
    Sink sink;
    std::vector<Bytes> actors;
    switch (p->kind) {
      case 0: gen_text_typing(*p, sink, actors); break;
      case 1: gen_map_lww(*p, sink, actors); break;
      case 2: gen_text_concurrent(*p, sink, actors); break;
      default: return -2;
    }
    out->n_changes = (uint32_t)(sink.offsets.size() - 1);
    out->n_actors = (uint32_t)actors.size();
    out->n_ops = sink.n_ops;
    out->raw_bytes = sink.raw_bytes;
    out->arena = (uint8_t*)malloc(sink.arena.size() ? sink.arena.size() : 1);
    out->offsets = (uint64_t*)malloc(sink.offsets.size() * sizeof(uint64_t));
    if (!out->arena || !out->offsets) return -3;
    memcpy(out->arena, sink.arena.data(), sink.arena.size());
    memcpy(out->offsets, sink.offsets.data(), sink.offsets.size() * sizeof(uint64_t));
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "amlog_generate: %s\n", e.what());
    return -1;
  }
}

extern "C" void amlog_free(amlog_log* log) {
  free(log->arena);
  free(log->offsets);
  log->arena = nullptr;
  log->offsets = nullptr;
}
