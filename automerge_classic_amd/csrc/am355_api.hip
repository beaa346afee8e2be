// C ABI of the replay engine (include/am355.h): context, staging, host-side causal scheduler, orchestration of
// the device stages, patch-IR download.
//
// Host-side logic restated from the reference (paths relative to the reference tree):
//   inflate of DEFLATEd changes    backend/columnar.js:813-823 inflateChange (zlib raw inflate)
//   causal scheduling              backend/new.js:1550-1597 applyChanges, :1822-1841 retry loop
//   actor table                    backend/new.js:1434-1451 getActorTable (first-applied order; the engine
//                                  additionally ranks actors lexicographically for numeric op-id comparison)
//   envelope                       backend/new.js:1870-1873, 2064-2067 (maxOp, clock, deps, pendingChanges)
#include "../../include/am355.h"
#include "am355_decode.h"
#include "am355_merge.h"
#include "am355_prims.h"
#include "am355_render.h"

#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

using namespace am355;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) return false;
    cap = want;
    return true;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T> T* as() { return (T*)p; }
};

struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return false;
    cap = want;
    return true;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T> T* as() { return (T*)p; }
};

struct Hash32 {
  uint8_t b[32];
  bool operator==(const Hash32& o) const { return memcmp(b, o.b, 32) == 0; }
};
struct Hash32Hasher {
  size_t operator()(const Hash32& h) const { size_t v; memcpy(&v, h.b, sizeof v); return v; }
};

}  // namespace

struct am355_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {};
  std::string err;
  uint32_t flags = 0;

  // staged batch
  std::vector<uint8_t> raw;        // uncompressed changes, host copy (the scheduler reads deps / actor ids here)
  std::vector<uint64_t> raw_off;
  uint32_t n_changes = 0;
  bool staged = false, replayed = false, ir_fetched = false;
  DevBuf d_arena, d_offsets, d_metas;
  HostBuf h_metas;

  // schedule
  std::vector<ChangePlan> plans;
  std::vector<uint32_t> amap;
  std::vector<ActorSpan> spans;
  std::vector<uint32_t> actor_tab_off;
  std::vector<std::string> actors;        // by rank
  std::vector<uint32_t> clock_actor;      // first-applied order
  std::vector<uint64_t> clock_seq;
  std::vector<uint8_t> heads;
  uint32_t n_applied = 0, n_pending = 0;
  uint64_t n_ops = 0, n_preds = 0, max_op = 0;
  DevBuf d_plans, d_amap, d_spans, d_tab_off;

  // op rows + merge buffers (one arena of u32 words per purpose)
  DevBuf d_cols, d_pred, d_merge, d_sort, d_ir, d_counts;
  OpCols cols{};
  MergeBufs mb{};
  PatchIR ir{};
  HostBuf h_counts;
  Counts counts{};

  // host IR
  HostBuf h_ir, h_rows;
  am355_patch_ir hir{};
  std::vector<uint32_t> actor_off;
  std::vector<uint8_t> actor_bytes;
  std::string json;

  am355_stats stats{};
};

static int fail(am355_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  c->err = buf;
  return code;
}

#define HIPCHK(ctx, call)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return fail(ctx, AM355_E_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

extern "C" am355_ctx* am355_create(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  am355_ctx* c = new am355_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete c; return nullptr; }
  return c;
}

extern "C" void am355_destroy(am355_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (DevBuf* b : {&c->d_arena, &c->d_offsets, &c->d_metas, &c->d_plans, &c->d_amap, &c->d_spans, &c->d_tab_off, &c->d_cols, &c->d_pred,
                    &c->d_merge, &c->d_sort, &c->d_ir, &c->d_counts})
    b->release();
  for (HostBuf* b : {&c->h_metas, &c->h_counts, &c->h_ir, &c->h_rows}) b->release();
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char* am355_last_error(const am355_ctx* c) { return c ? c->err.c_str() : "no context (no GPU?)"; }
extern "C" uint32_t am355_flags(const am355_ctx* c) { return c ? c->flags : 0; }

// ---------------------------------------------------------------------------------------------------------
// staging
// ---------------------------------------------------------------------------------------------------------
static bool read_uleb_host(const uint8_t* p, size_t len, size_t& off, uint64_t& out) {
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

extern "C" int am355_load_changes(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n) {
  if (!c || (!arena && n) || !offsets) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  (void)hipSetDevice(c->device);
  c->staged = c->replayed = c->ir_fetched = false;
  c->flags = 0;
  c->raw.clear();
  c->raw_off.assign(1, 0);
  c->raw.reserve(offsets[n] + 64);
  for (uint32_t i = 0; i < n; i++) {
    const uint8_t* p = arena + offsets[i];
    size_t len = offsets[i + 1] - offsets[i];
    if (len > 9 && p[8] == 2) {
      // chunk type 2: rebuild the uncompressed container (columnar.js:813-823); the checksum/hash are over that form
      size_t off = 9;
      uint64_t clen;
      if (!read_uleb_host(p, len, off, clen) || clen > len - off) { c->flags |= AM355_F_BAD_CHUNK; return fail(c, AM355_E_INVALID, "change %u: bad deflate container", i); }
      std::vector<uint8_t> out(std::max<size_t>(clen * 4, 1024));
      size_t outlen = 0;
      for (;;) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return fail(c, AM355_E_NOMEM, "inflateInit2");
        zs.next_in = (Bytef*)(p + off);
        zs.avail_in = (uInt)clen;
        zs.next_out = out.data();
        zs.avail_out = (uInt)out.size();
        int rc = inflate(&zs, Z_FINISH);
        outlen = zs.total_out;
        inflateEnd(&zs);
        if (rc == Z_STREAM_END) break;
        if (rc == Z_BUF_ERROR || rc == Z_OK) { out.resize(out.size() * 4); continue; }
        c->flags |= AM355_F_BAD_DEFLATE;
        return fail(c, AM355_E_INVALID, "change %u: invalid deflate data", i);
      }
      c->raw.insert(c->raw.end(), p, p + 8);
      c->raw.push_back(1);
      uint64_t v = outlen;
      do { uint8_t x = v & 0x7f; v >>= 7; if (v) x |= 0x80; c->raw.push_back(x); } while (v);
      c->raw.insert(c->raw.end(), out.begin(), out.begin() + outlen);
    } else {
      c->raw.insert(c->raw.end(), p, p + len);
    }
    c->raw_off.push_back(c->raw.size());
  }
  if (c->raw.size() >= 0xfff00000ull) return fail(c, AM355_E_UNSUPPORTED, "batch larger than 4 GiB (32-bit arena offsets)");
  c->n_changes = n;
  if (!c->d_arena.ensure(c->raw.size() + 64) || !c->d_offsets.ensure(sizeof(uint64_t) * (n + 1)) || !c->d_metas.ensure(sizeof(ChangeMeta) * (size_t)std::max(n, 1u)) ||
      !c->h_metas.ensure(sizeof(ChangeMeta) * (size_t)std::max(n, 1u)) || !c->d_counts.ensure(sizeof(Counts)) || !c->h_counts.ensure(sizeof(Counts)))
    return fail(c, AM355_E_NOMEM, "device allocation failed");
  HIPCHK(c, hipMemcpyAsync(c->d_arena.p, c->raw.data(), c->raw.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_offsets.p, c->raw_off.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->staged = true;
  c->stats = am355_stats{};
  c->stats.n_changes = n;
  c->stats.raw_bytes = c->raw.size();
  return AM355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// host scheduler
// ---------------------------------------------------------------------------------------------------------
static int bits_for64(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b)) b++;
  return b;
}

static int schedule(am355_ctx* c) {
  const ChangeMeta* metas = c->h_metas.as<ChangeMeta>();
  uint32_t n = c->n_changes;
  const uint8_t* raw = c->raw.data();
  uint32_t dev_flags = 0;
  for (uint32_t i = 0; i < n; i++) dev_flags |= metas[i].flags;
  if (dev_flags) {
    c->flags |= dev_flags;
    return fail(c, (dev_flags & (F_OVERFLOW | F_UNSUPPORTED)) ? AM355_E_UNSUPPORTED : AM355_E_INVALID, "malformed change (flags 0x%x)", dev_flags);
  }
  // ---- actor ids: global table ranked lexicographically (hex-string order == byte order, new.js:65) ----
  std::unordered_map<std::string, uint32_t> actor_ix;
  std::vector<std::string> names;
  std::vector<std::vector<uint32_t>> local(n);  // per change: provisional ids of [author, others...]
  for (uint32_t i = 0; i < n; i++) {
    const ChangeMeta& m = metas[i];
    const uint8_t* p = raw + m.base;
    auto intern = [&](const uint8_t* b, size_t len) {
      std::string s((const char*)b, len);
      auto it = actor_ix.find(s);
      if (it != actor_ix.end()) return it->second;
      uint32_t id = (uint32_t)names.size();
      actor_ix.emplace(s, id);
      names.push_back(std::move(s));
      return id;
    };
    local[i].push_back(intern(p + m.actor_off, m.actor_len));
    size_t off = m.others_off;
    for (uint32_t k = 0; k < m.n_other; k++) {
      uint64_t l;
      read_uleb_host(p, m.len, off, l);
      local[i].push_back(intern(p + off, (size_t)l));
      off += (size_t)l;
    }
  }
  uint32_t na = (uint32_t)names.size();
  std::vector<uint32_t> by_rank(na), rank(na);
  for (uint32_t i = 0; i < na; i++) by_rank[i] = i;
  std::sort(by_rank.begin(), by_rank.end(), [&](uint32_t x, uint32_t y) { return names[x] < names[y]; });  // std::string compares bytes as unsigned char
  for (uint32_t r = 0; r < na; r++) rank[by_rank[r]] = r;
  c->actors.resize(na);
  for (uint32_t r = 0; r < na; r++) c->actors[r] = names[by_rank[r]];

  // ---- causal scheduling (new.js:1550-1597 inside the retry loop of :1822-1841) ----
  std::unordered_map<Hash32, uint32_t, Hash32Hasher> known;  // applied hashes
  std::unordered_map<Hash32, bool, Hash32Hasher> heads;
  std::vector<uint64_t> clock(na, 0);
  std::vector<uint8_t> has_clock(na, 0), actor_read(na, 0);
  c->clock_actor.clear();
  std::vector<uint32_t> queue(n), next_q, applied_all;
  for (uint32_t i = 0; i < n; i++) queue[i] = i;
  uint32_t sched_flags = 0;
  while (!queue.empty()) {
    std::vector<uint32_t> applied;
    next_q.clear();
    for (uint32_t ci : queue) {
      const ChangeMeta& m = metas[ci];
      Hash32 h;
      memcpy(h.b, m.hash, 32);
      if (known.count(h)) continue;  // duplicate (new.js:1557)
      uint32_t author = rank[local[ci][0]];
      uint64_t expected = clock[author] + 1;
      bool ready = true;
      for (uint32_t k = 0; k < m.n_deps; k++) {
        Hash32 d;
        memcpy(d.b, raw + m.base + m.deps_off + 32 * k, 32);
        if (!known.count(d)) ready = false;
      }
      if (!ready) { next_q.push_back(ci); continue; }
      if (m.seq != expected) { sched_flags |= AM355_F_BAD_SEQ; break; }
      if (!has_clock[author]) { has_clock[author] = 1; c->clock_actor.push_back(author); }
      clock[author] = m.seq;
      known.emplace(h, ci);
      for (uint32_t k = 0; k < m.n_deps; k++) {
        Hash32 d;
        memcpy(d.b, raw + m.base + m.deps_off + 32 * k, 32);
        heads.erase(d);
      }
      heads[h] = true;
      applied.push_back(ci);
    }
    if (sched_flags) break;
    // changes are read in applied order; each may only mention actors already in the document (new.js:1442-1449)
    for (uint32_t ci : applied) {
      actor_read[rank[local[ci][0]]] = 1;
      for (uint32_t a : local[ci])
        if (!actor_read[rank[a]]) sched_flags |= AM355_F_UNKNOWN_ACTOR;
      applied_all.push_back(ci);
    }
    queue.swap(next_q);
    if (applied.empty()) break;
  }
  if (sched_flags) {
    c->flags |= sched_flags;
    return fail(c, AM355_E_INVALID, "change schedule rejected (flags 0x%x)", sched_flags);
  }
  c->n_applied = (uint32_t)applied_all.size();
  c->n_pending = (uint32_t)queue.size();
  c->clock_seq.clear();
  for (uint32_t a : c->clock_actor) c->clock_seq.push_back(clock[a]);
  {
    std::vector<Hash32> hs;
    for (auto& kv : heads) hs.push_back(kv.first);
    std::sort(hs.begin(), hs.end(), [](const Hash32& x, const Hash32& y) { return memcmp(x.b, y.b, 32) < 0; });
    c->heads.resize(hs.size() * 32);
    for (size_t i = 0; i < hs.size(); i++) memcpy(&c->heads[32 * i], hs[i].b, 32);
  }

  // ---- launch plan for the decode kernels, op-id -> row tables ----
  c->plans.clear();
  c->amap.clear();
  uint64_t ops = 0, preds = 0, max_op = 0;
  std::vector<std::vector<ActorSpan>> per_actor(na);
  for (uint32_t ci : applied_all) {
    const ChangeMeta& m = metas[ci];
    ChangePlan pl;
    pl.change = ci;
    pl.op_base = (uint32_t)ops;
    pl.pred_base = (uint32_t)preds;
    pl.amap_base = (uint32_t)c->amap.size();
    pl.author = rank[local[ci][0]];
    pl.n_actors = (uint32_t)local[ci].size();
    for (uint32_t a : local[ci]) c->amap.push_back(rank[a]);
    if (m.n_ops) {
      per_actor[pl.author].push_back(ActorSpan{(uint32_t)m.start_op, m.n_ops, pl.op_base});
      max_op = std::max<uint64_t>(max_op, m.start_op + m.n_ops - 1);
      c->plans.push_back(pl);
    }
    ops += m.n_ops;
    preds += m.n_preds;
    if (ops >= 0x7ffffff0ull || preds >= 0xfffffff0ull) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "more than 2^31 ops in one batch"); }
  }
  c->n_ops = ops;
  c->n_preds = preds;
  c->max_op = max_op;
  c->spans.clear();
  c->actor_tab_off.assign(na + 1, 0);
  for (uint32_t a = 0; a < na; a++) {
    auto& v = per_actor[a];
    std::sort(v.begin(), v.end(), [](const ActorSpan& x, const ActorSpan& y) { return x.start_op < y.start_op; });
    for (size_t k = 1; k < v.size(); k++)
      if ((uint64_t)v[k - 1].start_op + v[k - 1].n_ops > v[k].start_op) {
        c->flags |= AM355_F_DUP_OPID;
        return fail(c, AM355_E_INVALID, "overlapping op id ranges for one actor (duplicate operation ID)");
      }
    c->actor_tab_off[a] = (uint32_t)c->spans.size();
    c->spans.insert(c->spans.end(), v.begin(), v.end());
  }
  c->actor_tab_off[na] = (uint32_t)c->spans.size();
  return AM355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// replay
// ---------------------------------------------------------------------------------------------------------
template <class T>
static T* carve(uint8_t*& p, size_t count) {
  T* r = (T*)p;
  p += ((count * sizeof(T)) + 255) & ~(size_t)255;
  return r;
}

static size_t carve_size(size_t count, size_t elem) { return ((count * elem) + 255) & ~(size_t)255; }

extern "C" int am355_replay(am355_ctx* c) {
  if (!c) return AM355_E_ARG;
  if (!c->staged) return fail(c, AM355_E_STATE, "am355_load_changes must be called first");
  (void)hipSetDevice(c->device);
  c->replayed = c->ir_fetched = false;
  c->flags = 0;
  auto t_begin = std::chrono::steady_clock::now();
  hipStream_t st = c->stream;
  uint32_t n = c->n_changes;

  // ---- stage 1a: parse + hash + count (device), metas to host ----
  HIPCHK(c, hipEventRecord(c->ev[0], st));
  launch_parse_changes(c->d_arena.as<uint8_t>(), c->d_offsets.as<uint64_t>(), n, c->d_metas.as<ChangeMeta>(), st);
  HIPCHK(c, hipEventRecord(c->ev[1], st));
  HIPCHK(c, hipMemcpyAsync(c->h_metas.p, c->d_metas.p, sizeof(ChangeMeta) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));

  // ---- host: causal schedule, actor ranks, launch plan ----
  auto t_h0 = std::chrono::steady_clock::now();
  int rc = schedule(c);
  if (rc) return rc;
  auto t_h1 = std::chrono::steady_clock::now();

  uint32_t N = (uint32_t)c->n_ops, P = (uint32_t)c->n_preds, NA = (uint32_t)c->actors.size();
  int bits_ctr = bits_for64(c->max_op), bits_actor = bits_for64(NA ? NA - 1 : 0), bits_row = bits_for64(N);
  if (1 + bits_row + bits_ctr + bits_actor > 64) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "sort key wider than 64 bits"); }

  // ---- device buffers ----
  size_t np = c->plans.size();
  if (!c->d_plans.ensure(sizeof(ChangePlan) * std::max<size_t>(np, 1)) || !c->d_amap.ensure(4 * std::max<size_t>(c->amap.size(), 1)) ||
      !c->d_spans.ensure(sizeof(ActorSpan) * std::max<size_t>(c->spans.size(), 1)) || !c->d_tab_off.ensure(4 * (size_t)(NA + 1)))
    return fail(c, AM355_E_NOMEM, "device allocation failed");
  size_t Nc = (size_t)N + 1;
  {
    size_t bytes = 13 * carve_size(Nc, 4) + carve_size(Nc, 1);
    if (!c->d_cols.ensure(bytes) || !c->d_pred.ensure(2 * carve_size((size_t)P + 1, 4))) return fail(c, AM355_E_NOMEM, "device allocation failed (op rows)");
    uint8_t* p = c->d_cols.as<uint8_t>();
    OpCols& o = c->cols;
    o.obj_actor = carve<uint32_t>(p, Nc); o.obj_ctr = carve<uint32_t>(p, Nc); o.key_actor = carve<uint32_t>(p, Nc); o.key_ctr = carve<uint32_t>(p, Nc);
    o.key_off = carve<uint32_t>(p, Nc); o.key_len = carve<uint32_t>(p, Nc); o.action = carve<uint32_t>(p, Nc); o.val_tl = carve<uint32_t>(p, Nc);
    o.val_off = carve<uint32_t>(p, Nc); o.pred_first = carve<uint32_t>(p, Nc); o.pred_num = carve<uint32_t>(p, Nc); o.id_ctr = carve<uint32_t>(p, Nc);
    o.id_actor = carve<uint32_t>(p, Nc); o.insert = carve<uint8_t>(p, Nc);
    uint8_t* q = c->d_pred.as<uint8_t>();
    o.pred_actor = carve<uint32_t>(q, (size_t)P + 1);
    o.pred_ctr = carve<uint32_t>(q, (size_t)P + 1);
  }
  {
    size_t bytes = 10 * carve_size(Nc, 4) + 3 * carve_size(Nc, 8) + carve_size(Nc, 1) + carve_size(2 * Nc, 4) + 4 * carve_size(2 * Nc + 2, 4) +
                   3 * carve_size(Nc + 1, 4) + scan_workspace_bytes((uint32_t)Nc) + 256;
    size_t sort_bytes = 2 * carve_size(Nc, 8) + 2 * carve_size(Nc, 4) + sort_workspace_bytes((uint32_t)Nc) + 256;
    size_t ir_bytes = 5 * carve_size(Nc, 4) + 2 * carve_size(Nc, 4) + carve_size(Nc, 8) + 4 * carve_size(Nc, 4);
    if (!c->d_merge.ensure(bytes) || !c->d_sort.ensure(sort_bytes) || !c->d_ir.ensure(ir_bytes)) return fail(c, AM355_E_NOMEM, "device allocation failed (merge)");
    uint8_t* p = c->d_merge.as<uint8_t>();
    MergeBufs& b = c->mb;
    b.arena = c->d_arena.as<uint8_t>();
    b.ops = c->cols;
    b.n_ops = N; b.n_preds = P; b.n_actors = NA;
    b.actor_tab_off = c->d_tab_off.as<uint32_t>();
    b.spans = c->d_spans.as<ActorSpan>();
    b.bits_ctr = (uint32_t)bits_ctr; b.bits_actor = (uint32_t)bits_actor;
    b.obj_row = carve<uint32_t>(p, Nc); b.ref_row = carve<uint32_t>(p, Nc); b.succ_cnt = carve<uint32_t>(p, Nc); b.inc_cnt = carve<uint32_t>(p, Nc);
    b.val_cnt = carve<uint32_t>(p, Nc); b.obj_index = carve<uint32_t>(p, Nc);
    b.em_row = carve<uint32_t>(p, Nc); b.ins_row = carve<uint32_t>(p, Nc); b.upd_row = carve<uint32_t>(p, Nc); b.next_sib = carve<uint32_t>(p, Nc);
    b.inc_sum = carve<unsigned long long>(p, Nc); b.last_inc = carve<unsigned long long>(p, Nc); b.em_trig = carve<unsigned long long>(p, Nc);
    b.kind = carve<uint8_t>(p, Nc);
    b.first_child = carve<uint32_t>(p, 2 * Nc);
    b.succ_a = carve<uint32_t>(p, 2 * Nc + 2); b.succ_b = carve<uint32_t>(p, 2 * Nc + 2); b.dist_a = carve<uint32_t>(p, 2 * Nc + 2); b.dist_b = carve<uint32_t>(p, 2 * Nc + 2);
    b.order = carve<uint32_t>(p, Nc + 1); b.scan_a = carve<uint32_t>(p, Nc + 1); b.scan_b = carve<uint32_t>(p, Nc + 1);
    b.scan_ws = p;
    b.obj_first_pos = nullptr;  // carved from the IR arena below
    uint8_t* s = c->d_sort.as<uint8_t>();
    b.key_a = carve<uint64_t>(s, Nc); b.key_b = carve<uint64_t>(s, Nc); b.val_a = carve<uint32_t>(s, Nc); b.val_b = carve<uint32_t>(s, Nc);
    b.sort_ws = s;
    b.counts = c->d_counts.as<Counts>();
    uint8_t* r = c->d_ir.as<uint8_t>();
    PatchIR& ir = c->ir;
    ir.obj_make_row = carve<uint32_t>(r, Nc); ir.obj_map_begin = carve<uint32_t>(r, Nc); ir.obj_map_end = carve<uint32_t>(r, Nc);
    ir.obj_edit_begin = carve<uint32_t>(r, Nc); ir.obj_edit_end = carve<uint32_t>(r, Nc);
    ir.m_row = carve<uint32_t>(r, Nc); ir.m_flags = carve<uint32_t>(r, Nc); ir.m_counter = carve<long long>(r, Nc);
    ir.e_row = carve<uint32_t>(r, Nc); ir.e_elem = carve<uint32_t>(r, Nc); ir.e_index = carve<uint32_t>(r, Nc); ir.e_flags = carve<uint32_t>(r, Nc);
    b.obj_first_pos = b.em_row;  // em_row is dead once the map emissions are ordered (lists run afterwards)
  }
  HIPCHK(c, hipMemcpyAsync(c->d_plans.p, c->plans.data(), sizeof(ChangePlan) * np, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_amap.p, c->amap.data(), 4 * c->amap.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_spans.p, c->spans.data(), sizeof(ActorSpan) * c->spans.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_tab_off.p, c->actor_tab_off.data(), 4 * c->actor_tab_off.size(), hipMemcpyHostToDevice, st));

  // ---- stage 1b: column decode ----
  HIPCHK(c, hipEventRecord(c->ev[2], st));
  HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, sizeof(Counts), st));
  launch_decode_columns(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), c->d_plans.as<ChangePlan>(), (uint32_t)np, c->d_amap.as<uint32_t>(),
                        c->cols, &c->d_counts.as<Counts>()->flags, st);
  HIPCHK(c, hipMemcpyAsync(c->h_counts.p, c->d_counts.p, sizeof(Counts), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipEventRecord(c->ev[3], st));
  HIPCHK(c, hipStreamSynchronize(st));
  Counts* hc = c->h_counts.as<Counts>();
  if (hc->flags) {
    c->flags |= hc->flags;
    return fail(c, (hc->flags & (F_OVERFLOW | F_UNSUPPORTED)) ? AM355_E_UNSUPPORTED : AM355_E_INVALID, "malformed columns (flags 0x%x)", hc->flags);
  }

  // ---- stage 2: merge ----
  merge_phase1(c->mb, hc, st);
  HIPCHK(c, hipEventRecord(c->ev[4], st));
  if (hc->flags) {
    c->flags |= hc->flags;
    uint32_t hard = hc->flags & ~(uint32_t)(F_OVERFLOW | F_UNSUPPORTED);
    return fail(c, hard ? AM355_E_INVALID : AM355_E_UNSUPPORTED, "op set rejected (flags 0x%x)", hc->flags);
  }
  merge_phase2(c->mb, c->ir, hc, st);
  HIPCHK(c, hipEventRecord(c->ev[5], st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (hc->flags) {
    c->flags |= hc->flags;
    uint32_t hard = hc->flags & ~(uint32_t)(F_OVERFLOW | F_UNSUPPORTED);
    return fail(c, hard ? AM355_E_INVALID : AM355_E_UNSUPPORTED, "op set rejected (flags 0x%x)", hc->flags);
  }
  c->counts = *hc;
  c->counts.n_objects += 1;  // + _root
  auto t_end = std::chrono::steady_clock::now();

  am355_stats& s = c->stats;
  s.n_changes = n; s.n_applied = c->n_applied; s.n_pending = c->n_pending; s.n_actors = NA; s.n_objects = c->counts.n_objects;
  s.n_heads = (uint32_t)(c->heads.size() / 32); s.n_ops = c->n_ops; s.max_op = c->max_op; s.raw_bytes = c->raw.size();
  s.n_map_values = c->counts.n_map_emit; s.n_list_elems = c->counts.n_list_ins; s.n_edits = c->counts.n_edits;
  s.ir_bytes = (uint64_t)c->counts.n_objects * 20 + (uint64_t)c->counts.n_map_emit * 16 + (uint64_t)c->counts.n_edits * 16;
  (void)hipEventElapsedTime(&s.ms_parse, c->ev[0], c->ev[1]);
  (void)hipEventElapsedTime(&s.ms_decode, c->ev[2], c->ev[3]);
  (void)hipEventElapsedTime(&s.ms_merge, c->ev[3], c->ev[4]);
  (void)hipEventElapsedTime(&s.ms_order, c->ev[4], c->ev[5]);
  s.ms_host_schedule = std::chrono::duration<float, std::milli>(t_h1 - t_h0).count();
  s.ms_total = std::chrono::duration<float, std::milli>(t_end - t_begin).count();
  s.ms_sort = 0;
  c->replayed = true;
  return AM355_OK;
}

extern "C" int am355_get_stats(const am355_ctx* c, am355_stats* out) {
  if (!c || !out) return AM355_E_ARG;
  *out = c->stats;
  return AM355_OK;
}

extern "C" int am355_get_hashes(const am355_ctx* c, uint8_t* out) {
  if (!c || !out) return AM355_E_ARG;
  if (!c->staged || !c->h_metas.p) return AM355_E_STATE;
  const ChangeMeta* metas = (const ChangeMeta*)c->h_metas.p;
  for (uint32_t i = 0; i < c->n_changes; i++) memcpy(out + 32 * i, metas[i].hash, 32);
  return AM355_OK;
}

extern "C" int am355_get_raw(const am355_ctx* c, const uint8_t** arena, const uint64_t** offsets, uint32_t* n) {
  if (!c || !c->staged) return AM355_E_STATE;
  if (arena) *arena = c->raw.data();
  if (offsets) *offsets = c->raw_off.data();
  if (n) *n = c->n_changes;
  return AM355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// IR download + JSON
// ---------------------------------------------------------------------------------------------------------
extern "C" int am355_fetch_ir(am355_ctx* c, am355_patch_ir* out) {
  if (!c) return AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  (void)hipSetDevice(c->device);
  if (!c->ir_fetched) {
    hipStream_t st = c->stream;
    uint32_t N = (uint32_t)c->n_ops, NO = c->counts.n_objects, NM = c->counts.n_map_emit, NE = c->counts.n_edits;
    size_t bytes = 5 * carve_size(NO, 4) + 2 * carve_size(NM, 4) + carve_size(NM, 8) + 4 * carve_size(NE, 4) + 8 * carve_size(N, 4) + 4096;
    if (!c->h_ir.ensure(bytes)) return fail(c, AM355_E_NOMEM, "host allocation failed");
    uint8_t* p = c->h_ir.as<uint8_t>();
    am355_patch_ir& h = c->hir;
    auto pull = [&](const void* dev, size_t count, size_t elem) -> const void* {
      void* dst = p;
      p += carve_size(count, elem);
      if (count) (void)hipMemcpyAsync(dst, dev, count * elem, hipMemcpyDeviceToHost, st);
      return dst;
    };
    h.n_objects = NO; h.n_map = NM; h.n_edits = NE; h.n_rows = N;
    h.obj_make_row = (const uint32_t*)pull(c->ir.obj_make_row, NO, 4);
    h.obj_map_begin = (const uint32_t*)pull(c->ir.obj_map_begin, NO, 4);
    h.obj_map_end = (const uint32_t*)pull(c->ir.obj_map_end, NO, 4);
    h.obj_edit_begin = (const uint32_t*)pull(c->ir.obj_edit_begin, NO, 4);
    h.obj_edit_end = (const uint32_t*)pull(c->ir.obj_edit_end, NO, 4);
    h.m_row = (const uint32_t*)pull(c->ir.m_row, NM, 4);
    h.m_flags = (const uint32_t*)pull(c->ir.m_flags, NM, 4);
    h.m_counter = (const int64_t*)pull(c->ir.m_counter, NM, 8);
    h.e_row = (const uint32_t*)pull(c->ir.e_row, NE, 4);
    h.e_elem = (const uint32_t*)pull(c->ir.e_elem, NE, 4);
    h.e_index = (const uint32_t*)pull(c->ir.e_index, NE, 4);
    h.e_flags = (const uint32_t*)pull(c->ir.e_flags, NE, 4);
    h.row_id_ctr = (const uint32_t*)pull(c->cols.id_ctr, N, 4);
    h.row_id_actor = (const uint32_t*)pull(c->cols.id_actor, N, 4);
    h.row_action = (const uint32_t*)pull(c->cols.action, N, 4);
    h.row_val_tl = (const uint32_t*)pull(c->cols.val_tl, N, 4);
    h.row_val_off = (const uint32_t*)pull(c->cols.val_off, N, 4);
    h.row_key_off = (const uint32_t*)pull(c->cols.key_off, N, 4);
    h.row_key_len = (const uint32_t*)pull(c->cols.key_len, N, 4);
    h.row_obj_index = (const uint32_t*)pull(c->mb.obj_index, N, 4);
    HIPCHK(c, hipStreamSynchronize(st));
    h.max_op = c->max_op;
    h.n_actors = (uint32_t)c->actors.size();
    c->actor_off.assign(1, 0);
    c->actor_bytes.clear();
    for (auto& a : c->actors) {
      c->actor_bytes.insert(c->actor_bytes.end(), a.begin(), a.end());
      c->actor_off.push_back((uint32_t)c->actor_bytes.size());
    }
    h.actor_off = c->actor_off.data();
    h.actor_bytes = c->actor_bytes.data();
    h.n_clock = (uint32_t)c->clock_actor.size();
    h.clock_actor = c->clock_actor.data();
    h.clock_seq = c->clock_seq.data();
    h.n_heads = (uint32_t)(c->heads.size() / 32);
    h.heads = c->heads.data();
    h.pending = c->n_pending;
    h.arena = c->raw.data();
    c->ir_fetched = true;
  }
  if (out) *out = c->hir;
  return AM355_OK;
}

extern "C" int am355_patch_json(am355_ctx* c, const char** json, size_t* len) {
  if (!c) return AM355_E_ARG;
  int rc = am355_fetch_ir(c, nullptr);
  if (rc) return rc;
  std::string err;
  c->json.clear();
  if (!am355::render_patch_json(c->hir, c->json, err)) return fail(c, AM355_E_UNSUPPORTED, "%s", err.c_str());
  if (json) *json = c->json.c_str();
  if (len) *len = c->json.size();
  return AM355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// diagnostics
// ---------------------------------------------------------------------------------------------------------
extern "C" int am355_test_sort(am355_ctx* c, uint64_t* keys, uint32_t* vals, uint32_t n, int key_bits) {
  if (!c) return AM355_E_ARG;
  (void)hipSetDevice(c->device);
  DevBuf ka, kb, va, vb, ws;
  if (!ka.ensure(8 * (size_t)n + 8) || !kb.ensure(8 * (size_t)n + 8) || !va.ensure(4 * (size_t)n + 4) || !vb.ensure(4 * (size_t)n + 4) || !ws.ensure(sort_workspace_bytes(n)))
    return fail(c, AM355_E_NOMEM, "alloc");
  hipStream_t st = c->stream;
  (void)hipMemcpyAsync(ka.p, keys, 8 * (size_t)n, hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(va.p, vals, 4 * (size_t)n, hipMemcpyHostToDevice, st);
  int res = radix_sort_pairs(ka.as<uint64_t>(), va.as<uint32_t>(), kb.as<uint64_t>(), vb.as<uint32_t>(), n, 0, key_bits, ws.p, st);
  (void)hipMemcpyAsync(keys, res ? kb.p : ka.p, 8 * (size_t)n, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(vals, res ? vb.p : va.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st);
  HIPCHK(c, hipStreamSynchronize(st));
  HIPCHK(c, hipGetLastError());
  ka.release(); kb.release(); va.release(); vb.release(); ws.release();
  return AM355_OK;
}

extern "C" int am355_test_scan(am355_ctx* c, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total) {
  if (!c) return AM355_E_ARG;
  (void)hipSetDevice(c->device);
  DevBuf di, dt, ws;
  if (!di.ensure(4 * (size_t)n + 4) || !dt.ensure(4) || !ws.ensure(scan_workspace_bytes(n))) return fail(c, AM355_E_NOMEM, "alloc");
  hipStream_t st = c->stream;
  (void)hipMemcpyAsync(di.p, in, 4 * (size_t)n, hipMemcpyHostToDevice, st);
  exclusive_scan_u32(di.as<uint32_t>(), di.as<uint32_t>(), n, dt.as<uint32_t>(), ws.p, st);
  (void)hipMemcpyAsync(out, di.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(total, dt.p, 4, hipMemcpyDeviceToHost, st);
  HIPCHK(c, hipStreamSynchronize(st));
  di.release(); dt.release(); ws.release();
  return AM355_OK;
}

extern "C" int am355_get_rows(am355_ctx* c, uint32_t* obj_actor, uint32_t* obj_ctr, uint32_t* key_actor, uint32_t* key_ctr, uint32_t* key_off,
                              uint32_t* key_len, uint32_t* action, uint32_t* val_tl, uint32_t* val_off, uint32_t* pred_num, uint32_t* id_ctr,
                              uint32_t* id_actor, uint8_t* insert, uint32_t* succ_cnt) {
  if (!c || !c->replayed) return AM355_E_STATE;
  (void)hipSetDevice(c->device);
  size_t N = c->n_ops;
  hipStream_t st = c->stream;
  auto pull = [&](void* dst, const void* src, size_t bytes) { if (dst && bytes) (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st); };
  pull(obj_actor, c->cols.obj_actor, 4 * N); pull(obj_ctr, c->cols.obj_ctr, 4 * N); pull(key_actor, c->cols.key_actor, 4 * N);
  pull(key_ctr, c->cols.key_ctr, 4 * N); pull(key_off, c->cols.key_off, 4 * N); pull(key_len, c->cols.key_len, 4 * N);
  pull(action, c->cols.action, 4 * N); pull(val_tl, c->cols.val_tl, 4 * N); pull(val_off, c->cols.val_off, 4 * N);
  pull(pred_num, c->cols.pred_num, 4 * N); pull(id_ctr, c->cols.id_ctr, 4 * N); pull(id_actor, c->cols.id_actor, 4 * N);
  pull(insert, c->cols.insert, N); pull(succ_cnt, c->mb.succ_cnt, 4 * N);
  HIPCHK(c, hipStreamSynchronize(st));
  return AM355_OK;
}
