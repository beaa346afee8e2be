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
#include "am355_bigcol.h"
#include "am355_encode.h"
#include "am355_prims.h"
#include "am355_render.h"
#include "am355_host.h"
#include "am355_history.h"
#include "am355_delta.h"
#include "am355_apply.h"
#include "am355_sync.h"
#include "am355_sched.h"
#include "am355_hist.h"
#include "am355_canary.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace am355;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    canary_forget(p, cap);
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256 + (canary_on() ? (128u << 10) : 0u);
    if (hipMalloc(&p, want) != hipSuccess) return false;
    cap = want;
    return true;
  }
  // grows like ensure() but carries the first `keep` bytes over (device-to-device copy, synchronous)
  bool ensure_keep(size_t bytes, size_t keep) {
    if (bytes <= cap) return true;
    size_t want = bytes + bytes / 2 + 256;
    void* q = nullptr;
    if (hipMalloc(&q, want) != hipSuccess) return false;
    if (p && keep && hipMemcpy(q, p, keep, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(q); return false; }
    canary_forget(p, cap);
    if (p) (void)hipFree(p);
    p = q;
    cap = want;
    return true;
  }
  void release() {
    canary_forget(p, cap);
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

// Byte vector in pinned host memory (the raw arena: the H2D copy of pageable memory is a synchronous bounce through the
// driver's own staging buffer).
struct PinnedBytes {
  uint8_t* p = nullptr;
  size_t n = 0, cap = 0;
  ~PinnedBytes() { if (p) (void)hipHostFree(p); }
  uint8_t* data() { return p; }
  const uint8_t* data() const { return p; }
  size_t size() const { return n; }
  void clear() { n = 0; }
  uint8_t& back() { return p[n - 1]; }
  void reserve(size_t want) {
    if (want <= cap) return;
    size_t c2 = std::max(want + want / 8 + 4096, cap * 2);
    void* q = nullptr;
    if (hipHostMalloc(&q, c2, hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
    if (n) memcpy(q, p, n);
    if (p) (void)hipHostFree(p);
    p = (uint8_t*)q;
    cap = c2;
  }
  void resize(size_t want) { reserve(want); n = want; }
  void push_back(uint8_t b) { reserve(n + 1); p[n++] = b; }
  void append(const uint8_t* a, const uint8_t* b) { size_t k = (size_t)(b - a); reserve(n + k); if (k) memcpy(p + n, a, k); n += k; }
};

// A few persistent host threads for the byte-shovelling around the device work: gather of the change buffers into the pinned
// arena (+ the H2D copy of each slice), raw-DEFLATE of compressed changes / document columns, the document checksum.
// run(k, fn) executes fn(0..k-1), one index per worker at a time, and returns when all are done.
class HostPool {
 public:
  explicit HostPool(unsigned n) {
    for (unsigned i = 0; i < n; i++) workers_.emplace_back([this]() { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    hot_until_.store(0);
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  unsigned size() const { return (unsigned)workers_.size(); }
  // Wakes the workers ahead of a run(): for the next `us` microseconds they poll for work instead of sleeping on the condition
  // variable (waking 31 sleeping threads costs ~100 us -- longer than copying 13 MB with them). Called at the start of a C-ABI
  // call whose serial preamble gives them time to arrive.
  void prewake(unsigned n_workers, unsigned us = 300) {
    if (workers_.empty()) return;
    hot_until_.store(now_us() + us, std::memory_order_release);
    wake(n_workers);
  }
  void run(unsigned k, const std::function<void(unsigned)>& fn) {
    if (k == 0) return;
    // (without workers: highest index first -- task 0 of the staging jobs waits for the others)
    if (k == 1 || workers_.empty()) { for (unsigned i = k; i-- > 0;) fn(i); return; }
    {
      std::lock_guard<std::mutex> l(m_);
      fn_ = &fn; total_ = k; done_ = 0;
      uint64_t g = gen_.load(std::memory_order_relaxed) + 1;
      next_.store(g << 32, std::memory_order_relaxed);  // (generation | next index: a worker that arrives late must not draw an index of a later run)
      gen_.store(g, std::memory_order_release);
    }
    wake(k - 1);  // (only as many workers as there is work: waking all of them costs more than a short job takes)
    work(fn, k, gen_.load(std::memory_order_relaxed));  // the caller works too
    std::unique_lock<std::mutex> l(m_);
    cv_done_.wait(l, [&]() { return done_ == total_; });
    fn_ = nullptr;
  }
 private:
  void wake(unsigned n) {
    if (n >= workers_.size()) cv_.notify_all();
    else for (unsigned i = 0; i < n; i++) cv_.notify_one();
  }
  static uint64_t now_us() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void work(const std::function<void(unsigned)>& fn, unsigned total, uint64_t gen) {
    unsigned mine = 0;
    for (;;) {
      uint64_t v = next_.load(std::memory_order_acquire);
      if ((v >> 32) != (gen & 0xffffffffull) || (uint32_t)v >= total) break;
      if (!next_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) continue;
      fn((unsigned)(uint32_t)v);
      mine++;
    }
    if (mine) {
      std::lock_guard<std::mutex> l(m_);
      done_ += mine;
      if (done_ == total_) cv_done_.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(unsigned)>* f = nullptr;
      unsigned total = 0;
      {
        std::unique_lock<std::mutex> l(m_);
        for (;;) {
          if (stop_) return;
          uint64_t g = gen_.load(std::memory_order_acquire);
          if (g != seen && fn_) { seen = g; f = fn_; total = total_; break; }
          if (now_us() < hot_until_.load(std::memory_order_acquire)) {  // hot: poll without the lock
            l.unlock();
            for (int k = 0; k < 64; k++) {
#if defined(__x86_64__)
              __builtin_ia32_pause();
#endif
            }
            l.lock();
            continue;
          }
          cv_.wait(l);
        }
      }
      work(*f, total, seen);
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, cv_done_;
  const std::function<void(unsigned)>* fn_ = nullptr;
  std::atomic<uint64_t> next_{0};
  unsigned total_ = 0, done_ = 0;
  std::atomic<uint64_t> gen_{0}, hot_until_{0};
  bool stop_ = false;
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
  hipStream_t stream = nullptr;   // decode / merge critical path
  hipStream_t stream2 = nullptr;  // SHA-256 + dependency resolution, off the critical path
  hipStream_t stream3 = nullptr;  // second decoder class, side by side with the first
  hipStream_t stream4 = nullptr;  // small copies that must not queue behind kernels or fills (digests to the host, host-built tables to HBM)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev[8] = {};
  hipEvent_t ev_parse = nullptr, ev_b0 = nullptr, ev_b1 = nullptr;
  hipEvent_t ev_counts = nullptr, ev_runs = nullptr;  // merge stage: counter read-backs that do not drain the stream
  std::string err;
  uint32_t flags = 0;

  // staged batch
  PinnedBytes raw;                 // uncompressed changes, host copy in pinned memory (the scheduler reads deps / actor ids here)
  std::unique_ptr<HostPool> pool;  // host worker threads (staging, inflate, checksum)
  std::vector<uint64_t> raw_off;
  uint32_t n_changes = 0;
  bool staged = false, replayed = false, ir_fetched = false;
  bool has_unknown_cols = false;     // some change carries columns outside the modelled set (kept by the reference's save)
  bool staging_in_flight = false;    // am355_load_changes returned with its H2D copies still running on `stream`
  bool is_document = false;          // staged input is one saved document (am355_load_document) rather than changes
  ChangeMeta doc_meta{};             // column layout of the staged document inside `raw`
  std::vector<uint32_t> doc_actor_rank;  // document actor index -> lexicographic rank
  DevBuf d_arena, d_offsets, d_metas;
  HostBuf h_metas, h_offsets, h_sig;   // h_sig: HostSignals (device -> host result words without a copy)
  uint32_t sig_seq = 0;
  hipEvent_t ev_s1 = nullptr;          // the per-change digests (briefs) have arrived on the host
  hipEvent_t ev_fills = nullptr;       // merge fills done (when they run on stream4)
  hipEvent_t ev_sched = nullptr;   // host copies of the device scheduler's order / pass numbers complete (stream4)
  hipEvent_t ev_plan = nullptr, ev_tables = nullptr;  // k_plan_apply done (stream4 copies the digests behind it) | host-built tables in HBM
  void* counts_zeroed_at = nullptr; size_t counts_zeroed = 0;  // the counter block was cleared beside stage 1 (address, bytes)
  // AM355_PHASE_EVENTS=1: HIP events between the phases of a change replay (am355_stats.ms_parse / ms_decode / ms_merge / ms_order). Off by
  // default: every event record between two kernels of the main stream is a packet of its own in front of the next dispatch.
  bool phase_events = false;
  bool inline_fills = true;  // AM355_STAGE1_FILLS=stream: the fills of stage 1 as memsets on stream3 (round-2 form) instead of inside k_parse_changes
  DevBuf d_big, d_bigvals, d_ks;     // document load: token / record index, column values, keyStr run table
  HostBuf h_biginfo;
  BigColDesc doc_cols{};
  bool doc_serial = false;           // AM355_DOC_SERIAL=1: lane-serial column decoders (first version, kept for cross-checks)
  // stage-1 side tables (device) and their pinned host mirrors
  DevBuf d_entries, d_amap_base, d_amap_prov, d_slots, d_first_idx, d_hashes, d_hash_tab, d_min_idx, d_has_dep, d_words, d_slot_rank, d_scan1, d_plan_sums, d_dep_idx, d_self_idx, d_rank_ids;
  HostBuf h_dep_idx, h_self_idx, h_amap, h_amap_base;   // general scheduler: dependency / duplicate indexes and actor tables resolved on the device
  HostBuf h_slots, h_hashes, h_has_dep, h_words, h_stage, h_s1;
  DevBuf d_s1;                 // stage-1 results read by the host: flag words | distinct actor ids | one ChangeBrief per change
  ChangeBrief* hp_briefs = nullptr;
  uint32_t* hp_distinct = nullptr;
  bool have_host_metas = false;
  uint32_t amap_cap = 0, slot_mask = 0, hash_mask = 0;
  bool used_fast_path = false;

  // schedule
  std::vector<ChangePlan> plans;
  std::vector<uint32_t> applied_change, applied_op_base;  // applied changes in application order (plans get regrouped by decoder class)
  std::vector<uint8_t> doc_bytes;                          // the loaded document as given (Backend.save of an unchanged document returns it)
  std::vector<uint8_t> saved;                              // result of am355_save
  HistoryOutput history;                                   // result of am355_doc_changes
  bool history_ok = false; uint32_t history_flags = 0;
  std::vector<std::vector<uint8_t>> inflate_scratch;       // am355_load_document: inflated columns, longest first (capacity kept between loads)
  std::vector<uint32_t> doc_col_rows;                      // loaded document: values per op column (BigCol order), parallel decode only
  std::vector<std::pair<uint32_t, std::vector<uint8_t>>> doc_chg_cols;  // loaded document: change-metadata columns, inflated
  std::vector<uint8_t> doc_tail;                           // loaded document: headsIndexes + extraBytes
  bool doc_other_ops_cols = false;                         // loaded document has non-empty op columns outside the modelled set
  DevBuf d_save, d_enc, d_encout;
  HostBuf h_encout;
  std::vector<uint32_t> amap;
  std::vector<ActorSpan> spans;
  std::vector<uint32_t> actor_tab_off;
  std::vector<std::string> actors;        // by rank
  std::vector<uint32_t> clock_actor;      // first-applied order
  std::vector<uint64_t> clock_seq;
  std::vector<uint8_t> heads;
  uint32_t n_applied = 0, n_pending = 0;
  uint64_t n_ops = 0, n_preds = 0, max_op = 0;
  DevBuf d_plans, d_amap, d_tables;   // d_tables: plans | actor spans | span offsets | slot ranks or actor tables (run_device)
  ActorSpan* p_spans = nullptr;
  uint32_t* p_tab_off = nullptr;

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

  // incremental applyChanges (am355_apply_changes)
  std::vector<uint32_t> pending_change;   // queued changes (input indexes, queue order) after the last replay
  std::vector<uint32_t> pass_first_row;   // first op row of every scheduling pass after the first (general scheduler)
  DevBuf d_delta, d_pass, d_delta_edit, d_sched, d_hist;
  HostBuf h_sched;
  bool device_scheduled = false;   // the last general-path replay was scheduled by the device (am355_sched.hip)
  HostBuf h_delta;
  DeltaBufs delta{};
  ApplyPatch apply;
  bool apply_ready = false;
  std::vector<uint32_t> dep_first, dep_index;   // am355_get_dep_graph
  bool dep_graph_ready = false;
  DevBuf d_sync;                                // am355_sync_bloom_*: index list, filter bits, flags
  // am355_apply_changes: where the op streams of the calls so far began (a call of applyChanges, a scheduling pass of one) -- the
  // reference's merge calls never cross them --, whether that record is complete, and whether some call skipped values of a property
  // that holds a child object (then objectMeta.children of the reference differs from the visible values: delta_key_history)
  std::vector<uint32_t> stream_breaks;
  bool breaks_exact = true, children_hazard = false, in_apply = false;
  bool no_history = false;  // the staged changes are the rebuilt history of a LOADED document: the reference's objectMeta came from one pass over the document
  DevBuf d_breaks;
  bool state_checked = false;  // the state was built by am355_apply_changes calls (each checked for what later patches depend on) or is empty
  std::string apply_json;

  // objectId sharding (am355_set_shard): this context merges the objects rank `shard_rank` of `shard_world` owns
  uint32_t shard_rank = 0, shard_world = 1;
  std::vector<uint8_t> stitched;  // am355_import_fragments: the combined record tables
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


// No C++ exception may cross the C ABI (an escaped std::bad_alloc would terminate the host process: untrusted input must end in
// an error code, as the reference ends in a catchable exception).
template <class F>
static int guarded(am355_ctx* c, F body) {
  try {
    int rc = body();
    if (canary_on() && c) {  // AM355_CANARY=1 (am355_canary.h): did a kernel of this call write past one of its arrays?
      char msg[320];
      if (!canary_check(msg, sizeof msg)) return fail(c, AM355_E_DEVICE, "%s", msg);
    }
    return rc;
  } catch (const std::bad_alloc&) {
    return c ? fail(c, AM355_E_NOMEM, "out of host memory") : AM355_E_NOMEM;
  } catch (const std::exception& e) {
    return c ? fail(c, AM355_E_DEVICE, "internal error: %s", e.what()) : AM355_E_DEVICE;
  }
}

extern "C" am355_ctx* am355_create(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return nullptr;
  // every replay has a few host round trips of some microseconds each: wait for them actively (refused, harmlessly, when the
  // host process has already initialised the device with other flags)
  (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  am355_ctx* c = new am355_ctx();
  c->device = device;
  {
    unsigned hw = std::thread::hardware_concurrency();
    const char* env = getenv("AM355_HOST_THREADS");
    unsigned want = env && atoi(env) > 0 ? (unsigned)atoi(env) : std::min(hw ? hw : 4u, 64u);  // (measured on a 2 x 64-core host: inflating 4 k changes scales to ~64 threads; jobs wake only the workers they need)
    c->pool.reset(new HostPool(want > 1 ? want - 1 : 0));  // (the calling thread works too)
  }
  // the decode/merge stream outranks the hash stream: their small grids would otherwise share SIMDs and the
  // ALU-dense SHA-256 waves slow the latency-bound parse/decode waves down
  int prio_low = 0, prio_high = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
  if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_high) != hipSuccess) { delete c; return nullptr; }
  {
    // The hash stream (one lane per change: ~65 waves of dependent SHA-256 rounds for 4 k changes) gets its own few compute
    // units when AM355_HASH_CUS=n asks for it (hipExtStreamCreateWithCUMask: its waves then never share a SIMD with the
    // latency-bound kernels of the critical path); by default it is an ordinary low-priority stream.
    const char* env = getenv("AM355_HASH_CUS");
    int want = env ? atoi(env) : 0;
    bool made = false;
    if (want > 0) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > want) {
        int n_cu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
        for (int k = 0; k < want; k++) { int cu = n_cu - 1 - k; mask[(size_t)cu / 32] |= 1u << (cu % 32); }
        made = hipExtStreamCreateWithCUMask(&c->stream2, (uint32_t)mask.size(), mask.data()) == hipSuccess;
      }
    }
    if (!made && hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_low) != hipSuccess) { delete c; return nullptr; }
  }
  if (hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_high) != hipSuccess) { delete c; return nullptr; }
  if (hipStreamCreateWithPriority(&c->stream4, hipStreamNonBlocking, prio_high) != hipSuccess) { delete c; return nullptr; }
  if (hipEventCreate(&c->ev_fork) != hipSuccess || hipEventCreate(&c->ev_join) != hipSuccess) { delete c; return nullptr; }
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete c; return nullptr; }
  if (hipEventCreate(&c->ev_parse) != hipSuccess || hipEventCreate(&c->ev_b0) != hipSuccess || hipEventCreate(&c->ev_b1) != hipSuccess) { delete c; return nullptr; }
  if (hipEventCreate(&c->ev_counts) != hipSuccess || hipEventCreate(&c->ev_runs) != hipSuccess || hipEventCreate(&c->ev_s1) != hipSuccess) { delete c; return nullptr; }
  if (hipEventCreate(&c->ev_plan) != hipSuccess || hipEventCreate(&c->ev_tables) != hipSuccess || hipEventCreate(&c->ev_fills) != hipSuccess) { delete c; return nullptr; }
  if (hipEventCreate(&c->ev_sched) != hipSuccess) { delete c; return nullptr; }
  if (!c->h_sig.ensure(sizeof(HostSignals))) { delete c; return nullptr; }
  memset(c->h_sig.p, 0, sizeof(HostSignals));
  if (const char* e = getenv("AM355_PHASE_EVENTS")) c->phase_events = strcmp(e, "0") != 0;
  if (const char* e = getenv("AM355_STAGE1_FILLS")) c->inline_fills = strcmp(e, "stream") != 0;
  return c;
}

extern "C" void am355_destroy(am355_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->stream2);
  (void)hipStreamSynchronize(c->stream3);
  if (c->stream4) (void)hipStreamSynchronize(c->stream4);
  for (DevBuf* b : {&c->d_entries, &c->d_amap_base, &c->d_amap_prov, &c->d_slots, &c->d_first_idx, &c->d_hashes, &c->d_hash_tab, &c->d_min_idx, &c->d_has_dep,
                    &c->d_words, &c->d_slot_rank, &c->d_scan1, &c->d_plan_sums, &c->d_dep_idx, &c->d_self_idx, &c->d_rank_ids})
    b->release();
  for (HostBuf* b : {&c->h_slots, &c->h_hashes, &c->h_has_dep, &c->h_words, &c->h_stage, &c->h_s1, &c->h_dep_idx, &c->h_self_idx, &c->h_amap, &c->h_amap_base}) b->release();
  c->d_s1.release();
  for (hipEvent_t e : {c->ev_parse, c->ev_b0, c->ev_b1, c->ev_counts, c->ev_runs, c->ev_s1, c->ev_plan, c->ev_tables, c->ev_fills, c->ev_sched})
    if (e) (void)hipEventDestroy(e);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream4) (void)hipStreamDestroy(c->stream4);
  for (hipEvent_t e : {c->ev_fork, c->ev_join})
    if (e) (void)hipEventDestroy(e);
  c->d_delta.release(); c->d_delta_edit.release(); c->d_sched.release(); c->h_sched.release(); c->d_hist.release(); c->d_pass.release(); c->d_breaks.release(); c->h_delta.release(); c->d_sync.release();
  for (DevBuf* b : {&c->d_arena, &c->d_offsets, &c->d_metas, &c->d_plans, &c->d_amap, &c->d_tables, &c->d_cols, &c->d_pred,
                    &c->d_merge, &c->d_sort, &c->d_ir, &c->d_counts, &c->d_big, &c->d_bigvals, &c->d_ks, &c->d_save, &c->d_enc, &c->d_encout})
    b->release();
  for (HostBuf* b : {&c->h_metas, &c->h_offsets, &c->h_sig, &c->h_counts, &c->h_ir, &c->h_rows, &c->h_biginfo, &c->h_encout}) b->release();
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

constexpr size_t INFLATE_CAP = 0xfff00000ull;  // one staged batch / document is addressed with 32-bit arena offsets

// keep_staged: the changes staged so far stay where they are -- in the pinned arena and in HBM -- and the batch goes behind them
// (am355_apply_changes onto a state whose changes were all applied in the order they are staged: only the batch crosses the link)
static int load_changes_impl(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n, bool keep_staged = false) {
  if (!c || (!arena && n) || !offsets) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  (void)hipSetDevice(c->device);
  const uint32_t k0 = keep_staged ? c->n_changes : 0;  // changes and bytes kept in front of the batch
  const size_t b0 = keep_staged ? c->raw.size() : 0;
  if (c->staging_in_flight) { c->staging_in_flight = false; HIPCHK(c, hipStreamSynchronize(c->stream)); }  // (copies of the previous batch still read the pinned arena)
  c->staged = c->replayed = c->ir_fetched = false;
  c->apply_ready = false;
  c->state_checked = false;
  c->is_document = false;
  c->flags = 0;
  if (n && offsets[n] - offsets[0] >= ((uint64_t)1 << 20))
    c->pool->prewake(offsets[1] - offsets[0] > 9 && arena[offsets[0] + 8] == 2 ? c->pool->size() : 4);  // (compressed changes: every thread inflates)
  for (uint32_t i = 0; i < n; i++)
    if (offsets[i] > offsets[i + 1]) return fail(c, AM355_E_ARG, "change offsets must be ascending (offsets[%u] > offsets[%u])", i, i + 1);
  if (offsets[n] - offsets[0] + b0 >= INFLATE_CAP) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "batch larger than 4 GiB (32-bit arena offsets)"); }
  // ---- gather into the pinned raw arena + H2D, in slices handled by the host pool ----
  // Slice k covers a contiguous run of changes of about equal bytes. Phase A (parallel): changes of chunk type 2 are inflated
  // and their uncompressed containers rebuilt (columnar.js:813-823; checksum / hash are over that form) into a slice-local
  // buffer; a slice without compressed changes has nothing to do. Then the slice sizes are summed (host, O(slices)) and
  // phase B (parallel) copies every slice to its place in the pinned arena, fills its offsets and enqueues its H2D copy, so
  // that the DMA engine works on early slices while the host threads are still gathering later ones.
  const size_t in_bytes = (size_t)(offsets[n] - offsets[0]);
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "load_changes: %-22s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  unsigned n_slices = 1;
  // (AM355_SLICE_BYTES: bytes per slice, 2 MiB by default -- every H2D copy has a fixed cost of some microseconds --; the tests lower it to exercise the sliced path on small inputs)
  const char* slice_env = getenv("AM355_SLICE_BYTES");
  const size_t slice_bytes = slice_env && atol(slice_env) > 0 ? (size_t)atol(slice_env) : (size_t)1 << 21;
  // compressed changes (chunk type 2; the first and the middle change are taken as representative) are inflated slice by slice on
  // the host threads: many small slices keep all of them busy (zlib runs at a few hundred MB/s per thread)
  const bool deflated = n && ((offsets[1] - offsets[0] > 9 && arena[offsets[0] + 8] == 2) || (offsets[n / 2 + 1] - offsets[n / 2] > 9 && arena[offsets[n / 2] + 8] == 2));
  const size_t per_slice = deflated && !slice_env ? (size_t)64 << 10 : slice_bytes;
  if (in_bytes >= 2 * per_slice && n >= 16)
    n_slices = (unsigned)std::min<size_t>({(size_t)(c->pool->size() + 1) * (deflated ? 8 : 2), in_bytes / per_slice, (size_t)n / 8});
  if (n_slices < 1) n_slices = 1;
  struct Slice { uint32_t c0 = 0, c1 = 0; size_t out_bytes = 0, base = 0; bool any_deflated = false; int err = 0; uint32_t err_change = 0; std::vector<uint8_t> tmp; std::vector<uint32_t> tmp_len; };
  std::vector<Slice> slices(n_slices);
  {
    uint32_t ci = 0;
    for (unsigned k = 0; k < n_slices; k++) {
      slices[k].c0 = ci;
      uint64_t target = offsets[0] + (uint64_t)in_bytes * (k + 1) / n_slices;
      while (ci < n && (k + 1 == n_slices || offsets[ci + 1] <= target)) ci++;
      slices[k].c1 = ci;
    }
    slices[n_slices - 1].c1 = n;
  }
  auto phase_a = [&](unsigned k) {
    Slice& sl = slices[k];
    for (uint32_t i = sl.c0; i < sl.c1; i++) {
      size_t len = (size_t)(offsets[i + 1] - offsets[i]);
      if (len > 9 && arena[offsets[i] + 8] == 2) { sl.any_deflated = true; break; }
    }
    if (!sl.any_deflated) { sl.out_bytes = (size_t)(offsets[sl.c1] - offsets[sl.c0]); return; }
    sl.tmp_len.resize(sl.c1 - sl.c0);
    std::vector<uint8_t> out;
    for (uint32_t i = sl.c0; i < sl.c1 && !sl.err; i++) {
      const uint8_t* p = arena + offsets[i];
      size_t len = (size_t)(offsets[i + 1] - offsets[i]);
      size_t before = sl.tmp.size();
      if (len > 9 && p[8] == 2) {
        size_t off = 9;
        uint64_t clen;
        if (!read_uleb_host(p, len, off, clen) || clen > len - off) { sl.err = 10; sl.err_change = i; break; }
        int irc = inflate_raw(p + off, (size_t)clen, out, INFLATE_CAP);
        if (irc) { sl.err = irc; sl.err_change = i; break; }
        sl.tmp.insert(sl.tmp.end(), p, p + 8);
        sl.tmp.push_back(1);
        uint64_t v = out.size();
        do { uint8_t x = v & 0x7f; v >>= 7; if (v) x |= 0x80; sl.tmp.push_back(x); } while (v);
        sl.tmp.insert(sl.tmp.end(), out.begin(), out.end());
      } else {
        sl.tmp.insert(sl.tmp.end(), p, p + len);
      }
      if (sl.tmp.size() >= INFLATE_CAP) { sl.err = 2; sl.err_change = i; break; }
      sl.tmp_len[i - sl.c0] = (uint32_t)(sl.tmp.size() - before);
    }
    sl.out_bytes = sl.tmp.size();
  };
  c->pool->run(n_slices, phase_a);
  lap("inflate / sizes");
  size_t total = 0;
  for (Slice& sl : slices) {
    if (sl.err == 10) { c->flags |= AM355_F_BAD_CHUNK; return fail(c, AM355_E_INVALID, "change %u: bad deflate container", sl.err_change); }
    if (sl.err == 3) return fail(c, AM355_E_NOMEM, "inflate: out of memory");
    if (sl.err == 2) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "change %u: inflated size beyond the 4 GiB batch limit", sl.err_change); }
    if (sl.err) { c->flags |= AM355_F_BAD_DEFLATE; return fail(c, AM355_E_INVALID, "change %u: invalid or truncated deflate data", sl.err_change); }
    sl.base = total;
    total += sl.out_bytes;
    if (total + b0 >= INFLATE_CAP) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "batch larger than 4 GiB (32-bit arena offsets)"); }
  }
  const uint32_t n_all = k0 + n;
  c->raw.resize(b0 + total);
  c->raw_off.resize((size_t)n_all + 1);
  c->raw_off[n_all] = b0 + total;
  c->n_changes = n_all;
  if (!(keep_staged ? c->d_arena.ensure_keep(b0 + total + 64, b0) : c->d_arena.ensure(total + 64)) || !c->d_offsets.ensure(sizeof(uint64_t) * ((size_t)n_all + 1)) ||
      !c->d_metas.ensure(sizeof(ChangeMeta) * (size_t)std::max(n_all, 1u)) || !c->h_metas.ensure(sizeof(ChangeMeta) * (size_t)std::max(n_all, 1u)) ||
      !c->d_counts.ensure(sizeof(Counts)) || !c->h_counts.ensure(2 * sizeof(Counts)) || !c->h_offsets.ensure(sizeof(uint64_t) * ((size_t)n_all + 1)))
    return fail(c, AM355_E_NOMEM, "device allocation failed");
  // (from here on `raw`, `d_raw` and `roff` address the batch's part: byte b0 of the arena, entry k0 of the offsets)
  uint8_t* raw = c->raw.data() + b0;
  uint8_t* d_raw = c->d_arena.as<uint8_t>() + b0;
  uint64_t* roff = c->raw_off.data() + k0;
  std::vector<hipError_t> h2d(n_slices, hipSuccess);
  bool any_deflated = false;
  for (const Slice& sl : slices) any_deflated = any_deflated || sl.any_deflated;
  const char* gather_env = getenv("AM355_GATHER_UNIT");  // (tests: bytes per copy unit, lowered to run the grouped gather on small inputs)
  if (!any_deflated && (total >= ((size_t)4 << 20) || (gather_env && total > 0))) {
    // Plain changes: the arena is one contiguous copy of the input, pageable -> pinned by host threads, pinned -> HBM by the DMA
    // engine, pipelined. Measured on the EPYC 9575F host (tools/micro/pinned_memcpy.cpp, profiles/r02_ab_staging_*): one thread
    // copies 16 MiB into pinned memory in 0.27 ms, one 16 MiB H2D command takes 0.30 ms (56 GB/s: the link), every H2D command
    // costs ~10 us whatever its size, and a sleeping pool thread needs ~0.1 ms to start working. So: the CALLING thread starts
    // copying at once and is the one that enqueues; units of 256 KiB are drawn from a shared counter by the caller and four pool
    // threads (a thread's FIRST unit runs at a fraction of the later rate: cold source lines); the first DMA command goes out
    // after one unit, every following one covers twice as much, up to 8 MiB -- few commands, none waiting for its bytes.
    const size_t unit = gather_env && atol(gather_env) > 0 ? (size_t)atol(gather_env) : (size_t)256 << 10;
    const size_t n_units = (total + unit - 1) / unit;
    std::vector<uint32_t> group_of(n_units);
    std::vector<size_t> group_first;  // first unit of each group (+ end)
    for (size_t u = 0, span = 1; u < n_units; span = std::min<size_t>(span * 2, 64)) {
      group_first.push_back(u);
      for (size_t k = 0; k < span && u < n_units; k++, u++) group_of[u] = (uint32_t)group_first.size() - 1;
    }
    const size_t n_groups = group_first.size();
    group_first.push_back(n_units);
    std::vector<std::atomic<uint32_t>> left(n_groups);
    for (size_t g = 0; g < n_groups; g++) left[g].store((uint32_t)(group_first[g + 1] - group_first[g]));
    h2d.assign(n_groups, hipSuccess);
    const uint8_t* src = arena + offsets[0];
    const uint64_t off0 = offsets[0];
    std::atomic<size_t> next_unit{0};
    const unsigned n_helpers = (unsigned)std::min<size_t>(4, std::min<size_t>(n_units > 1 ? n_units - 1 : 0, c->pool->size()));
    lap("  buffers ready");
    c->pool->run(n_helpers + 1, [&](unsigned task) {
      const bool issuer = task == 0;  // (the calling thread: it draws the first index)
      size_t next_group = 0;
      auto issue_ready = [&]() {
        while (next_group < n_groups && left[next_group].load(std::memory_order_acquire) == 0) {
          size_t gb = group_first[next_group] * unit, ge = std::min(total, group_first[next_group + 1] * unit);
          h2d[next_group] = hipMemcpyAsync(d_raw + gb, raw + gb, ge - gb, hipMemcpyHostToDevice, c->stream);
          if (trace) fprintf(stderr, "load_changes:   group %zu (%zu KiB) enqueued +%8.3f ms\n", next_group, (ge - gb) >> 10, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
          next_group++;
        }
      };
      if (issuer) (void)hipSetDevice(c->device);
      for (;;) {
        size_t u = next_unit.fetch_add(1, std::memory_order_relaxed);
        if (u >= n_units) break;
        size_t b = u * unit, e = std::min(total, b + unit);
        memcpy(raw + b, src + b, e - b);
        left[group_of[u]].fetch_sub(1, std::memory_order_acq_rel);
        if (issuer) issue_ready();
      }
      if (issuer) {
        while (next_group < n_groups) {
          issue_ready();
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
      }
    });
    for (uint32_t i = 0; i < n; i++) roff[i] = b0 + (offsets[i] - off0);
  } else {
    // slices (inflated or plain) to their place in the arena in parallel; the H2D copies go out in few large commands: consecutive
    // slices are grouped to >= 2 MiB and one thread (task 0) enqueues a group as soon as its slices have landed
    std::vector<uint32_t> group_of(n_slices);
    std::vector<size_t> group_begin{0};
    {
      size_t acc = 0;
      for (unsigned k = 0; k < n_slices; k++) {
        group_of[k] = (uint32_t)group_begin.size() - 1;
        acc += slices[k].out_bytes;
        if (acc >= ((size_t)2 << 20) && k + 1 < n_slices) { group_begin.push_back(slices[k + 1].base); acc = 0; }
      }
      group_begin.push_back(total);
    }
    const size_t n_groups = group_begin.size() - 1;
    std::vector<std::atomic<uint32_t>> left(n_groups);
    for (auto& x : left) x.store(0);
    for (unsigned k = 0; k < n_slices; k++) left[group_of[k]].fetch_add(1);
    h2d.assign(n_groups, hipSuccess);
    c->pool->run(n_slices + 1, [&](unsigned task) {
      if (task == 0) {
        (void)hipSetDevice(c->device);
        for (size_t g = 0; g < n_groups; g++) {
          while (left[g].load(std::memory_order_acquire) != 0) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
          }
          size_t gb = group_begin[g], ge = group_begin[g + 1];
          if (ge > gb) h2d[g] = hipMemcpyAsync(d_raw + gb, raw + gb, ge - gb, hipMemcpyHostToDevice, c->stream);
        }
        return;
      }
      Slice& sl = slices[task - 1];
      if (sl.any_deflated) {
        if (sl.out_bytes) memcpy(raw + sl.base, sl.tmp.data(), sl.out_bytes);
        size_t o = b0 + sl.base;
        for (uint32_t i = sl.c0; i < sl.c1; i++) { roff[i] = o; o += sl.tmp_len[i - sl.c0]; }
      } else {
        if (sl.out_bytes) memcpy(raw + sl.base, arena + offsets[sl.c0], sl.out_bytes);
        for (uint32_t i = sl.c0; i < sl.c1; i++) roff[i] = b0 + sl.base + (offsets[i] - offsets[sl.c0]);
      }
      left[group_of[task - 1]].fetch_sub(1, std::memory_order_acq_rel);
    });
  }
  lap("gathered, H2D enqueued");
  for (hipError_t e : h2d)
    if (e != hipSuccess) return fail(c, AM355_E_DEVICE, "hipMemcpyAsync (arena): %s", hipGetErrorString(e));
  memcpy(c->h_offsets.p, c->raw_off.data(), sizeof(uint64_t) * ((size_t)n_all + 1));  // (pinned mirror: the copy below must not bounce through the driver)
  HIPCHK(c, hipMemcpyAsync(c->d_offsets.p, c->h_offsets.p, sizeof(uint64_t) * ((size_t)n_all + 1), hipMemcpyHostToDevice, c->stream));
  // No wait here: am355_replay enqueues behind these copies on the same stream, so its host-side set-up runs beside the tail of
  // the DMA instead of after a wake-up. The pinned arena is only rewritten by the next load, which waits first.
  c->staging_in_flight = true;
  static const bool stage_sync = getenv("AM355_STAGE_SYNC") != nullptr;  // (diagnostic: wait for the copies here, as round 1 did)
  if (trace || stage_sync) { c->staging_in_flight = false; HIPCHK(c, hipStreamSynchronize(c->stream)); lap("H2D done"); }
  c->staged = true;
  c->stats = am355_stats{};
  c->stats.n_changes = n_all;
  c->stats.raw_bytes = c->raw.size();
  return AM355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// document staging (Backend.load): columnar.js:1006-1038 decodeDocumentHeader, 1062-1067 inflateColumn,
// new.js:1645-1675 readDocumentChanges.  Host work is the container/header parse, the chunk checksum (one SHA-256
// over the whole chunk is sequential by construction), zlib inflate of the columns and the scan of the change
// metadata (clock); the op columns go to HBM for the device decode + patch.
// ---------------------------------------------------------------------------------------------------------
namespace {
// host-side RLE-uint / delta reader for the (small) change-metadata columns
struct HostRle {
  const uint8_t* p; size_t len, off = 0; int64_t count = 0; int state = 0; int64_t last = 0; bool last_null = true; bool is_signed;
  HostRle(const uint8_t* p_, size_t l, bool sg) : p(p_), len(l), is_signed(sg) {}
  bool done() const { return count == 0 && off >= len; }
  bool leb(bool sg, int64_t& out) {
    uint64_t v = 0; int shift = 0;
    while (off < len && shift < 64) {
      uint8_t b = p[off++];
      v |= (uint64_t)(b & 0x7f) << shift; shift += 7;
      if (!(b & 0x80)) { if (sg && (b & 0x40) && shift < 64) v |= ~0ull << shift; out = (int64_t)v; return true; }
    }
    return false;
  }
  bool next(bool& is_null, int64_t& v) {
    if (done()) { is_null = true; v = 0; return true; }
    if (count == 0) {
      int64_t n;
      if (!leb(true, n)) return false;
      if (n > 1) { if (!leb(is_signed, last)) return false; last_null = false; state = 1; count = n; }
      else if (n == 1) return false;
      else if (n < 0) { state = 2; count = -n; }
      else { int64_t z; if (!leb(false, z) || z <= 0) return false; state = 3; count = z; last_null = true; }
    }
    count--;
    if (state == 2) { if (!leb(is_signed, last)) return false; last_null = false; }
    is_null = last_null; v = last;
    return true;
  }
};
}  // namespace

static int load_document_impl(am355_ctx* c, const uint8_t* doc, size_t len) {
  if (!c || !doc) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  (void)hipSetDevice(c->device);
  if (c->staging_in_flight) { c->staging_in_flight = false; HIPCHK(c, hipStreamSynchronize(c->stream)); }
  c->staged = c->replayed = c->ir_fetched = false;
  c->history_ok = false;
  c->is_document = true;
  c->flags = 0;
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "load_document: %-30s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  auto bad0 = [&](uint32_t flag, const char* msg) { c->flags |= flag; return fail(c, AM355_E_INVALID, "%s", msg); };
  if (len < 10 || doc[0] != 0x85 || doc[1] != 0x6f || doc[2] != 0x4a || doc[3] != 0x83) return bad0(AM355_F_BAD_MAGIC, "Data does not begin with magic bytes 85 6f 4a 83");
  size_t off = 9;
  uint64_t clen;
  if (!read_uleb_host(doc, len, off, clen) || clen != len - off) return bad0(AM355_F_BAD_CHUNK, "Encoded document has trailing data or is truncated");
  if (doc[8] != 0) return bad0(AM355_F_BAD_CHUNK, "Unexpected chunk type");
  // The chunk checksum (one SHA-256 over the whole chunk: sequential by construction) runs on a pool thread beside the column
  // inflates below. The reference verifies it before it reads the header (columnar.js:699-705), so a malformed header is only
  // reported once the checksum is known to match.
  bool sum_done = false, sum_ok = false;
  auto check_sum = [&]() {
    uint8_t digest[32];
    sha256_digest(doc + 8, len - 8, digest);
    sum_ok = memcmp(digest, doc + 4, 4) == 0;
    sum_done = true;
  };
  auto bad = [&](uint32_t flag, const char* msg) {
    if (!sum_done) check_sum();
    if (!sum_ok) { flag = AM355_F_BAD_CHECKSUM; msg = "checksum does not match data"; }
    c->flags |= flag;
    return fail(c, AM355_E_INVALID, "%s", msg);
  };
  const uint8_t* h = doc + off;
  size_t hl = (size_t)clen, ho = 0;
  uint64_t na, nh;
  if (!read_uleb_host(h, hl, ho, na) || na > hl) return bad(AM355_F_BAD_LEB, "bad document header");
  c->actors.clear();
  for (uint64_t i = 0; i < na; i++) {
    uint64_t l;
    if (!read_uleb_host(h, hl, ho, l) || l > hl - ho) return bad(AM355_F_BAD_LEB, "bad document header");
    c->actors.emplace_back((const char*)h + ho, (size_t)l);
    ho += (size_t)l;
  }
  if (!read_uleb_host(h, hl, ho, nh) || nh > (hl - ho) / 32) return bad(AM355_F_BAD_LEB, "bad document header");
  c->heads.assign(h + ho, h + ho + nh * 32);
  ho += (size_t)nh * 32;
  struct Col { uint64_t id, len; std::vector<uint8_t>* data = nullptr; const uint8_t* p = nullptr; size_t n = 0; };  // data: inflated bytes (a scratch vector of the context)
  auto read_dir = [&](std::vector<Col>& cols) -> bool {
    uint64_t n;
    if (!read_uleb_host(h, hl, ho, n) || n > hl) return false;
    int64_t last = -1;
    for (uint64_t i = 0; i < n; i++) {
      Col col;
      if (!read_uleb_host(h, hl, ho, col.id) || !read_uleb_host(h, hl, ho, col.len)) return false;
      if ((int64_t)(col.id & ~8ull) <= last) return false;  // Columns must be in ascending order (deflate bit ignored)
      last = (int64_t)(col.id & ~8ull);
      cols.push_back(std::move(col));
    }
    return true;
  };
  std::vector<Col> ccols, ocols;
  if (!read_dir(ccols) || !read_dir(ocols)) return bad(AM355_F_BAD_COLUMNS, "bad column directory");
  // column slices, then: checksum | copy of the document bytes (Backend.save of an unchanged document returns them, new.js:2034) |
  // raw-DEFLATE of every compressed column (columnar.js:1062-1067), all on the host pool, longest columns first
  std::vector<Col*> all_cols;
  for (Col& col : ccols) all_cols.push_back(&col);
  for (Col& col : ocols) all_cols.push_back(&col);
  for (Col* col : all_cols) {
    if (col->len > hl - ho) return bad(AM355_F_BAD_CHUNK, "document columns exceed the chunk");
    col->p = h + ho;
    col->n = (size_t)col->len;
    ho += (size_t)col->len;
  }
  {
    std::vector<Col*> deflated;
    for (Col* col : all_cols)
      if (col->id & 8) deflated.push_back(col);
    std::sort(deflated.begin(), deflated.end(), [](const Col* x, const Col* y) { return x->len > y->len; });
    std::vector<int> irc(deflated.size(), 0);
    // (inflate buffers live in the context: the k-th longest column of the next document finds its pages already mapped -- a fresh
    // 34 MB vector costs ~10 ms of page faults on the thread that is the critical path of this call)
    if (c->inflate_scratch.size() < deflated.size()) c->inflate_scratch.resize(deflated.size());
    for (size_t k = 0; k < deflated.size(); k++) deflated[k]->data = &c->inflate_scratch[k];
    const unsigned n_tasks = (unsigned)deflated.size() + 2;
    c->pool->run(n_tasks, [&](unsigned t) {
      // (the two longest columns first, then the checksum, which takes about as long as a mid-sized column)
      unsigned sum_slot = std::min<unsigned>(2, (unsigned)deflated.size()), copy_slot = sum_slot + 1;
      if (t == sum_slot) { check_sum(); return; }
      if (t == copy_slot) { c->doc_bytes.assign(doc, doc + len); return; }
      size_t k = t < sum_slot ? t : t - 2;
      Col* col = deflated[k];
      irc[k] = inflate_raw(col->p, (size_t)col->len, *col->data, INFLATE_CAP);
    });
    lap("inflate | checksum | copy");
    if (!sum_ok) return bad(AM355_F_BAD_CHECKSUM, "checksum does not match data");
    int rd = 0;
    for (Col* col : all_cols) {  // (errors in column order, as a sequential reader would meet them)
      if (!(col->id & 8)) continue;
      size_t k = (size_t)(std::find(deflated.begin(), deflated.end(), col) - deflated.begin());
      if (irc[k]) { rd = irc[k] == 1 ? 2 : irc[k] == 2 ? 3 : 4; break; }
      col->p = col->data->data();
      col->n = col->data->size();
      col->id ^= 8;
    }
    if (rd == 2) return bad(AM355_F_BAD_DEFLATE, "invalid or truncated deflate data in a document column");
    if (rd == 3) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "document column inflates beyond the 4 GiB limit"); }
    if (rd == 4) return fail(c, AM355_E_NOMEM, "inflate: out of memory");
  }
  // (headsIndexes and extraBytes follow; neither influences the patch: kept for am355_save. The reference reads one index per
  // head when anything follows the columns, columnar.js:1032-1034)
  if (ho < hl) {
    size_t to = ho;
    for (uint64_t i = 0; i < nh; i++) {
      uint64_t ix;
      if (!read_uleb_host(h, hl, to, ix) || ix >= (1ull << 53)) return bad(AM355_F_BAD_LEB, "bad head index after the columns");
    }
  }
  c->doc_tail.assign(h + ho, h + hl);
  c->doc_chg_cols.clear();
  for (Col& col : ccols) c->doc_chg_cols.emplace_back((uint32_t)col.id, std::vector<uint8_t>(col.p, col.p + col.n));
  c->doc_other_ops_cols = false;
  for (Col& col : ocols) {
    static const uint64_t known[] = {0x01, 0x02, 0x11, 0x13, 0x15, 0x21, 0x23, 0x34, 0x42, 0x56, 0x57, 0x80, 0x81, 0x83};
    bool k = false;
    for (uint64_t id : known) k = k || id == col.id;
    if (!k && col.n) c->doc_other_ops_cols = true;
  }

  // ---- change metadata: clock in first-appearance order, seq continuity (new.js:1645-1675) ----
  auto find = [](std::vector<Col>& cols, uint64_t id) -> Col* { for (Col& x : cols) if (x.id == id) return &x; return nullptr; };
  {
    Col* ca = find(ccols, 0x01);
    Col* cs = find(ccols, 0x03);
    HostRle ra(ca ? ca->p : nullptr, ca ? ca->n : 0, false), rs(cs ? cs->p : nullptr, cs ? cs->n : 0, true);
    std::vector<uint64_t> clock(na, 0);
    std::vector<uint8_t> seen(na, 0);
    c->clock_actor.clear();
    int64_t seq_abs = 0;
    uint32_t n_changes = 0;
    while (!ra.done()) {
      bool an, sn;
      int64_t a, dv;
      if (!ra.next(an, a) || !rs.next(sn, dv)) return bad(AM355_F_BAD_RLE, "malformed change metadata columns");
      if (an || a < 0 || (uint64_t)a >= na) return bad(AM355_F_BAD_ROW, "bad actor index in change metadata");
      if (!sn) seq_abs += dv;
      uint64_t seq = sn ? 0 : (uint64_t)seq_abs;
      if (seq != 1 && seq != clock[a] + 1) { c->flags |= AM355_F_BAD_SEQ; return fail(c, AM355_E_INVALID, "Expected seq %llu, got %llu", (unsigned long long)clock[a] + 1, (unsigned long long)seq); }
      if (!seen[a]) { seen[a] = 1; c->clock_actor.push_back((uint32_t)a); }  // document actor index for now, ranks below
      clock[a] = seq;
      n_changes++;
      if (n_changes > (1u << 26)) return fail(c, AM355_E_UNSUPPORTED, "more than 2^26 changes in one document");  // (a run length can claim any count)
    }
    c->n_changes = n_changes;
    c->clock_seq.clear();
    for (uint32_t a : c->clock_actor) c->clock_seq.push_back(clock[a]);
  }
  lap("change metadata");
  // ---- actor ranks: op-id comparison on the device is numeric on (ctr, rank) ----
  {
    std::vector<uint32_t> order(na);
    for (uint32_t i = 0; i < na; i++) order[i] = i;
    std::vector<std::string> names = c->actors;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return names[x] < names[y]; });
    c->doc_actor_rank.assign(na, 0);
    for (uint32_t r = 0; r < na; r++) { c->doc_actor_rank[order[r]] = r; c->actors[r] = names[order[r]]; }
    for (uint32_t& a : c->clock_actor) a = c->doc_actor_rank[a];
  }
  // ---- op columns -> one arena; layout recorded like a change's column directory ----
  c->raw.clear();
  c->raw_off.assign(1, 0);
  ChangeMeta& m = c->doc_meta;
  memset(&m, 0, sizeof m);
  m.n_entries = (uint32_t)na;
  // (placement first -- offsets only --, the bytes follow in parallel pieces together with their H2D copies)
  struct Piece { const uint8_t* src; size_t dst, n; };
  std::vector<Piece> pieces;
  size_t arena_bytes = 0;
  uint8_t last_byte = 0;
  auto place = [&](int slot, uint64_t id) {
    Col* col = find(ocols, id);
    if (col && arena_bytes + col->n >= 0xfff00000ull) { arena_bytes = 0xfff00000ull; return; }
    m.col_off[slot] = (uint32_t)arena_bytes;
    m.col_len[slot] = col ? (uint32_t)col->n : 0;
    if (col && col->n) {
      for (size_t o = 0; o < col->n; o += (size_t)4 << 20) pieces.push_back(Piece{col->p + o, arena_bytes + o, std::min<size_t>((size_t)4 << 20, col->n - o)});
      last_byte = col->p[col->n - 1];
      arena_bytes += col->n;
    }
  };
  // the LEB-tokenisable columns first (BigCol order), the two byte-string columns after them
  static const struct { int slot; uint64_t id; uint32_t kind; } big[BIG_NCOL] = {
      {C_OBJ_ACTOR, 0x01, BK_UINT}, {C_OBJ_CTR, 0x02, BK_UINT}, {C_KEY_ACTOR, 0x11, BK_UINT}, {C_KEY_CTR, 0x13, BK_DELTA}, {C_ID_ACTOR, 0x21, BK_UINT},
      {C_ID_CTR, 0x23, BK_DELTA}, {C_INSERT, 0x34, BK_BOOL}, {C_ACTION, 0x42, BK_UINT}, {C_VAL_LEN, 0x56, BK_UINT}, {C_PRED_NUM, 0x80, BK_UINT},
      {C_PRED_ACTOR, 0x81, BK_UINT}, {C_PRED_CTR, 0x83, BK_DELTA}};
  for (int k = 0; k < BIG_NCOL; k++) {
    place(big[k].slot, big[k].id);
    c->doc_cols.off[k] = m.col_off[big[k].slot];
    c->doc_cols.len[k] = m.col_len[big[k].slot];
    c->doc_cols.kind[k] = big[k].kind;
    // every column must end on the last byte of a number (the device finds numbers by their terminating bytes)
    if (m.col_len[big[k].slot] && (last_byte & 0x80)) return bad(AM355_F_BAD_LEB, "incomplete number");
  }
  c->doc_cols.tok_bytes = (uint32_t)arena_bytes;
  place(C_KEY_STR, 0x15); place(C_VAL_RAW, 0x57);
  {
    const char* e = getenv("AM355_DOC_SERIAL");
    c->doc_serial = e && *e == '1';
  }
  if (arena_bytes >= 0xfff00000ull) return fail(c, AM355_E_UNSUPPORTED, "document larger than 4 GiB (32-bit arena offsets)");
  lap("columns placed");
  c->raw.resize(arena_bytes);
  c->raw_off.push_back(arena_bytes);
  m.len = (uint32_t)arena_bytes;
  if (!c->d_arena.ensure(arena_bytes + 64) || !c->d_metas.ensure(sizeof(ChangeMeta)) || !c->h_metas.ensure(sizeof(ChangeMeta)) ||
      !c->d_counts.ensure(sizeof(Counts)) || !c->h_counts.ensure(2 * sizeof(Counts)))
    return fail(c, AM355_E_NOMEM, "device allocation failed");
  {
    std::vector<hipError_t> h2d(pieces.size(), hipSuccess);
    uint8_t* raw = c->raw.data();
    c->pool->run((unsigned)pieces.size(), [&](unsigned k) {
      (void)hipSetDevice(c->device);
      const Piece& pc = pieces[k];
      memcpy(raw + pc.dst, pc.src, pc.n);
      h2d[k] = hipMemcpyAsync(c->d_arena.as<uint8_t>() + pc.dst, raw + pc.dst, pc.n, hipMemcpyHostToDevice, c->stream);
    });
    for (hipError_t e : h2d)
      if (e != hipSuccess) return fail(c, AM355_E_DEVICE, "hipMemcpyAsync (document columns): %s", hipGetErrorString(e));
  }
  lap("gathered, H2D enqueued");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  lap("H2D done");
  c->staged = true;
  c->stats = am355_stats{};
  c->stats.n_changes = c->n_changes;
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

// Open-addressing set of 32-byte hashes (keyed by their first 8 bytes, verified by full comparison).
struct HashSet {
  std::vector<const uint8_t*> slot;
  size_t mask = 0;
  void init(size_t n) {
    size_t cap = 16;
    while (cap < n * 2 + 2) cap <<= 1;
    slot.assign(cap, nullptr);
    mask = cap - 1;
  }
  static uint64_t key(const uint8_t* h) { uint64_t v; memcpy(&v, h, 8); return v * 0x9e3779b97f4a7c15ull; }
  const uint8_t** find(const uint8_t* h) {
    size_t i = (size_t)(key(h) >> 20) & mask;
    while (slot[i]) {
      if (slot[i] != (const uint8_t*)1 && memcmp(slot[i], h, 32) == 0) return &slot[i];
      i = (i + 1) & mask;
    }
    return nullptr;
  }
  bool has(const uint8_t* h) { return find(h) != nullptr; }
  void add(const uint8_t* h) {
    size_t i = (size_t)(key(h) >> 20) & mask;
    while (slot[i] && slot[i] != (const uint8_t*)1) i = (i + 1) & mask;
    slot[i] = h;
  }
  void del(const uint8_t* h) {
    const uint8_t** p = find(h);
    if (p) *p = (const uint8_t*)1;  // tombstone
  }
};

// General scheduler: exact restatement of the reference's retry loop for any delivery order, duplicates and
// missing dependencies. Used when the device-side checks cannot prove the in-order fast path.
static uint32_t rank_device_actors(am355_ctx* c, std::vector<uint32_t>& slot_rank);
// dev_amap / dev_amap_base: host copies of the device's actor tables (slot numbers) or null
static int schedule(am355_ctx* c, const uint32_t* dev_amap, const uint32_t* dev_amap_base) {
  auto T0 = std::chrono::steady_clock::now();
  const ChangeMeta* metas = c->h_metas.as<ChangeMeta>();
  const uint8_t* hashes = c->h_hashes.as<uint8_t>();
  uint32_t n = c->n_changes;
  const uint8_t* raw = c->raw.data();
  uint32_t dev_flags = 0;
  for (uint32_t i = 0; i < n; i++) dev_flags |= metas[i].flags;
  if (dev_flags) {
    c->flags |= dev_flags;
    return fail(c, (dev_flags & (F_OVERFLOW | F_UNSUPPORTED)) ? AM355_E_UNSUPPORTED : AM355_E_INVALID, "malformed change (flags 0x%x)", dev_flags);
  }
  // ---- actor ids: global table ranked lexicographically (hex-string order == byte order, new.js:65) ----
  // dev_amap != null: the device has interned every actor-table entry (k_actor_intern): per change its entries are slot numbers at
  // dev_amap[dev_amap_base[i] ..], the distinct ids are ranked from the device's list. Otherwise (more distinct actors than that list
  // holds) the host interns: changes of one author nearly always carry the same "other actors" table, so each author's last table is
  // memoised (bytes compared), which turns O(changes x actors) string interning into O(changes) memcmp.
  std::vector<uint32_t> local_off_v, local_ids_v, rank;
  const uint32_t *local_off, *local_ids;
  uint32_t na;
  if (dev_amap) {
    na = rank_device_actors(c, rank);
    local_off = dev_amap_base;
    local_ids = dev_amap;
  } else {
  std::unordered_map<std::string, uint32_t> actor_ix;
  std::vector<std::string> names;
  local_off_v.assign(n + 1, 0);
  local_ids_v.reserve((size_t)n * 2);
  struct Memo { const uint8_t* p = nullptr; uint32_t len = 0, n_other = 0, first = 0; };
  std::vector<Memo> memo;
  auto intern = [&](const uint8_t* b, size_t len) {
    std::string s((const char*)b, len);
    auto it = actor_ix.find(s);
    if (it != actor_ix.end()) return it->second;
    uint32_t id = (uint32_t)names.size();
    actor_ix.emplace(s, id);
    names.push_back(std::move(s));
    memo.emplace_back();
    return id;
  };
  for (uint32_t i = 0; i < n; i++) {
    const ChangeMeta& m = metas[i];
    const uint8_t* p = raw + m.base;
    uint32_t author = intern(p + m.actor_off, m.actor_len);
    local_ids_v.push_back(author);
    // bytes of the other-actors table: from others_off up to the column directory; its exact end is found by parsing
    Memo& mm = memo[author];
    size_t off = m.others_off;
    if (mm.p && mm.n_other == m.n_other && m.others_off + mm.len <= m.len && memcmp(mm.p, p + m.others_off, mm.len) == 0) {
      for (uint32_t k = 0; k < m.n_other; k++) local_ids_v.push_back(local_ids_v[mm.first + k]);
    } else {
      uint32_t first = (uint32_t)local_ids_v.size();
      for (uint32_t k = 0; k < m.n_other; k++) {
        uint64_t l;
        read_uleb_host(p, m.len, off, l);
        uint32_t id = intern(p + off, (size_t)l);
        local_ids_v.push_back(id);
        off += (size_t)l;
      }
      Memo& m2 = memo[author];  // (memo may have grown)
      m2.p = p + m.others_off;
      m2.len = (uint32_t)(off - m.others_off);
      m2.n_other = m.n_other;
      m2.first = first;
    }
    local_off_v[i + 1] = (uint32_t)local_ids_v.size();
  }
  na = (uint32_t)names.size();
  std::vector<uint32_t> by_rank(na);
  rank.assign(na, 0);
  for (uint32_t i = 0; i < na; i++) by_rank[i] = i;
  std::sort(by_rank.begin(), by_rank.end(), [&](uint32_t x, uint32_t y) { return names[x] < names[y]; });  // std::string compares bytes as unsigned char
  for (uint32_t r = 0; r < na; r++) rank[by_rank[r]] = r;
  c->actors.resize(na);
  for (uint32_t r = 0; r < na; r++) c->actors[r] = names[by_rank[r]];
  local_off = local_off_v.data();
  local_ids = local_ids_v.data();
  }
  auto T1 = std::chrono::steady_clock::now();

  auto T2 = std::chrono::steady_clock::now();
  // ---- causal scheduling (new.js:1550-1597 inside the retry loop of :1822-1841) ----
  // The device has resolved every hash to an index (k_deps_resolve): self[ci] = first change of the batch with ci's hash (ci itself
  // unless it is a duplicate), dep(ci, k) = first change with that dependency's hash or NONE32. "hash known" is then applied[index],
  // and the retry loop of the reference runs on integers.
  const uint32_t* self = c->h_self_idx.as<uint32_t>();
  const uint32_t* dep_idx = c->h_dep_idx.as<uint32_t>();
  std::vector<uint8_t> is_head(n, 0);
  std::vector<uint64_t> clock(na, 0);
  std::vector<uint8_t> has_clock(na, 0), actor_read(na, 0);
  c->clock_actor.clear();
  // The retry loop applies, pass after pass, every queued change whose dependencies were applied earlier -- in an earlier pass or
  // earlier in the same pass (the queue keeps its order). So the pass a change is applied in is
  //     pass(c) = max over its dependencies d of  pass(d) + (d sits after c in the queue ? 1 : 0)        (0 without dependencies),
  // infinite if a dependency is not in the batch or is itself never applied; the application order is (pass, position). Later copies
  // of a change are dropped once the first copy is applied. One memoised walk over the dependency edges instead of one scan of the
  // queue per pass (64 synced rounds delivered in random order need dozens of passes).
  std::vector<uint32_t> applied_all, applied_pass;
  uint32_t sched_flags = 0, n_pending = 0;
  {
    constexpr uint32_t UNSET = 0xffffffffu, NEVER = 0xfffffffeu, BUSY = 0xfffffffdu;
    // Copies of one change (the same hash several times in the queue) form a group named by its first copy (self[]): every copy is
    // ready as soon as ITS position allows -- a copy standing behind the dependencies its first copy stands in front of is ready a
    // pass earlier -- and the reference applies whichever copy becomes ready first, (pass, position) minimal, dropping the others as
    // duplicates from then on (new.js:1566). gpass[F] / gpos[F]: pass and position at which group F is applied. (Round 3 applied
    // the FIRST copy only: wrong application order -- visible in the order of the `clock` keys -- whenever a later copy was ready
    // sooner; found by the device scheduler's tests against the oracle, which has it right.)
    std::vector<uint32_t> gpass(n, UNSET), gpos(n, 0), pass(n, NEVER), stack;
    std::vector<uint32_t> copy_next(n, UNSET), copy_tail(n, UNSET);   // the copies of a group, ascending
    for (uint32_t ci = 0; ci < n; ci++) {
      uint32_t F = self[ci] < n ? self[ci] : ci;
      if (F != ci) { uint32_t tail = copy_tail[F] == UNSET ? F : copy_tail[F]; copy_next[tail] = ci; copy_tail[F] = ci; }
    }
    // (dependency list of a change as a compact (first, count) pair: the walk below visits every edge twice and the change records
    // are 176 bytes apart)
    std::vector<uint32_t> dep_first(n), dep_count(n);
    for (uint32_t ci = 0; ci < n; ci++) { const ChangeMeta& m = metas[ci]; dep_first[ci] = (uint32_t)((m.base + m.deps_off) >> 5); dep_count[ci] = m.n_deps; }
    auto dep_of = [&](uint32_t ci, uint32_t k) { return dep_idx[dep_first[ci] + k]; };
    for (uint32_t root = 0; root < n; root++) {
      if ((self[root] < n ? self[root] : root) != root || gpass[root] != UNSET) continue;
      stack.push_back(root);
      while (!stack.empty()) {
        const uint32_t F = stack.back();
        if (gpass[F] != UNSET && gpass[F] != BUSY) { stack.pop_back(); continue; }
        // every copy of the group from the groups of its dependencies
        bool pushed = false;
        uint32_t best_p = NEVER, best_pos = 0;
        for (uint32_t ci = F; ci != UNSET && !pushed; ci = copy_next[ci]) {
          uint32_t p = 0;
          for (uint32_t k = 0, nd = dep_count[ci]; k < nd; k++) {
            const uint32_t d = dep_of(ci, k);
            if (d >= n) { p = NEVER; break; }
            if (gpass[d] == UNSET) { gpass[F] = BUSY; stack.push_back(d); pushed = true; break; }
            if (gpass[d] == BUSY || gpass[d] == NEVER) { p = NEVER; break; }  // (a dependency cycle would need a hash collision: never applied)
            const uint32_t q = gpass[d] + (gpos[d] > ci ? 1u : 0u);
            p = q > p ? q : p;
          }
          if (pushed) break;
          if (p != NEVER && (best_p == NEVER || p < best_p)) { best_p = p; best_pos = ci; }  // (copies ascend: the first of the earliest pass)
        }
        if (pushed) continue;  // come back when the dependencies are known
        gpass[F] = best_p;
        gpos[F] = best_pos;
        if (best_p != NEVER) pass[best_pos] = best_p;
        stack.pop_back();
      }
    }
    // application order: by (pass, position) -- a counting sort over the passes
    uint32_t max_pass = 0;
    for (uint32_t ci = 0; ci < n; ci++)
      if (pass[ci] < BUSY && pass[ci] > max_pass) max_pass = pass[ci];
    std::vector<uint32_t> start(max_pass + 2, 0);
    for (uint32_t ci = 0; ci < n; ci++)
      if (pass[ci] < BUSY) start[pass[ci] + 1]++;
    for (uint32_t p = 0; p <= max_pass; p++) start[p + 1] += start[p];
    applied_all.resize(start[max_pass + 1]);
    for (uint32_t ci = 0; ci < n; ci++)
      if (pass[ci] < BUSY) applied_all[start[pass[ci]]++] = ci;
    // what stays queued: the changes of which no copy is ever applied
    c->pending_change.clear();
    for (uint32_t ci = 0; ci < n; ci++) {
      uint32_t first = self[ci] < n ? self[ci] : ci;
      if (gpass[first] >= BUSY) { n_pending++; c->pending_change.push_back(ci); }
    }
    applied_pass.resize(applied_all.size());
    for (size_t t = 0; t < applied_all.size(); t++) applied_pass[t] = pass[applied_all[t]];
    // sequence numbers, clock, heads and the actor rule in application order (new.js:1571-1578, 1582-1583, 1442-1449)
    for (uint32_t ci : applied_all) {
      const ChangeMeta& m = metas[ci];
      uint32_t author = rank[local_ids[local_off[ci]]];
      if (m.seq != clock[author] + 1) { sched_flags |= AM355_F_BAD_SEQ; break; }
      if (!has_clock[author]) { has_clock[author] = 1; c->clock_actor.push_back(author); }
      clock[author] = m.seq;
      for (uint32_t k = 0, nd = dep_count[ci]; k < nd; k++) is_head[gpos[dep_of(ci, k)]] = 0;  // (the copy of the dependency that was applied)
      is_head[ci] = 1;
    }
    // each change may only mention actors already in the document when it is read: the reference reads the changes of a pass
    // after the whole pass has been scheduled
    if (!sched_flags) {
      size_t i = 0;
      while (i < applied_all.size()) {
        size_t j = i;
        uint32_t p = pass[applied_all[i]];
        while (j < applied_all.size() && pass[applied_all[j]] == p) j++;
        for (size_t t = i; t < j; t++) {
          uint32_t ci = applied_all[t];
          actor_read[rank[local_ids[local_off[ci]]]] = 1;
          for (uint32_t k = local_off[ci]; k < local_off[ci + 1]; k++)
            if (!actor_read[rank[local_ids[k]]]) sched_flags |= AM355_F_UNKNOWN_ACTOR;
        }
        i = j;
      }
    }
  }
  if (sched_flags) {
    c->flags |= sched_flags;
    return fail(c, AM355_E_INVALID, "change schedule rejected (flags 0x%x)", sched_flags);
  }
  c->n_applied = (uint32_t)applied_all.size();
  c->n_pending = n_pending;
  c->clock_seq.clear();
  for (uint32_t a : c->clock_actor) c->clock_seq.push_back(clock[a]);
  {
    std::vector<const uint8_t*> hs;
    for (uint32_t ci : applied_all)
      if (is_head[ci]) hs.push_back(hashes + 32 * (size_t)ci);
    std::sort(hs.begin(), hs.end(), [](const uint8_t* x, const uint8_t* y) { return memcmp(x, y, 32) < 0; });
    c->heads.resize(hs.size() * 32);
    for (size_t i = 0; i < hs.size(); i++) memcpy(&c->heads[32 * i], hs[i], 32);
  }

  auto T3 = std::chrono::steady_clock::now();
  // ---- launch plan for the decode kernels, op-id -> row tables ----
  c->plans.clear();
  c->amap.clear();
  c->amap.reserve(local_off[n]);
  uint64_t ops = 0, preds = 0, max_op = 0;
  std::vector<std::vector<ActorSpan>> per_actor(na);
  c->applied_change.clear();
  c->applied_op_base.clear();
  c->pass_first_row.clear();
  for (size_t t = 0; t < applied_all.size(); t++) {
    uint32_t ci = applied_all[t];
    const ChangeMeta& m = metas[ci];
    if (t > 0 && applied_pass[t] != applied_pass[t - 1]) c->pass_first_row.push_back((uint32_t)ops);  // (am355_apply_changes: a merge call never spans two passes)
    c->applied_change.push_back(ci);  // (changes without ops are applied too: they have no plan, but a place in the history)
    c->applied_op_base.push_back((uint32_t)ops);
    ChangePlan pl;
    pl.change = ci;
    pl.op_base = (uint32_t)ops;
    pl.pred_base = (uint32_t)preds;
    pl.amap_base = (uint32_t)c->amap.size();
    pl.author = rank[local_ids[local_off[ci]]];
    pl.n_actors = local_off[ci + 1] - local_off[ci];
    for (uint32_t k = local_off[ci]; k < local_off[ci + 1]; k++) c->amap.push_back(rank[local_ids[k]]);
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
  if (getenv("AM355_DEBUG_TIMING")) {
    auto T4 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "schedule: actors %.3f rank %.3f causal %.3f plan %.3f ms\n", ms(T0, T1), ms(T1, T2), ms(T2, T3), ms(T3, T4));
  }
  return AM355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// replay
// ---------------------------------------------------------------------------------------------------------
template <class T>
static T* carve(uint8_t*& p, size_t count) {
  T* r = (T*)p;
  canary_note(p, count * sizeof(T));
  p += carve_round(count * sizeof(T));
  return r;
}

static size_t carve_size(size_t count, size_t elem) { return carve_round(count * elem); }

static uint32_t pow2_at_least(uint64_t v) {
  uint32_t p = 64;
  while (p < v) p <<= 1;
  return p;
}

// words shared with the device: [0] flags of the critical-path kernels, [1] fast-path word (stream A part),
// [2] total actor-table entries, [3] flags of the hash stream, [4] fast-path word (stream B part)
enum { W_FLAGS_A = 0, W_FAST_A = 1, W_TOTAL_ENTRIES = 2, W_FLAGS_B = 3, W_FAST_B = 4, W_NUM = 8 };

static int error_for_flags(am355_ctx* c, uint32_t f, const char* what) {
  c->flags |= f;
  uint32_t hard = f & ~(uint32_t)(F_OVERFLOW | F_UNSUPPORTED);
  return fail(c, hard ? AM355_E_INVALID : AM355_E_UNSUPPORTED, "%s (flags 0x%x)", what, f);
}

// distinct actor ids as interned by the device (k_actor_intern) -> lexicographic ranks (hex-string order == byte order, new.js:65):
// fills slot_rank[slot] and c->actors (by rank). Returns the number of actors.
static uint32_t rank_device_actors(am355_ctx* c, std::vector<uint32_t>& slot_rank) {
  const uint32_t* distinct = c->hp_distinct;
  const unsigned long long* slots = (const unsigned long long*)(distinct + 2 + distinct_capacity());  // ((offset + 1) << 16) | length
  const uint8_t* raw = c->raw.data();
  uint32_t n_slots = c->slot_mask + 1;
  struct Ent { uint32_t slot, off, len; };
  static thread_local std::vector<Ent> ents;
  ents.clear();
  uint32_t nd = distinct[0];
  for (uint32_t k = 0; k < nd; k++) {
    uint32_t i = distinct[1 + k];
    ents.push_back(Ent{i, (uint32_t)((slots[k] >> 16) - 1), (uint32_t)(slots[k] & 0xffff)});
  }
  std::sort(ents.begin(), ents.end(), [&](const Ent& x, const Ent& y) {
    uint32_t m = std::min(x.len, y.len);
    int r = m ? memcmp(raw + x.off, raw + y.off, m) : 0;
    return r ? r < 0 : x.len < y.len;
  });
  uint32_t na = (uint32_t)ents.size();
  slot_rank.assign(n_slots, 0);
  c->actors.resize(na);
  for (uint32_t r = 0; r < na; r++) {
    slot_rank[ents[r].slot] = r;
    c->actors[r].assign((const char*)raw + ents[r].off, ents[r].len);
  }
  return na;
}

// Host half of the in-order fast path: O(changes + actors log actors), no allocation in steady state. Everything that
// needs the change hashes (dependency resolution, heads) has been checked on the device and is confirmed when stream
// B is joined.
// `order` (general path, device scheduler am355_sched.hip): the applied changes in application order (n_applied of them) with the
// scheduling pass of every change in `pass`; null: every change is applied, in input order (in-order fast path).
static int plan_fast(am355_ctx* c, std::vector<uint32_t>& slot_rank, const uint32_t* order = nullptr, uint32_t n_applied = 0, const uint32_t* pass = nullptr) {
  const ChangeBrief* br = c->hp_briefs;
  uint32_t n = order ? n_applied : c->n_changes;
  uint32_t na = rank_device_actors(c, slot_rank);
  static thread_local std::vector<uint64_t> clock;
  static thread_local std::vector<uint32_t> span_cnt;
  clock.assign(na, 0);
  span_cnt.assign(na + 1, 0);
  c->clock_actor.clear();
  c->plans.clear();
  c->plans.reserve(n);
  c->applied_change.resize(n);
  c->applied_op_base.resize(n);
  c->pass_first_row.clear();
  uint64_t ops = 0, preds = 0, entries = 0, max_op = 0;
  for (uint32_t t = 0; t < n; t++) {
    const uint32_t ci = order ? order[t] : t;
    const ChangeBrief& m = br[ci];
    if (order && t > 0 && pass[ci] != pass[order[t - 1]]) c->pass_first_row.push_back((uint32_t)ops);  // (am355_apply_changes: a merge call never spans two passes)
    c->applied_change[t] = ci;
    c->applied_op_base[t] = (uint32_t)ops;
    uint32_t author = slot_rank[m.author_slot];
    if (m.seq != clock[author] + 1) { c->flags |= AM355_F_BAD_SEQ; return fail(c, AM355_E_INVALID, "sequence number %llu out of order", (unsigned long long)m.seq); }
    if (clock[author] == 0) c->clock_actor.push_back(author);
    clock[author] = m.seq;
    if (m.n_ops) {
      c->plans.push_back(ChangePlan{ci, (uint32_t)ops, (uint32_t)preds, (uint32_t)entries, author, m.n_entries});
      span_cnt[author]++;
      max_op = std::max<uint64_t>(max_op, (uint64_t)m.start_op + m.n_ops - 1);
    }
    ops += m.n_ops;
    preds += m.n_preds;
    entries += m.n_entries;
    if (ops >= 0x7ffffff0ull || preds >= 0xfffffff0ull) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "more than 2^31 ops in one batch"); }
  }
  c->n_applied = n;
  c->n_pending = 0;
  c->pending_change.clear();
  c->n_ops = ops;
  c->n_preds = preds;
  c->max_op = max_op;
  c->clock_seq.clear();
  for (uint32_t a : c->clock_actor) c->clock_seq.push_back(clock[a]);
  // per-actor tables of (start_op, n_ops, op_base): counting layout, then verify ascending and disjoint
  c->actor_tab_off.assign(na + 1, 0);
  for (uint32_t a = 0; a < na; a++) c->actor_tab_off[a + 1] = c->actor_tab_off[a] + span_cnt[a];
  c->spans.resize(c->actor_tab_off[na]);
  for (uint32_t a = 0; a < na; a++) span_cnt[a] = c->actor_tab_off[a];
  for (const ChangePlan& pl : c->plans) c->spans[span_cnt[pl.author]++] = ActorSpan{br[pl.change].start_op, br[pl.change].n_ops, pl.op_base};
  for (uint32_t a = 0; a < na; a++) {
    ActorSpan* v = c->spans.data() + c->actor_tab_off[a];
    size_t k_n = c->actor_tab_off[a + 1] - c->actor_tab_off[a];
    bool sorted = true;
    for (size_t k = 1; k < k_n; k++) sorted = sorted && v[k - 1].start_op <= v[k].start_op;
    if (!sorted) std::sort(v, v + k_n, [](const ActorSpan& x, const ActorSpan& y) { return x.start_op < y.start_op; });
    for (size_t k = 1; k < k_n; k++)
      if ((uint64_t)v[k - 1].start_op + v[k - 1].n_ops > v[k].start_op) {
        c->flags |= AM355_F_DUP_OPID;
        return fail(c, AM355_E_INVALID, "overlapping op id ranges for one actor (duplicate operation ID)");
      }
  }
  return AM355_OK;
}

// Device buffers for N op rows / P preds, decode, merge, patch IR. `slot_rank` != null: actor tables are the
// device-interned slots (fast path); null: c->amap holds ranks (general path).
// Device buffers for N op rows / P preds (op rows, merge scratch, sort scratch, patch IR), carved from a few arenas.
static int setup_buffers(am355_ctx* c, uint32_t NA) {
  uint32_t N = (uint32_t)c->n_ops, P = (uint32_t)c->n_preds;
  int bits_ctr = bits_for64(c->max_op), bits_actor = bits_for64(NA ? NA - 1 : 0), bits_row = bits_for64(N);
  if (1 + bits_row + bits_ctr + bits_actor > 64) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "sort key wider than 64 bits"); }
  size_t Nc = (size_t)N + 1;
  canary_scope("replay buffers (setup_buffers: op rows, preds, merge scratch, sort scratch, patch IR)");
  {
    size_t bytes = 13 * carve_size(Nc, 4) + carve_size(Nc, 1);
    if (!c->d_cols.ensure(bytes) || !c->d_pred.ensure(2 * carve_size((size_t)P + 1, 4))) return fail(c, AM355_E_NOMEM, "device allocation failed (op rows)");
    canary_forget(c->d_cols.p, c->d_cols.cap); canary_forget(c->d_pred.p, c->d_pred.cap);
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
    size_t cw = carry_words(N);
    size_t bytes = 10 * carve_size(Nc, 4) + 3 * carve_size(Nc, 8) + carve_size(Nc, 1) + carve_size(2 * Nc + 2, 4) + 4 * carve_size(2 * Nc + 2, 4) +
                   3 * carve_size(Nc + 1, 4) + scan_workspace_bytes((uint32_t)(2 * Nc + 2)) + 256 +
                   6 * carve_size(Nc + 3, 4) + 6 * carve_size(cw, 4) + carve_size(2048, 4);
    size_t sort_bytes = 2 * carve_size(Nc, 8) + 2 * carve_size(Nc, 4) + sort_workspace_bytes((uint32_t)Nc) + 256;
    size_t ir_bytes = carve_size(Nc + 1, sizeof(am355_ir_object)) + carve_size(Nc, sizeof(am355_ir_map)) + carve_size(Nc + 1, sizeof(am355_ir_edit)) +
                      4 * carve_size(Nc, 4);
    if (!c->d_merge.ensure(bytes) || !c->d_sort.ensure(sort_bytes) || !c->d_ir.ensure(ir_bytes) || !c->d_counts.ensure(merge_counts_bytes(N)))
      return fail(c, AM355_E_NOMEM, "device allocation failed (merge)");
    canary_forget(c->d_merge.p, c->d_merge.cap); canary_forget(c->d_sort.p, c->d_sort.cap); canary_forget(c->d_ir.p, c->d_ir.cap);
    uint8_t* p = c->d_merge.as<uint8_t>();
    MergeBufs& b = c->mb;
    b.arena = c->d_arena.as<uint8_t>();
    b.ops = c->cols;
    b.n_ops = N; b.n_preds = P; b.n_actors = NA;
    b.shard_rank = c->shard_rank; b.shard_world = c->shard_world;
    b.sig = c->h_sig.as<HostSignals>(); b.sig_seq = c->sig_seq;
    b.actor_tab_off = c->p_tab_off;
    b.spans = c->p_spans;
    b.bits_ctr = (uint32_t)bits_ctr; b.bits_actor = (uint32_t)bits_actor;
    b.zero_base = p;
    b.succ_cnt = carve<uint32_t>(p, Nc); b.inc_cnt = carve<uint32_t>(p, Nc); b.val_cnt = carve<uint32_t>(p, Nc);
    b.inc_sum = carve<unsigned long long>(p, Nc); b.last_inc = carve<unsigned long long>(p, Nc);
    b.zero_bytes = (size_t)(p - (uint8_t*)b.zero_base) - (canary_on() ? 256 : 0);
    canary_allow(b.zero_base, b.zero_bytes);
    b.obj_row = carve<uint32_t>(p, Nc); b.ref_row = carve<uint32_t>(p, Nc); b.obj_index = carve<uint32_t>(p, Nc);
    b.em_row = carve<uint32_t>(p, Nc); b.ins_row = carve<uint32_t>(p, Nc); b.upd_row = carve<uint32_t>(p, Nc); b.next_sib = carve<uint32_t>(p, Nc);
    b.em_trig = carve<unsigned long long>(p, Nc);
    b.kind = carve<uint8_t>(p, Nc);
    // order | first_child | child_head (start of euler_b) are contiguous: one 0xff fill per replay (merge_prepare)
    b.order = carve<uint32_t>(p, Nc + 1);
    b.first_child = carve<uint32_t>(p, 2 * Nc + 2);
    b.euler_b = carve<unsigned long long>(p, 2 * Nc + 2); b.euler_a = carve<unsigned long long>(p, 2 * Nc + 2);
    b.scan_a = carve<uint32_t>(p, Nc + 1); b.scan_b = carve<uint32_t>(p, Nc + 1);
    b.scan_ws = p;
    canary_note(p, scan_workspace_bytes((uint32_t)(2 * Nc + 2)));
    p += carve_round(scan_workspace_bytes((uint32_t)(2 * Nc + 2)));
    b.run_heads = carve<uint32_t>(p, Nc + 3); b.row_run = carve<uint32_t>(p, Nc + 3); b.obj_n = carve<uint32_t>(p, Nc + 3);
    b.obj_first_pos = carve<uint32_t>(p, Nc + 3); b.list_vis = carve<uint32_t>(p, Nc + 3); b.list_cnt = carve<uint32_t>(p, Nc + 3);
    b.cs_ins.wg_sum = carve<uint32_t>(p, cw); b.cs_make.wg_sum = carve<uint32_t>(p, cw); b.cs_runs.wg_sum = carve<uint32_t>(p, cw);
    b.cs_vis.wg_sum = carve<uint32_t>(p, cw); b.cs_cnt.wg_sum = carve<uint32_t>(p, cw); b.cs_erec.wg_sum = carve<uint32_t>(p, cw);
    b.head_child = carve<uint32_t>(p, 2048);
    // unordered child lists (k_child_push) live in the second Euler buffer, which list ranking only uses afterwards
    b.child_head = (uint32_t*)b.euler_b;
    b.child_next = b.child_head + (2 * Nc + 2);
    b.fill_base = b.order;
    b.fill_bytes = (size_t)((uint8_t*)(b.child_head + 2 * Nc + 1) - (uint8_t*)b.order);
    canary_allow(b.fill_base, b.fill_bytes);
    uint8_t* s = c->d_sort.as<uint8_t>();
    b.key_a = carve<uint64_t>(s, Nc); b.key_b = carve<uint64_t>(s, Nc); b.val_a = carve<uint32_t>(s, Nc); b.val_b = carve<uint32_t>(s, Nc);
    b.sort_ws = s;
    merge_bind_counts(b, c->d_counts.p);
    uint8_t* r = c->d_ir.as<uint8_t>();
    PatchIR& ir = c->ir;
    ir.obj = carve<am355_ir_object>(r, Nc + 1); ir.map = carve<am355_ir_map>(r, Nc); ir.edit = carve<am355_ir_edit>(r, Nc + 1);
    ir.e_row = carve<uint32_t>(r, Nc); ir.e_elem = carve<uint32_t>(r, Nc); ir.e_index = carve<uint32_t>(r, Nc); ir.e_flags = carve<uint32_t>(r, Nc);
  }
  canary_arm();
  return AM355_OK;
}

// Decode + merge + patch IR for the planned changes. `slot_rank` != null: actor tables are the device-interned slots
// (fast path); null: c->amap holds ranks (general path).
static int run_device(am355_ctx* c, const std::vector<uint32_t>* slot_rank) {
  hipStream_t st = c->stream;
  size_t np = c->plans.size();
  uint32_t NA = (uint32_t)c->actors.size();
  // the host-built tables (plans, actor spans, span offsets, slot ranks or actor translation tables) live in one device block so
  // that they travel in ONE host-to-device copy from the pinned staging buffer
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t b_plans = sizeof(ChangePlan) * np, b_spans = sizeof(ActorSpan) * c->spans.size(), b_tab = 4 * c->actor_tab_off.size();
  size_t b_rank = slot_rank ? 4 * slot_rank->size() : 0, b_amap = slot_rank ? 0 : 4 * c->amap.size();
  size_t o_spans = al(b_plans + 16), o_tab = o_spans + al(b_spans + 16), o_x = o_tab + al(b_tab + 16), tables_bytes = o_x + al(std::max(b_rank, b_amap) + 16);
  if (!c->d_tables.ensure(tables_bytes) || !c->h_stage.ensure(tables_bytes)) return fail(c, AM355_E_NOMEM, "device allocation failed");
  uint8_t* d_tables = c->d_tables.as<uint8_t>();
  c->p_spans = (ActorSpan*)(d_tables + o_spans);
  c->p_tab_off = (uint32_t*)(d_tables + o_tab);
  // every merge run signals under its own sequence number: when the optimistic in-order run of this replay is discarded (the hash
  // stream found a late or missing dependency) the general path merges again, and must not take the first run's counters -- already
  // signalled under the replay's number -- for its own
  c->sig_seq++;
  int rcb = setup_buffers(c, (uint32_t)c->actors.size());
  if (rcb) return rcb;
  // decoder classes: changes whose columns fit the small LDS footprint first, then the large footprint, then the (rare)
  // ones with a column too long for LDS staging
  uint32_t n_small = 0, n_large = 0;
  {
    const ChangeBrief* br = c->hp_briefs;
    std::vector<ChangePlan> large, serial;
    size_t w = 0;
    for (size_t i = 0; i < np; i++) {
      uint32_t f = br[c->plans[i].change].flags_fits;
      if (f & 0x40000000u) c->plans[w++] = c->plans[i];
      else if (f & 0x80000000u) large.push_back(c->plans[i]);
      else serial.push_back(c->plans[i]);
    }
    n_small = (uint32_t)w;
    n_large = (uint32_t)large.size();
    for (auto& pl : large) c->plans[w++] = pl;
    for (auto& pl : serial) c->plans[w++] = pl;
  }
  // (pageable std::vector memory would make the copy a synchronous bounce through the driver's own staging)
  const uint32_t* d_amap;
  const uint32_t* d_rank = nullptr;
  {
    uint8_t* h = c->h_stage.as<uint8_t>();
    if (b_plans) memcpy(h, c->plans.data(), b_plans);
    if (b_spans) memcpy(h + o_spans, c->spans.data(), b_spans);
    if (b_tab) memcpy(h + o_tab, c->actor_tab_off.data(), b_tab);
    if (slot_rank) {
      if (b_rank) memcpy(h + o_x, slot_rank->data(), b_rank);
      d_amap = c->d_amap_prov.as<uint32_t>();
      d_rank = (const uint32_t*)(d_tables + o_x);
    } else {
      if (b_amap) memcpy(h + o_x, c->amap.data(), b_amap);
      d_amap = (const uint32_t*)(d_tables + o_x);
    }
    HIPCHK(c, hipMemcpyAsync(d_tables, h, o_x + std::max(b_rank, b_amap), hipMemcpyHostToDevice, st));
  }

  // ---- stage 1b: column decode; the zero-fills of the merge stage and the second decoder class run beside it on stream3 ----
  HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, c->mb.counts_bytes, st));
  HIPCHK(c, hipEventRecord(c->ev_fork, st));
  HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
  merge_prepare(c->mb, c->stream3);
  HIPCHK(c, hipEventRecord(c->ev[2], st));  // brackets the decode launch only
  launch_decode_columns(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), (const ChangePlan*)d_tables, n_small, n_large, (uint32_t)np - n_small - n_large, d_amap,
                        d_rank, c->cols, &c->d_counts.as<Counts>()->flags, st, c->stream3, c->shard_rank, c->shard_world);
  HIPCHK(c, hipEventRecord(c->ev[3], st));
  HIPCHK(c, hipEventRecord(c->ev_join, c->stream3));
  HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));

  // ---- stage 2: merge (the decode flags land in the same counter block and are read with the first counters) ----
  Counts* hc = c->h_counts.as<Counts>();
  merge_run(c->mb, c->ir, hc, st, (c->phase_events || !c->mb.sig) ? c->ev_counts : nullptr, c->ev_runs);
  HIPCHK(c, hipEventRecord(c->ev[5], st));
  if (hc->flags) return error_for_flags(c, hc->flags, "op set rejected");
  c->counts = *hc;
  c->counts.n_objects += 1;  // + _root
  return AM355_OK;
}

// In-order fast path with the device-side plan (k_plan): the decode kernels are launched from the device-built plans as soon as
// the host knows the totals; the host's own planning (sequence numbers, clock, per-actor span tables: plan_fast) runs while the
// decode kernels do, and its tables reach the device before k_resolve needs them.
// `go` != null (general path): the plans in d_plans are in the application order the device scheduler found (am355_sched.hip); the
// host's half runs over that order (go->order / go->pass, host copies complete at go->ready).
struct GeneralOrder { const uint32_t* order; const uint32_t* pass; uint32_t n_applied; hipEvent_t ready; };
static int run_device_planned(am355_ctx* c, const PlanTotals& tot, uint32_t n_distinct, float* ms_host_plan, const GeneralOrder* go = nullptr) {
  hipStream_t st = c->stream;
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "  planned: %-26s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  uint32_t n = c->n_changes;
  c->n_ops = tot.n_ops;
  c->n_preds = tot.n_preds;
  c->max_op = tot.max_op;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  // device block for the host-built tables: actor spans (at most one per change) | span offsets (one per actor + 1)
  size_t o_tab = al(sizeof(ActorSpan) * (size_t)n + 16), tables_bytes = o_tab + al(4 * ((size_t)n_distinct + 1) + 16);
  if (!c->d_tables.ensure(tables_bytes) || !c->h_stage.ensure(tables_bytes)) return fail(c, AM355_E_NOMEM, "device allocation failed");
  uint8_t* d_tables = c->d_tables.as<uint8_t>();
  c->p_spans = (ActorSpan*)d_tables;
  c->p_tab_off = (uint32_t*)(d_tables + o_tab);
  c->sig_seq++;  // (see run_device)
  int rcb = setup_buffers(c, n_distinct);
  if (rcb) return rcb;
  lap("buffers carved");
  // the decode launch first (every HIP call before it is device idle time); the merge stage's fills follow on stream3 -- they depend
  // on nothing of this replay -- and stream3 only waits for the counter reset when a second decoder class runs there
  if (!(c->counts_zeroed_at == c->d_counts.p && c->mb.counts_bytes <= c->counts_zeroed)) HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, c->mb.counts_bytes, st));
  c->counts_zeroed_at = nullptr;  // (one replay's worth: the merge kernels are about to write it)
  if (tot.n_small && (tot.n_large || tot.n_serial)) {
    HIPCHK(c, hipEventRecord(c->ev_fork, st));
    HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
  }
  if (c->phase_events) HIPCHK(c, hipEventRecord(c->ev[2], st));
  launch_decode_planned(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), c->d_plans.as<ChangePlan>(), c->d_plans.as<ChangePlan>() + std::max(n, 1u), n,
                        tot.n_small, tot.n_large, tot.n_serial, c->d_amap_prov.as<uint32_t>(), c->d_slot_rank.as<uint32_t>(), c->cols,
                        &c->d_counts.as<Counts>()->flags, st, c->stream3, c->shard_rank, c->shard_world);
  lap("decode launched");
  if (c->phase_events) HIPCHK(c, hipEventRecord(c->ev[3], st));
  if (tot.n_small && (tot.n_large || tot.n_serial)) {
    // (stream3 carries the second decoder class -- 0.27 ms for a batch of fat changes --: the fills, which depend on nothing, would
    // start behind it and k_resolve would wait for them; they go to the copy stream, which is idle now)
    merge_prepare(c->mb, c->stream4);
    HIPCHK(c, hipEventRecord(c->ev_fills, c->stream4));
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_fills, 0));
  } else {
    merge_prepare(c->mb, c->stream3);
  }
  HIPCHK(c, hipEventRecord(c->ev_join, c->stream3));
  HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
  lap("fills enqueued");
  // ---- host half of the plan, beside the decode kernels (the digests were copied right behind k_plan) ----
  HIPCHK(c, hipEventSynchronize(c->ev_s1));
  if (go) HIPCHK(c, hipEventSynchronize(go->ready));
  auto t0 = std::chrono::steady_clock::now();
  std::vector<uint32_t> slot_rank;
  int rc = go ? plan_fast(c, slot_rank, go->order, go->n_applied, go->pass) : plan_fast(c, slot_rank);
  *ms_host_plan = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  lap("plan_fast done");
  if (rc == AM355_OK && (c->n_ops != tot.n_ops || c->n_preds != tot.n_preds || c->max_op != tot.max_op || c->actors.size() != n_distinct))
    rc = fail(c, AM355_E_DEVICE, "internal: device and host plans disagree (%llu / %u ops)", (unsigned long long)c->n_ops, tot.n_ops);
  if (rc) { (void)hipStreamSynchronize(st); return rc; }
  {
    uint8_t* h = c->h_stage.as<uint8_t>();
    size_t b_spans = sizeof(ActorSpan) * c->spans.size(), b_tab = 4 * c->actor_tab_off.size();
    if (b_spans) memcpy(h, c->spans.data(), b_spans);
    memcpy(h + o_tab, c->actor_tab_off.data(), b_tab);
    // on stream4, beside the decode kernels: in the main stream the copy would start when the decode kernels end (and on
    // stream3 when the merge fills end) and k_resolve would wait for it
    HIPCHK(c, hipMemcpyAsync(d_tables, h, o_tab + b_tab, hipMemcpyHostToDevice, c->stream4));
    HIPCHK(c, hipEventRecord(c->ev_tables, c->stream4));
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_tables, 0));
  }
  lap("tables enqueued");
  Counts* hc = c->h_counts.as<Counts>();
  merge_run(c->mb, c->ir, hc, st, (c->phase_events || !c->mb.sig) ? c->ev_counts : nullptr, c->ev_runs);
  HIPCHK(c, hipEventRecord(c->ev[5], st));
  if (hc->flags) return error_for_flags(c, hc->flags, "op set rejected");
  c->counts = *hc;
  c->counts.n_objects += 1;  // + _root
  return AM355_OK;
}

// Backend.load(bytes) + getPatch: device decode of the document's op columns, then the whole-document patch of the
// (already canonical) rows. new.js:1695-1750, 1604-1635.
static int replay_document(am355_ctx* c) {
  auto t_begin = std::chrono::steady_clock::now();
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "replay_document: %-28s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  hipStream_t st = c->stream;
  uint32_t NA = (uint32_t)c->actors.size();
  if (!c->d_plans.ensure(sizeof(ChangePlan)) || !c->d_amap.ensure(4 * (size_t)std::max(NA, 1u)) || !c->d_words.ensure(4 * W_NUM) || !c->h_words.ensure(4 * W_NUM))
    return fail(c, AM355_E_NOMEM, "device allocation failed");
  HIPCHK(c, hipEventRecord(c->ev[0], st));
  c->doc_col_rows.clear();
  if (c->doc_serial) {
    // first version: two lanes count rows / succ entries, then one lane per column group decodes value by value
    HIPCHK(c, hipMemcpyAsync(c->d_metas.p, &c->doc_meta, sizeof(ChangeMeta), hipMemcpyHostToDevice, st));
    launch_doc_count(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), st);
    HIPCHK(c, hipMemcpyAsync(c->h_metas.p, c->d_metas.p, sizeof(ChangeMeta), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    HIPCHK(c, hipStreamSynchronize(st));
    const ChangeMeta* hm = c->h_metas.as<ChangeMeta>();
    if (hm->flags) return error_for_flags(c, hm->flags, "malformed document columns");
    c->n_ops = hm->n_ops;
    c->n_preds = hm->n_preds;
    c->n_applied = c->n_changes;
    c->n_pending = 0;
    c->max_op = 0xffffffffu >> 8;  // only sizes sort keys, which the document path never builds
    if (c->n_ops >= 0x7ffffff0ull) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "more than 2^31 rows in one document"); }
    int rc = setup_buffers(c, NA);
    if (rc) return rc;
    ChangePlan pl{0, 0, 0, 0, NONE32, NA};  // author NONE32 = document mode: ids come from the idActor / idCtr columns
    HIPCHK(c, hipMemcpyAsync(c->d_plans.p, &pl, sizeof pl, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_amap.p, c->doc_actor_rank.data(), 4 * (size_t)NA, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, c->mb.counts_bytes, st));
    HIPCHK(c, hipMemsetAsync(c->d_words.p, 0, 4 * W_NUM, st));
    HIPCHK(c, hipEventRecord(c->ev[2], st));
    launch_decode_document(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), c->d_plans.as<ChangePlan>(), c->d_amap.as<uint32_t>(), c->cols,
                           &c->d_counts.as<Counts>()->flags, st);
  } else {
    // parallel column decode (am355_bigcol.hip); the keyStr column is indexed on the second stream meanwhile
    const BigColDesc& d = c->doc_cols;
    const ChangeMeta& m = c->doc_meta;
    if (!c->d_big.ensure(bigcol_work_bytes(d.tok_bytes)) || !c->d_ks.ensure(keystr_work_bytes(m.col_len[C_KEY_STR])) || !c->h_biginfo.ensure(sizeof(BigColInfo)))
      return fail(c, AM355_E_NOMEM, "device allocation failed (document index)");
    BigColWork w;
    canary_forget(c->d_big.p, c->d_big.cap);
    bigcol_carve(w, c->d_big.p, d.tok_bytes);
    canary_arm();
    uint32_t *ks_start, *ks_off, *ks_len;
    uint32_t* d_words = c->d_words.as<uint32_t>();
    HIPCHK(c, hipMemsetAsync(d_words, 0, 4 * W_NUM, st));
    HIPCHK(c, hipMemcpyAsync(c->d_amap.p, c->doc_actor_rank.data(), 4 * (size_t)NA, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev_b0, st));
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_b0, 0));
    KeyStage ks;
    canary_forget(c->d_ks.p, c->d_ks.cap);
    keystr_index_begin(c->d_arena.as<uint8_t>(), m.col_off[C_KEY_STR], m.col_len[C_KEY_STR], c->d_ks.p, ks, d_words + W_TOTAL_ENTRIES, d_words + W_FAST_B,
                       c->stream2);
    // (the key stream needs no host decision any more: both halves are enqueued at once and run beside the token index)
    keystr_index_finish(ks, false, &ks_start, &ks_off, &ks_len, d_words + W_FLAGS_B, c->stream2);
    HIPCHK(c, hipEventRecord(c->ev_b1, c->stream2));
    BigColInfo* hi = c->h_biginfo.as<BigColInfo>();
    bigcol_index_tokens(c->d_arena.as<uint8_t>(), d, w, st);
    HIPCHK(c, hipMemcpyAsync(hi, w.info, sizeof(BigColInfo), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));  // number count: everything after runs over numbers, not bytes
    bigcol_index_records(c->d_arena.as<uint8_t>(), d, w, hi->n_tokens, st);
    HIPCHK(c, hipMemcpyAsync(hi, w.info, sizeof(BigColInfo), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    lap("enqueued index");
    HIPCHK(c, hipStreamSynchronize(st));
    lap("token index done");
    if (hi->flags) { (void)hipStreamSynchronize(c->stream2); return error_for_flags(c, hi->flags, "malformed document columns"); }
    BigColInfo info = *hi;
    c->doc_col_rows.assign(info.rows, info.rows + BIG_NCOL);
    uint32_t N = info.rows[BC_ACTION], Pcap = info.rows[BC_SUCC_ACTOR];
    if (N >= 0x7ffffff0u) { (void)hipStreamSynchronize(c->stream2); c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "more than 2^31 rows in one document"); }
    if (!c->d_bigvals.ensure(bigcol_vals_bytes(N, Pcap))) { (void)hipStreamSynchronize(c->stream2); return fail(c, AM355_E_NOMEM, "device allocation failed (document columns)"); }
    BigColVals v;
    canary_forget(c->d_bigvals.p, c->d_bigvals.cap);
    bigcol_carve_vals(v, c->d_bigvals.p, N, Pcap);
    canary_arm();
    HIPCHK(c, hipEventRecord(c->ev[2], st));
    bigcol_expand(d, w, info, v, N, Pcap, st);
    HIPCHK(c, hipMemcpyAsync(hi, w.info, sizeof(BigColInfo), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    lap("expand done");
    if (hi->flags) { (void)hipStreamSynchronize(c->stream2); return error_for_flags(c, hi->flags, "malformed document columns"); }
    if (hi->n_succ > Pcap) { (void)hipStreamSynchronize(c->stream2); return error_for_flags(c, AM355_F_UNSUPPORTED, "succ columns shorter than succNum announces"); }
    c->n_ops = N;
    c->n_preds = hi->n_succ;
    c->n_applied = c->n_changes;
    c->n_pending = 0;
    c->max_op = 0xffffffffu >> 8;  // only sizes sort keys, which the document path never builds
    int rc = setup_buffers(c, NA);
    if (rc) { (void)hipStreamSynchronize(c->stream2); return rc; }
    HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, c->mb.counts_bytes, st));
    uint32_t* flags = &c->d_counts.as<Counts>()->flags;
    bigcol_assemble(v, N, (uint32_t)c->n_preds, c->d_amap.as<uint32_t>(), NA, m.col_off[C_VAL_RAW], m.col_len[C_VAL_RAW], c->cols, flags, st);
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_b1, 0));
    launch_keystr_expand(ks_start, ks_off, ks_len, d_words + W_TOTAL_ENTRIES, N, c->cols.key_off, c->cols.key_len, st);
    HIPCHK(c, hipMemcpyAsync(c->h_words.as<uint32_t>() + W_FLAGS_B, d_words + W_FLAGS_B, 4, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipEventRecord(c->ev[3], st));
  // maxOp = max over op ids and succ counters (new.js:1627-1630)
  uint32_t* d_max = c->d_words.as<uint32_t>();
  max_u32(c->cols.id_ctr, (uint32_t)c->n_ops, d_max, st);
  max_u32(c->cols.pred_ctr, (uint32_t)c->n_preds, d_max, st);
  HIPCHK(c, hipMemcpyAsync(c->h_words.p, d_max, 4, hipMemcpyDeviceToHost, st));
  Counts* hc = c->h_counts.as<Counts>();
  doc_patch(c->mb, c->ir, hc, st);
  HIPCHK(c, hipEventRecord(c->ev[4], st));
  HIPCHK(c, hipEventRecord(c->ev[5], st));
  HIPCHK(c, hipStreamSynchronize(st));
  lap("patch done");
  if (!c->doc_serial && c->h_words.as<uint32_t>()[W_FLAGS_B]) return error_for_flags(c, c->h_words.as<uint32_t>()[W_FLAGS_B], "malformed key column");
  if (hc->flags) return error_for_flags(c, hc->flags, "document rejected");
  c->max_op = c->h_words.as<uint32_t>()[0];
  c->counts = *hc;
  c->counts.n_objects += 1;  // + _root
  auto t_end = std::chrono::steady_clock::now();
  am355_stats& s = c->stats;
  s.n_changes = c->n_changes; s.n_applied = c->n_changes; s.n_pending = 0; s.n_actors = NA; s.n_objects = c->counts.n_objects;
  s.n_heads = (uint32_t)(c->heads.size() / 32); s.n_ops = c->n_ops; s.max_op = c->max_op; s.raw_bytes = c->raw.size();
  s.n_map_values = c->counts.n_map_emit; s.n_list_elems = c->counts.n_list_ins; s.n_edits = c->counts.n_edits;
  s.ir_bytes = (uint64_t)c->counts.n_objects * sizeof(am355_ir_object) + (uint64_t)c->counts.n_map_emit * sizeof(am355_ir_map) +
               ((uint64_t)c->counts.n_erecs + 1) * sizeof(am355_ir_edit);
  (void)hipEventElapsedTime(&s.ms_parse, c->ev[0], c->ev[1]);
  (void)hipEventElapsedTime(&s.ms_decode, c->ev[2], c->ev[3]);
  (void)hipEventElapsedTime(&s.ms_merge, c->ev[3], c->ev[4]);
  s.ms_order = 0; s.ms_hash_stream = 0; s.ms_host_schedule = 0; s.fast_path = 1;
  s.ms_total = std::chrono::duration<float, std::milli>(t_end - t_begin).count();
  c->replayed = true;
  return AM355_OK;
}

static int replay_impl(am355_ctx* c) {
  if (!c) return AM355_E_ARG;
  if (!c->staged) return fail(c, AM355_E_STATE, "am355_load_changes must be called first");
  (void)hipSetDevice(c->device);
  c->replayed = c->ir_fetched = false;
  c->dep_graph_ready = false;
  c->flags = 0;
  if (c->is_document) return replay_document(c);
  auto t_begin = std::chrono::steady_clock::now();
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "replay: %-28s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  hipStream_t sa = c->stream, sb = c->stream2;
  uint32_t n = c->n_changes;
  size_t n1 = std::max<size_t>(n, 1);
  c->slot_mask = pow2_at_least(4 * (uint64_t)n + 64) - 1;
  c->hash_mask = pow2_at_least(2 * (uint64_t)n + 64) - 1;
  if (c->amap_cap < 16 * n1 + 1024) c->amap_cap = (uint32_t)(16 * n1 + 1024);
  if (!c->d_entries.ensure(4 * n1) || !c->d_amap_base.ensure(4 * (n1 + 1)) || !c->d_amap_prov.ensure(4 * (size_t)c->amap_cap) ||
      !c->d_slots.ensure(8 * (size_t)(c->slot_mask + 1)) || !c->d_first_idx.ensure(4 * (size_t)(c->slot_mask + 1)) || !c->d_hashes.ensure(32 * n1) ||
      !c->d_hash_tab.ensure(4 * (size_t)(c->hash_mask + 1)) || !c->d_min_idx.ensure(4 * n1) || !c->d_has_dep.ensure(n1) || !c->d_words.ensure(4 * W_NUM) ||
      !c->d_scan1.ensure(scan_workspace_bytes((uint32_t)n1)) || !c->h_slots.ensure(8 * (size_t)(c->slot_mask + 1)) || !c->h_hashes.ensure(32 * n1) ||
      !c->h_has_dep.ensure(n1) || !c->h_words.ensure(4 * W_NUM) || !c->d_plans.ensure(2 * sizeof(ChangePlan) * n1) ||
      !c->d_slot_rank.ensure(4 * (size_t)(c->slot_mask + 1)) || !c->d_plan_sums.ensure(plan_block_sums_bytes(n)) ||
      !c->d_dep_idx.ensure(4 * (c->raw.size() / 32 + 2)) || !c->d_self_idx.ensure(4 * n1) || !c->d_rank_ids.ensure(rank_ids_bytes()))
    return fail(c, AM355_E_NOMEM, "device allocation failed (stage 1)");
  c->have_host_metas = false;
  // what the host reads after stage 1 -- a few flag words, the distinct actor ids, one brief per change -- sits in one device
  // block: one memset clears the words and the distinct counter, one copy brings everything back
  const size_t s1_distinct = 64, s1_briefs = s1_distinct + ((12 * (size_t)distinct_capacity() + 16 + 63) & ~(size_t)63);
  const size_t s1_bytes = s1_briefs + sizeof(ChangeBrief) * n1;
  if (!c->d_s1.ensure(s1_bytes) || !c->h_s1.ensure(s1_bytes)) return fail(c, AM355_E_NOMEM, "device allocation failed (stage 1)");
  uint32_t* d_wa = c->d_s1.as<uint32_t>();                        // W_FLAGS_A, W_FAST_A, W_TOTAL_ENTRIES
  uint32_t* d_distinct = (uint32_t*)(c->d_s1.as<uint8_t>() + s1_distinct);
  ChangeBrief* d_briefs = (ChangeBrief*)(c->d_s1.as<uint8_t>() + s1_briefs);
  const uint32_t* h_wa = c->h_s1.as<uint32_t>();
  c->hp_distinct = (uint32_t*)(c->h_s1.as<uint8_t>() + s1_distinct);
  c->hp_briefs = (ChangeBrief*)(c->h_s1.as<uint8_t>() + s1_briefs);
  uint32_t* d_words = c->d_words.as<uint32_t>();  // stream B's words (W_FLAGS_B, W_FAST_B)
  uint32_t* h_words = c->h_words.as<uint32_t>();
  HostSignals* sig = c->h_sig.as<HostSignals>();
  PlanTotals tot{};
  static const bool hash_after_parse = []() { const char* e = getenv("AM355_HASH_START"); return !(e && !strcmp(e, "intern")); }();

  // ---- stream A: parse. The fills of stage 1 (flag words, actor hash table) depend on nothing of this replay: they run on stream3
  //      beside the parse kernel instead of in front of the kernels that need them ----
  HIPCHK(c, hipEventRecord(c->ev[0], sa));
  // the merge stage's counter block too: its size follows from the op count, which is at most one op per encoded byte for any
  // batch worth hurrying (a run length may claim more: the planned path then clears it in front of the decode as before)
  const size_t cb = merge_counts_bytes((uint32_t)std::min<size_t>(c->raw.size(), 0x7ffffff0u));
  c->counts_zeroed_at = nullptr;
  const bool counts_too = c->d_counts.ensure(cb);
  if (c->inline_fills) {
    // (cleared by the parse kernel's workgroups on their way in: no second stream, no event wait in front of the next kernel)
    ParseFills f{};
    auto add = [&](void* q, size_t bytes, uint32_t v) { f.p[f.n] = (uint32_t*)q; f.n_words[f.n] = (bytes + 3) / 4; f.value[f.n] = v; f.n++; };
    add(d_words, 4 * W_NUM, 0);
    add(d_wa, s1_distinct + 16, 0);
    add(c->d_slots.p, 8 * (size_t)(c->slot_mask + 1), 0);
    add(c->d_first_idx.p, 4 * (size_t)(c->slot_mask + 1), 0xffffffffu);
    if (counts_too) add(c->d_counts.p, cb, 0);
    launch_parse_changes(c->d_arena.as<uint8_t>(), c->d_offsets.as<uint64_t>(), n, c->d_metas.as<ChangeMeta>(), c->d_entries.as<uint32_t>(), f, sa);
    HIPCHK(c, hipEventRecord(c->ev_parse, sa));
  } else {
    launch_parse_changes(c->d_arena.as<uint8_t>(), c->d_offsets.as<uint64_t>(), n, c->d_metas.as<ChangeMeta>(), c->d_entries.as<uint32_t>(), ParseFills{}, sa);
    HIPCHK(c, hipEventRecord(c->ev_parse, sa));
    HIPCHK(c, hipMemsetAsync(d_words, 0, 4 * W_NUM, c->stream3));
    HIPCHK(c, hipMemsetAsync(d_wa, 0, s1_distinct + 16, c->stream3));
    HIPCHK(c, hipMemsetAsync(c->d_slots.p, 0, 8 * (size_t)(c->slot_mask + 1), c->stream3));
    HIPCHK(c, hipMemsetAsync(c->d_first_idx.p, 0xff, 4 * (size_t)(c->slot_mask + 1), c->stream3));
    if (counts_too) HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, cb, c->stream3));
    HIPCHK(c, hipEventRecord(c->ev_join, c->stream3));
    HIPCHK(c, hipStreamWaitEvent(sa, c->ev_join, 0));
  }
  if (counts_too) {
    c->counts_zeroed_at = c->d_counts.p;
    c->counts_zeroed = cb;
  }

  // ---- stream B: SHA-256 of every change, hash table, dependency resolution; joined at the very end. Its commands are enqueued
  //      behind the stage-1 kernels of stream A. AM355_HASH_ENQUEUE=early enqueues them right behind the parse launch, which starts
  //      the SHA kernel ~35 us sooner; measured on the same box (profiles/r03_ab_hash_enqueue.txt) that costs the replay 0.14 ms:
  //      the kernels between decode and the compaction take 0.25 instead of 0.10 ms with stream B's commands queued first ----
  auto enqueue_stream_b = [&]() -> int {
    // (it starts after the parse kernel -- AM355_HASH_START=intern: after the actor kernels --: those grids are as small as the hash
    // grid, one wave per 64 changes, and the ALU-dense SHA waves would otherwise share their SIMDs and slow them down)
    HIPCHK(c, hipStreamWaitEvent(sb, hash_after_parse ? c->ev_parse : c->ev[1], 0));
    if (!c->inline_fills) HIPCHK(c, hipStreamWaitEvent(sb, c->ev_join, 0));  // (its flag words are cleared on stream3)
    HIPCHK(c, hipEventRecord(c->ev_b0, sb));
    HIPCHK(c, hipMemsetAsync(c->d_hash_tab.p, 0, 4 * (size_t)(c->hash_mask + 1), sb));
    HIPCHK(c, hipMemsetAsync(c->d_has_dep.p, 0, n1, sb));
    launch_hash_changes(c->d_arena.as<uint8_t>(), c->d_offsets.as<uint64_t>(), n, c->d_hashes.as<uint8_t>(), c->d_min_idx.as<uint32_t>(),
                    c->d_hash_tab.as<uint32_t>(), c->hash_mask, d_words + W_FLAGS_B, sb);
    launch_deps_resolve(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), c->d_hashes.as<uint8_t>(), n, c->d_hash_tab.as<uint32_t>(), c->hash_mask,
                    c->d_min_idx.as<uint32_t>(), c->d_has_dep.as<uint8_t>(), d_words + W_FAST_B, c->d_dep_idx.as<uint32_t>(), c->d_self_idx.as<uint32_t>(), sb);
    HIPCHK(c, hipMemcpyAsync(c->h_hashes.p, c->d_hashes.p, 32 * (size_t)n, hipMemcpyDeviceToHost, sb));
    HIPCHK(c, hipMemcpyAsync(c->h_has_dep.p, c->d_has_dep.p, n, hipMemcpyDeviceToHost, sb));
    HIPCHK(c, hipMemcpyAsync(h_words + W_FLAGS_B, d_words + W_FLAGS_B, 8, hipMemcpyDeviceToHost, sb));
    HIPCHK(c, hipEventRecord(c->ev_b1, sb));
    return AM355_OK;
  };
  static const bool enqueue_early = []() { const char* e = getenv("AM355_HASH_ENQUEUE"); return e && !strcmp(e, "early"); }();
  if (hash_after_parse && enqueue_early) { int rb = enqueue_stream_b(); if (rb) return rb; }
  exclusive_scan_u32(c->d_entries.as<uint32_t>(), c->d_amap_base.as<uint32_t>(), n, d_wa + W_TOTAL_ENTRIES, c->d_scan1.p, sa);
  for (int attempt = 0;; attempt++) {
    if (attempt) {
      HIPCHK(c, hipMemsetAsync(c->d_slots.p, 0, 8 * (size_t)(c->slot_mask + 1), sa));
      HIPCHK(c, hipMemsetAsync(c->d_first_idx.p, 0xff, 4 * (size_t)(c->slot_mask + 1), sa));
      HIPCHK(c, hipMemsetAsync(d_distinct, 0, 4, sa));
    }
    launch_actor_intern(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), n, c->d_amap_base.as<uint32_t>(), c->d_amap_prov.as<uint32_t>(), c->amap_cap,
                        c->d_slots.as<unsigned long long>(), c->slot_mask, c->d_first_idx.as<uint32_t>(), d_wa + W_FLAGS_A, d_wa + W_FAST_A,
                        d_distinct, c->d_rank_ids.p, d_briefs, c->d_slot_rank.as<uint32_t>(), c->d_plan_sums.as<unsigned long long>(), d_wa + 8, sa);
    // device half of the in-order plan (actor ranks, per-change bases, decoder classes): the decode kernels start from it
    // (its totals, and the stage-1 words the host decides on, reach the host through HostSignals: no copy, no blocking wait)
    c->sig_seq++;
    launch_plan(d_briefs, n, d_distinct, c->d_slot_rank.as<uint32_t>(), c->slot_mask, c->d_plan_sums.as<unsigned long long>(), c->d_plans.as<ChangePlan>(),
                c->d_plans.as<ChangePlan>() + n1, d_wa, d_wa + 8, sig, c->sig_seq, sa);
    // the host's own half of the plan needs a 32-byte digest per change and the handful of distinct actor ids: they follow
    // -- on stream4, behind the plan kernel: in stream A the copy (and its dispatch gap) would sit in front of the decode kernels
    HIPCHK(c, hipEventRecord(c->ev_plan, sa));
    HIPCHK(c, hipStreamWaitEvent(c->stream4, c->ev_plan, 0));
    HIPCHK(c, hipMemcpyAsync(c->h_s1.p, c->d_s1.p, s1_briefs + sizeof(ChangeBrief) * n, hipMemcpyDeviceToHost, c->stream4));
    HIPCHK(c, hipEventRecord(c->ev_s1, c->stream4));
    if (c->phase_events || !hash_after_parse) HIPCHK(c, hipEventRecord(c->ev[1], sa));
    if (attempt == 0 && !(hash_after_parse && enqueue_early)) { int rb = enqueue_stream_b(); if (rb) return rb; }
    lap("stage 1 enqueued");
    if (!wait_host_signal(&sig->plan_seq, c->sig_seq, sa)) {
      (void)hipStreamSynchronize(sb);
      (void)hipStreamSynchronize(c->stream3);
      return fail(c, AM355_E_DEVICE, "the device did not report the plan of this replay (%s)", hipGetErrorString(hipGetLastError()));
    }
    memcpy(&tot, (const void*)&sig->plan, sizeof tot);
    lap("stage 1 totals read");
    if (!(tot.fast_a & FF_CAPACITY) || attempt) break;
    // the staging buffer for actor tables was too small: grow to the measured total and redo the interning
    c->amap_cap = tot.total_entries + 1024;
    if (!c->d_amap_prov.ensure(4 * (size_t)c->amap_cap)) return fail(c, AM355_E_NOMEM, "device allocation failed (actor tables)");
    HIPCHK(c, hipMemsetAsync(d_wa + W_FAST_A, 0, 4, sa));
    HIPCHK(c, hipMemsetAsync(d_wa + 8, 0, 32, sa));  // (plan words)
  }

  // ---- host: flags, in-order plan ----
  auto t_h0 = std::chrono::steady_clock::now();
  float ms_host = 0;
  int rc = AM355_OK;
  // (tot.flags_a: validity flags of the stage-1 kernels OR'ed with those of every change; k_plan saw all the digests)
  if (tot.flags_a) { (void)hipStreamSynchronize(sa); (void)hipStreamSynchronize(sb); return error_for_flags(c, tot.flags_a, "malformed change"); }
  c->has_unknown_cols = tot.reserved[0] != 0;
  bool fast = tot.fast_a == 0;
  if (tot.n_distinct > distinct_capacity()) fast = false;  // thousands of actors: the general path interns them on the host
  const bool planned = !tot.fallback && !getenv("AM355_HOST_PLAN");
  std::vector<uint32_t> slot_rank;
  int opt_rc = AM355_OK;
  uint32_t opt_flags = 0;
  std::string opt_err;
  if (fast && planned) {
    float ms_plan = 0;
    opt_rc = run_device_planned(c, tot, tot.n_distinct, &ms_plan);  // optimistic: confirmed (or discarded) when stream B is joined
    ms_host += ms_plan;  // (host planning time; it runs beside the decode kernels)
    lap("run_device (device plan) done");
    opt_flags = c->flags;
    opt_err = c->err;
  } else {
    HIPCHK(c, hipEventSynchronize(c->ev_s1));  // the digests
    if (tot.fallback) {  // (k_plan stopped before it looked at the changes: their flags come from the digests)
      const ChangeBrief* br = c->hp_briefs;
      uint32_t dev_flags = 0;
      c->has_unknown_cols = false;
      for (uint32_t i = 0; i < n; i++) {
        dev_flags |= br[i].flags_fits & 0x1fffffffu;
        if (br[i].flags_fits & 0x20000000u) c->has_unknown_cols = true;
      }
      if (dev_flags) { (void)hipStreamSynchronize(sa); (void)hipStreamSynchronize(sb); return error_for_flags(c, dev_flags, "malformed change"); }
    }
    if (fast) {
      opt_rc = plan_fast(c, slot_rank);
      ms_host += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_h0).count();
      lap("plan_fast done");
      if (opt_rc == AM355_OK) opt_rc = run_device(c, &slot_rank);  // optimistic: confirmed (or discarded) when stream B is joined
      lap("run_device done");
      opt_flags = c->flags;
      opt_err = c->err;
    }
  }
  // ---- join stream B ----
  HIPCHK(c, hipEventSynchronize(c->ev_b1));
  lap("hash stream joined");
  if (h_words[W_FLAGS_B]) return error_for_flags(c, h_words[W_FLAGS_B], "checksum does not match data");
  if (fast && h_words[W_FAST_B]) fast = false;
  if (fast) {
    if (opt_rc != AM355_OK) { c->flags = opt_flags; c->err = opt_err; return opt_rc; }
    // heads: changes nobody depends on, sorted (new.js:1582-1583, 1593)
    auto t0 = std::chrono::steady_clock::now();
    const uint8_t* hs = c->h_hashes.as<uint8_t>();
    const uint8_t* dep = c->h_has_dep.as<uint8_t>();
    std::vector<const uint8_t*> heads;
    for (uint32_t i = 0; i < n; i++)
      if (!dep[i]) heads.push_back(hs + 32 * (size_t)i);
    std::sort(heads.begin(), heads.end(), [](const uint8_t* x, const uint8_t* y) { return memcmp(x, y, 32) < 0; });
    c->heads.resize(heads.size() * 32);
    for (size_t i = 0; i < heads.size(); i++) memcpy(&c->heads[32 * i], heads[i], 32);
    ms_host += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  } else {
    // ---- general path (any delivery order, duplicates, missing dependencies) ----
    c->flags = 0;
    // the device's actor tables, when its list of distinct ids holds them all (else the host interns)
    const bool dev_actors = tot.n_distinct <= distinct_capacity() && !(tot.fast_a & FF_CAPACITY) && tot.total_entries <= c->amap_cap;
    bool served = false;
    const bool host_schedule = getenv("AM355_HOST_SCHEDULE") != nullptr;  // (A/B and tests: the host's scheduler for every batch)
    if (dev_actors && planned && !host_schedule && n > 0) {
      // The scheduler runs on the device (am355_sched.hip): pass numbers by relaxation over the dependency indexes stream B resolved,
      // application order by a stable sort, the decode plans in that order. The host reads the totals from the pinned words, launches
      // the decode kernels from the device-built plans and does its own half (sequence numbers, clock, span tables) beside them, as
      // on the in-order path.
      if (!c->d_sched.ensure(sched_bytes(n, c->slot_mask)) || !c->h_sched.ensure(13 * (size_t)n1 + 64)) return fail(c, AM355_E_NOMEM, "device allocation failed (scheduler)");
      SchedBufs sbuf;
      sched_bind(sbuf, c->d_sched.p, n, c->slot_mask);
      canary_arm();
      c->sig_seq++;
      uint32_t* d_order = nullptr;
      launch_sched_general(c->d_metas.as<ChangeMeta>(), d_briefs, n, c->d_dep_idx.as<uint32_t>(), c->d_self_idx.as<uint32_t>(), c->d_amap_prov.as<uint32_t>(),
                           c->d_amap_base.as<uint32_t>(), c->amap_cap, c->d_slot_rank.as<uint32_t>(), c->slot_mask, sbuf, &d_order, c->d_plans.as<ChangePlan>(),
                           c->d_plans.as<ChangePlan>() + n1, d_wa, d_distinct, sig, c->sig_seq, sa);
      // what the host's half needs: order | pass | first copies | head marks -- on the copy stream, behind the scheduler
      uint32_t* h_order = c->h_sched.as<uint32_t>();
      uint32_t *h_pass = h_order + n1, *h_self = h_pass + n1;
      uint8_t* h_is_head = (uint8_t*)(h_self + n1);
      HIPCHK(c, hipEventRecord(c->ev_plan, sa));
      HIPCHK(c, hipStreamWaitEvent(c->stream4, c->ev_plan, 0));
      HIPCHK(c, hipMemcpyAsync(h_order, d_order, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream4));
      HIPCHK(c, hipMemcpyAsync(h_pass, sbuf.pass, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream4));
      HIPCHK(c, hipMemcpyAsync(h_self, c->d_self_idx.p, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream4));
      HIPCHK(c, hipMemcpyAsync(h_is_head, sbuf.is_head, (size_t)n, hipMemcpyDeviceToHost, c->stream4));
      HIPCHK(c, hipEventRecord(c->ev_sched, c->stream4));
      if (!wait_host_signal(&sig->plan_seq, c->sig_seq, sa)) {
        (void)hipStreamSynchronize(c->stream4);
        return fail(c, AM355_E_DEVICE, "the device did not report the schedule of this replay (%s)", hipGetErrorString(hipGetLastError()));
      }
      PlanTotals gen{};
      memcpy(&gen, (const void*)&sig->plan, sizeof gen);
      lap("device schedule read");
      // the relaxation ran out of sweeps, sums beyond 32 bits, or an actor named before its first change: the host's scheduler decides
      // (and raises the exact flags of an invalid batch)
      if (!gen.reserved[3] && !gen.flags_a && !gen.fallback) {
        GeneralOrder go{h_order, h_pass, gen.reserved[1], c->ev_sched};
        float ms_plan = 0;
        rc = run_device_planned(c, gen, tot.n_distinct, &ms_plan, &go);
        ms_host += ms_plan;
        if (rc) return rc;
        auto t0 = std::chrono::steady_clock::now();
        // what stays queued: the changes of which no copy is ever applied (new.js:1566, 1866)
        // (copies of one change: whichever copy became ready first was applied, the others were dropped as duplicates then)
        c->pending_change.clear();
        std::vector<uint8_t> group_applied(n, 0);
        for (uint32_t ci = 0; ci < n; ci++)
          if (h_pass[ci] != SCHED_NEVER) group_applied[h_self[ci] < n ? h_self[ci] : ci] = 1;
        for (uint32_t ci = 0; ci < n; ci++)
          if (!group_applied[h_self[ci] < n ? h_self[ci] : ci]) c->pending_change.push_back(ci);
        c->n_pending = (uint32_t)c->pending_change.size();
        const uint8_t* hs = c->h_hashes.as<uint8_t>();
        std::vector<const uint8_t*> heads;
        for (uint32_t i = 0; i < n; i++)
          if (h_is_head[i]) heads.push_back(hs + 32 * (size_t)i);
        std::sort(heads.begin(), heads.end(), [](const uint8_t* x, const uint8_t* y) { return memcmp(x, y, 32) < 0; });
        c->heads.resize(heads.size() * 32);
        for (size_t i = 0; i < heads.size(); i++) memcpy(&c->heads[32 * i], heads[i], 32);
        ms_host += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        served = true;
      } else {
        (void)hipStreamSynchronize(c->stream4);
      }
    }
    if (!served) {
      // exact scheduling on the host (thousands of actors, pathological dependency chains, and every batch the reference rejects:
      // the flags it raises are the host's), then decode / merge of exactly the applied changes
      size_t dep_words = c->raw.size() / 32 + 2;
      if (!c->h_dep_idx.ensure(4 * dep_words) || !c->h_self_idx.ensure(4 * n1)) return fail(c, AM355_E_NOMEM, "host allocation failed");
      HIPCHK(c, hipMemcpyAsync(c->h_metas.p, c->d_metas.p, sizeof(ChangeMeta) * n, hipMemcpyDeviceToHost, sa));
      HIPCHK(c, hipMemcpyAsync(c->h_dep_idx.p, c->d_dep_idx.p, 4 * dep_words, hipMemcpyDeviceToHost, sa));  // (stream B has been joined)
      HIPCHK(c, hipMemcpyAsync(c->h_self_idx.p, c->d_self_idx.p, 4 * (size_t)n, hipMemcpyDeviceToHost, sa));
      if (dev_actors) {
        if (!c->h_amap.ensure(4 * ((size_t)tot.total_entries + 1)) || !c->h_amap_base.ensure(4 * (n1 + 1))) return fail(c, AM355_E_NOMEM, "host allocation failed");
        HIPCHK(c, hipMemcpyAsync(c->h_amap.p, c->d_amap_prov.p, 4 * (size_t)tot.total_entries, hipMemcpyDeviceToHost, sa));
        HIPCHK(c, hipMemcpyAsync(c->h_amap_base.p, c->d_amap_base.p, 4 * (size_t)n, hipMemcpyDeviceToHost, sa));
      }
      HIPCHK(c, hipStreamSynchronize(sa));
      if (dev_actors) c->h_amap_base.as<uint32_t>()[n] = tot.total_entries;
      HIPCHK(c, hipEventSynchronize(c->ev_s1));  // (the distinct-actor list rides with the digests)
      auto t0 = std::chrono::steady_clock::now();
      rc = schedule(c, dev_actors ? c->h_amap.as<uint32_t>() : nullptr, dev_actors ? c->h_amap_base.as<uint32_t>() : nullptr);
      ms_host += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rc) return rc;
      rc = run_device(c, nullptr);
      if (rc) return rc;
    }
    c->device_scheduled = served;
  }
  c->used_fast_path = fast;
  lap("end");
  auto t_end = std::chrono::steady_clock::now();

  am355_stats& s = c->stats;
  uint32_t NA = (uint32_t)c->actors.size();
  s.n_changes = n; s.n_applied = c->n_applied; s.n_pending = c->n_pending; s.n_actors = NA; s.n_objects = c->counts.n_objects;
  s.n_heads = (uint32_t)(c->heads.size() / 32); s.n_ops = c->n_ops; s.max_op = c->max_op; s.raw_bytes = c->raw.size();
  s.n_map_values = c->counts.n_map_emit; s.n_list_elems = c->counts.n_list_ins; s.n_edits = c->counts.n_edits;
  s.ir_bytes = (uint64_t)c->counts.n_objects * sizeof(am355_ir_object) + (uint64_t)c->counts.n_map_emit * sizeof(am355_ir_map) +
               ((uint64_t)c->counts.n_erecs + 1) * sizeof(am355_ir_edit);
  // (the last kernel has signalled its counters; its remaining workgroups retire within microseconds: poll, do not block)
  while (hipEventQuery(c->ev[5]) == hipErrorNotReady) {}
  s.ms_parse = s.ms_decode = s.ms_merge = s.ms_order = 0;
  if (c->phase_events) {
    (void)hipEventElapsedTime(&s.ms_parse, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&s.ms_decode, c->ev[2], c->ev[3]);
  }
  if (c->n_ops && c->phase_events) {  // (ev_counts: after resolve / emit / compaction, before the ordering kernels)
    (void)hipEventElapsedTime(&s.ms_merge, c->ev[3], c->ev_counts);
    (void)hipEventElapsedTime(&s.ms_order, c->ev_counts, c->ev[5]);
  }
  (void)hipEventElapsedTime(&s.ms_hash_stream, c->ev_b0, c->ev_b1);  // hash stream (SHA-256 + dependency resolution), overlapped
  s.ms_host_schedule = ms_host;
  s.fast_path = fast ? 1 : (c->device_scheduled ? 2 : 0);
  s.ms_total = std::chrono::duration<float, std::milli>(t_end - t_begin).count();
  c->replayed = true;
  if (!c->in_apply) {  // (one call of Backend.loadChanges: its scheduling passes are the op streams)
    c->stream_breaks = c->pass_first_row;
    c->breaks_exact = true;
    c->children_hazard = false;
    c->no_history = false;
  }
  return AM355_OK;
}

extern "C" int am355_set_phase_events(am355_ctx* c, int on) {
  if (!c) return AM355_E_ARG;
  c->phase_events = on != 0;
  return AM355_OK;
}

extern "C" int am355_get_stats(const am355_ctx* c, am355_stats* out) {
  if (!c || !out) return AM355_E_ARG;
  *out = c->stats;
  return AM355_OK;
}

extern "C" int am355_get_hashes(const am355_ctx* c, uint8_t* out) {
  if (!c || !out) return AM355_E_ARG;
  if (!c->replayed || !c->h_hashes.p || c->is_document) return AM355_E_STATE;  // a document stores no per-change hashes
  memcpy(out, c->h_hashes.p, 32 * (size_t)c->n_changes);
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
extern "C" int am355_get_applied(const am355_ctx* c, uint32_t* out, uint32_t* n_applied) {
  if (!c || !n_applied) return AM355_E_ARG;
  if (!c->replayed || c->is_document) return AM355_E_STATE;
  *n_applied = (uint32_t)c->applied_change.size();
  if (out) memcpy(out, c->applied_change.data(), 4 * c->applied_change.size());
  return AM355_OK;
}

// with_edits = false (am355_apply_changes): the object and map tables only -- the edit records of a text document are megabytes and
// setupPatches looks at them only when a touched object hangs in a list (c->hir.edits stays null; a later full fetch copies all three)
static int fetch_ir_impl(am355_ctx* c, am355_patch_ir* out, bool with_edits = true) {
  if (!c) return AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  (void)hipSetDevice(c->device);
  if (!c->ir_fetched) {
    hipStream_t st = c->stream;
    uint32_t NO = c->counts.n_objects, NM = c->counts.n_map_emit, NR = c->counts.n_erecs, NV = c->counts.n_edits;
    size_t bytes = carve_size(NO, sizeof(am355_ir_object)) + carve_size(NM, sizeof(am355_ir_map)) + carve_size((size_t)NR + 1, sizeof(am355_ir_edit)) + 4096;
    if (!c->h_ir.ensure(bytes)) return fail(c, AM355_E_NOMEM, "host allocation failed");
    uint8_t* p = c->h_ir.as<uint8_t>();
    am355_patch_ir& h = c->hir;
    auto pull = [&](const void* dev, size_t count, size_t elem) -> const void* {
      void* dst = p;
      p += carve_size(count, elem);
      if (count) (void)hipMemcpyAsync(dst, dev, count * elem, hipMemcpyDeviceToHost, st);
      return dst;
    };
    h.n_objects = NO; h.n_map = NM; h.n_edits = NR; h.n_values = NV;
    h.objects = (const am355_ir_object*)pull(c->ir.obj, NO, sizeof(am355_ir_object));
    h.map = (const am355_ir_map*)pull(c->ir.map, NM, sizeof(am355_ir_map));
    h.edits = with_edits ? (const am355_ir_edit*)pull(c->ir.edit, (size_t)NR + 1, sizeof(am355_ir_edit)) : nullptr;
    // (a ~0.1 ms copy: polled, not slept on -- a blocking wait adds an interrupt wake-up of tens of microseconds to a call of 140)
    {
      hipError_t q;
      while ((q = hipStreamQuery(st)) == hipErrorNotReady) {}
      if (q != hipSuccess) HIPCHK(c, q);
    }
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
    h.arena_len = c->raw.size();
    c->ir_fetched = with_edits;
  }
  if (out) *out = c->hir;
  return AM355_OK;
}

static int patch_json_impl(am355_ctx* c, const char** json, size_t* len) {
  if (!c) return AM355_E_ARG;
  int rc = fetch_ir_impl(c, nullptr);
  if (rc) return rc;
  std::string err;
  c->json.clear();
  if (!am355::render_patch_json(c->hir, c->json, err)) return fail(c, AM355_E_UNSUPPORTED, "%s", err.c_str());
  if (json) *json = c->json.c_str();
  if (len) *len = c->json.size();
  return AM355_OK;
}


// ---------------------------------------------------------------------------------------------------------
// Backend.applyChanges with its incremental patch (include/am355.h am355_apply_changes)
// ---------------------------------------------------------------------------------------------------------

// the device stage of am355_apply_changes over the replayed state of the context: rows >= T0 are the batch
static int run_delta_stage(am355_ctx* c, uint32_t T0, DeltaCounts* hc, bool check_only) {
  hipStream_t st = c->stream;
  const uint32_t N = (uint32_t)c->n_ops, NN = N - T0;
  const uint32_t NO = c->counts.n_objects, NM = c->counts.n_map_emit, NL = c->counts.n_list_ins;
  if (c->pass_first_row.size() > 4096) return fail(c, AM355_E_UNSUPPORTED, "more scheduling passes than the incremental patch stage handles");
  if (!c->d_delta.ensure(delta_bytes(N, NN, NM, NO, NL)) || !c->d_pass.ensure(4 * (c->pass_first_row.size() + 1))) return fail(c, AM355_E_NOMEM, "device allocation failed (delta)");
  DeltaBufs& d = c->delta;
  delta_bind(d, c->d_delta.p, N, NN, NM, NO, NL);
  canary_arm();
  d.T0 = T0; d.n_new = NN; d.n_obj = NO; d.n_map = NM; d.n_list = NL;
  d.bits_new = (uint32_t)bits_for64(NN ? NN - 1 : 0);
  std::vector<uint32_t> pass_rows;
  for (uint32_t r : c->pass_first_row) if (r > T0) pass_rows.push_back(r);
  d.n_pass = (uint32_t)pass_rows.size();
  d.pass_rows = c->d_pass.as<uint32_t>();
  // every row at which an op stream began: those of the earlier calls, this call's first row, its later passes
  std::vector<uint32_t> breaks;
  for (uint32_t r : c->stream_breaks) if (r < T0) breaks.push_back(r);
  if (T0) breaks.push_back(T0);
  breaks.insert(breaks.end(), pass_rows.begin(), pass_rows.end());
  if (!c->d_breaks.ensure(4 * (breaks.size() + 1))) return fail(c, AM355_E_NOMEM, "device allocation failed (delta)");
  d.n_breaks = (uint32_t)breaks.size();
  d.breaks = c->d_breaks.as<uint32_t>();
  d.breaks_exact = c->breaks_exact ? 1u : 0u;
  if (!breaks.empty()) HIPCHK(c, hipMemcpyAsync(c->d_breaks.p, breaks.data(), 4 * breaks.size(), hipMemcpyHostToDevice, st));
  if (d.n_pass) HIPCHK(c, hipMemcpyAsync(c->d_pass.p, pass_rows.data(), 4 * pass_rows.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipStreamSynchronize(st));  // (pageable sources)
  auto grow = [](void* user, size_t records) -> am355_ir_edit* {
    am355_ctx* cx = (am355_ctx*)user;
    return cx->d_delta_edit.ensure(sizeof(am355_ir_edit) * records) ? cx->d_delta_edit.as<am355_ir_edit>() : nullptr;
  };
  delta_run(c->mb, c->ir, d, hc, st, check_only, grow, c);
  HIPCHK(c, hipGetLastError());
  return AM355_OK;
}

static int apply_changes_impl(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n) {
  if (!c || (!arena && n) || !offsets) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  (void)hipSetDevice(c->device);
  c->apply_ready = false;
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "apply_changes: %-26s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  if (c->shard_world > 1) return fail(c, AM355_E_UNSUPPORTED, "am355_apply_changes on a sharded context");
  const bool have_state = c->staged;
  if (have_state && c->is_document) return fail(c, AM355_E_UNSUPPORTED, "the state was made by am355_load_document: applyChanges onto it is served by the JS path");
  if (have_state && !c->replayed) return fail(c, AM355_E_STATE, "the context holds no replayed state (the last replay failed?)");
  for (uint32_t i = 0; i < n; i++)
    if (offsets[i] > offsets[i + 1]) return fail(c, AM355_E_ARG, "change offsets must be ascending (offsets[%u] > offsets[%u])", i, i + 1);
  if (have_state && !c->state_checked) {
    // The state came from ONE am355_load_changes + am355_replay (Backend.loadChanges: one call of the reference). The patches of later
    // calls lean on objectMeta.children of the reference being the visible values of every property that holds a visible child
    // object (am355_apply.cpp) -- true unless the merge loop of that one call skipped values of such a property: checked now, with
    // every row of the state taken as the batch.
    DeltaCounts pre{};
    int prc = run_delta_stage(c, 0, &pre, true);
    if (prc) return prc;
    if (pre.hazard) c->children_hazard = true;  // (from here on no property is taken to list its visible values without asking)
    c->state_checked = true;
  }
  // ---- the queue of the call: changes applied so far (application order) | the batch | changes still queued (new.js:1822) ----
  const uint32_t n_old_applied = have_state ? (uint32_t)c->applied_change.size() : 0;
  const uint64_t old_ops = have_state ? c->n_ops : 0;
  // When every staged change was applied, in the order it is staged, and nothing is queued -- the usual case -- the staged bytes are
  // already that queue's front, in the pinned arena and in HBM: only the batch is gathered and copied behind them.
  bool append = have_state && c->pending_change.empty() && n_old_applied == c->n_changes && !getenv("AM355_APPLY_RESTAGE");
  for (uint32_t i = 0; append && i < n_old_applied; i++) append = c->applied_change[i] == i;
  int rc;
  if (append) {
    lap("queue = staged changes + batch");
    rc = load_changes_impl(c, arena, offsets, n, true);
  } else {
    std::vector<uint8_t> comb;
    std::vector<uint64_t> off;
    size_t bytes = (size_t)(offsets[n] - offsets[0]);
    if (have_state) bytes += c->raw.size();
    comb.reserve(bytes + 64);
    off.reserve((size_t)n_old_applied + n + c->pending_change.size() + 1);
    off.push_back(0);
    auto put_old = [&](uint32_t ci) {
      const uint8_t* p = c->raw.data() + c->raw_off[ci];
      comb.insert(comb.end(), p, p + (c->raw_off[ci + 1] - c->raw_off[ci]));
      off.push_back(comb.size());
    };
    if (have_state) for (uint32_t ci : c->applied_change) put_old(ci);
    for (uint32_t i = 0; i < n; i++) {
      comb.insert(comb.end(), arena + offsets[i], arena + offsets[i + 1]);
      off.push_back(comb.size());
    }
    if (have_state) for (uint32_t ci : c->pending_change) put_old(ci);
    lap("queue assembled");
    rc = load_changes_impl(c, comb.data(), off.data(), (uint32_t)off.size() - 1);
  }
  if (rc) { c->staged = false; return rc; }
  lap("staged");
  if (!have_state) { c->stream_breaks.clear(); c->breaks_exact = true; c->children_hazard = false; c->no_history = false; }
  c->in_apply = true;
  rc = replay_impl(c);
  c->in_apply = false;
  if (rc) { c->staged = false; return rc; }
  lap("replayed");
  // the earlier changes must have been applied again, first and in their order: rows [0, old_ops) are the state before the call
  bool prefix_ok = c->applied_change.size() >= n_old_applied && c->n_ops >= old_ops;
  for (uint32_t i = 0; prefix_ok && i < n_old_applied; i++) prefix_ok = c->applied_change[i] == i;
  if (prefix_ok && n_old_applied < c->applied_change.size()) prefix_ok = c->applied_op_base[n_old_applied] == old_ops;
  if (!prefix_ok) { c->staged = false; return fail(c, AM355_E_DEVICE, "internal: the earlier changes were not re-applied first"); }

  // ---- delta stage on the device ----
  hipStream_t st = c->stream;
  const uint32_t NO = c->counts.n_objects;
  DeltaBufs& d = c->delta;
  DeltaCounts hc{};
  // A call refused from here on leaves the context WITHOUT a state (include/am355.h): the replay above merged the batch, and a later
  // call must not get patches relative to a state that silently holds a batch whose call boundary nobody recorded.
  auto drop_state = [&](int code) { c->staged = c->replayed = c->ir_fetched = false; return code; };
  rc = run_delta_stage(c, (uint32_t)old_ops, &hc, false);
  if (rc) return drop_state(rc);
  lap("delta stage");
  c->state_checked = true;  // (a call the engine served: checked; a refused call leaves the state to the JS path)
  if (hc.flags) {
    c->state_checked = false;
    drop_state(0);
    if ((hc.flags & AM355_F_UNSUPPORTED) && hc.reason != NONE32) {
      c->flags |= hc.flags;
      return fail(c, AM355_E_UNSUPPORTED, "incremental patch not served: %s (JS path)", delta_reason_text(hc.reason));
    }
    return error_for_flags(c, hc.flags, "incremental patch not served");
  }

  // ---- tables to the host, setupPatches, assembly ----
  rc = fetch_ir_impl(c, nullptr, false);
  if (rc) return drop_state(rc);
  lap("document tables on the host");
  const uint32_t n_dmap = hc.n_kept + hc.n_place, n_dedits = hc.n_erecs;
  size_t b_link = carve_size(NO, sizeof(ObjLink)), b_map = carve_size(n_dmap, sizeof(am355_ir_map)), b_edit = carve_size((size_t)n_dedits + 1, sizeof(am355_ir_edit));
  if (!c->h_delta.ensure(b_link + b_map + b_edit + 256)) return fail(c, AM355_E_NOMEM, "host allocation failed");
  uint8_t* hp = c->h_delta.as<uint8_t>();
  ObjLink* h_link = (ObjLink*)hp;
  am355_ir_map* h_map = (am355_ir_map*)(hp + b_link);
  am355_ir_edit* h_edit = (am355_ir_edit*)(hp + b_link + b_map);
  HIPCHK(c, hipMemcpyAsync(h_link, d.link, sizeof(ObjLink) * (size_t)NO, hipMemcpyDeviceToHost, st));
  if (n_dmap) HIPCHK(c, hipMemcpyAsync(h_map, d.map, sizeof(am355_ir_map) * (size_t)n_dmap, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(h_edit, d.edit, sizeof(am355_ir_edit) * ((size_t)n_dedits + 1), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  std::string err;
  std::unordered_map<uint32_t, KeyHistory> known;
  std::vector<uint32_t> need;
  const bool ask_always = c->children_hazard;
  if (hc.hazard) c->children_hazard = true;  // (this call skipped values of a property with a child object: later calls ask)
  for (int round = 0;; round++) {
    rc = assemble_apply_patch(c->hir, h_link, h_map, n_dmap, h_edit, n_dedits, known, ask_always, need, c->apply, err);
    if (rc == AM355_E_UNSUPPORTED && need.empty() && !c->hir.edits && err == "edit records needed") {
      // a touched object hangs in a list: setupPatches needs the whole-document edit records of that list
      int frc = fetch_ir_impl(c, nullptr, true);
      if (frc) return drop_state(frc);
      lap("document edit records on the host");
      continue;
    }
    if (rc != AM355_E_UNSUPPORTED || need.empty() || round == 16 || need.size() > 256 || c->no_history) break;
    // the walk met objects that are no longer visible: what the reference's objectMeta lists for their property follows from the
    // history of the rows on it (am355_delta.hip, delta_key_history)
    std::vector<KeyHistory> st_of(need.size());
    if (delta_key_history(c->mb, c->ir, d, need.data(), (uint32_t)need.size(), st_of.data(), st) != 0) return drop_state(0), fail(c, AM355_E_DEVICE, "key history: %s", hipGetErrorString(hipGetLastError()));
    for (size_t i = 0; i < need.size(); i++) known[need[i]] = st_of[i];
    lap("property histories");
  }
  if (rc) { if (rc == AM355_E_UNSUPPORTED) c->flags |= AM355_F_UNSUPPORTED; drop_state(0); return fail(c, rc, "%s", err.c_str()); }
  // the op streams of this call (this engine re-applies the earlier changes in front of them: their rows keep their numbers)
  if (old_ops) c->stream_breaks.push_back((uint32_t)old_ops);
  for (uint32_t r : c->pass_first_row) if (r > old_ops) c->stream_breaks.push_back(r);
  c->apply_ready = true;
  c->apply_json.clear();
  lap("patch assembled");
  return AM355_OK;
}

extern "C" int am355_reset(am355_ctx* c) {
  if (!c) return AM355_E_ARG;
  (void)hipSetDevice(c->device);
  if (c->staging_in_flight) { c->staging_in_flight = false; (void)hipStreamSynchronize(c->stream); }
  c->staged = c->replayed = c->ir_fetched = c->apply_ready = false;
  c->state_checked = true;
  c->stream_breaks.clear();
  c->breaks_exact = true;
  c->children_hazard = false;
  c->no_history = false;
  c->is_document = false;
  c->flags = 0;
  c->n_changes = 0;
  c->applied_change.clear();
  c->pending_change.clear();
  return AM355_OK;
}

extern "C" int am355_forget_call_history(am355_ctx* c, int from_document) {
  if (!c) return AM355_E_ARG;
  c->breaks_exact = false;
  if (from_document) c->no_history = true;
  return AM355_OK;
}

extern "C" int am355_get_pending(const am355_ctx* c, uint32_t* out, uint32_t* n_pending) {
  if (!c || !n_pending) return AM355_E_ARG;
  if (!c->replayed || c->is_document) return AM355_E_STATE;
  *n_pending = (uint32_t)c->pending_change.size();
  if (out && !c->pending_change.empty()) memcpy(out, c->pending_change.data(), 4 * c->pending_change.size());
  return AM355_OK;
}

static int apply_patch_json_impl(am355_ctx* c, const char** json, size_t* len) {
  if (!c) return AM355_E_ARG;
  if (!c->apply_ready) return fail(c, AM355_E_STATE, "am355_apply_changes must succeed first");
  if (c->apply_json.empty()) {
    std::string err;
    if (!am355::render_patch_json(c->apply.ir, c->apply_json, err)) { c->apply_json.clear(); return fail(c, AM355_E_UNSUPPORTED, "%s", err.c_str()); }
  }
  if (json) *json = c->apply_json.c_str();
  if (len) *len = c->apply_json.size();
  return AM355_OK;
}

extern "C" int am355_fetch_apply_ir(am355_ctx* c, am355_patch_ir* out) {
  if (!c) return AM355_E_ARG;
  if (!c->apply_ready) return fail(c, AM355_E_STATE, "am355_apply_changes must succeed first");
  if (out) *out = c->apply.ir;
  return AM355_OK;
}


// ---------------------------------------------------------------------------------------------------------
// sync protocol, bulk side (include/am355.h)
// ---------------------------------------------------------------------------------------------------------
static int get_dep_graph_impl(am355_ctx* c, const uint32_t** dep_first, const uint32_t** dep_index, uint32_t* n_changes) {
  if (!c) return AM355_E_ARG;
  if (!c->replayed || c->is_document) return fail(c, AM355_E_STATE, "a replayed state of changes is needed");
  (void)hipSetDevice(c->device);
  if (!c->dep_graph_ready) {
    const uint32_t n = c->n_changes;
    const size_t dep_words = c->raw.size() / 32 + 2;
    // the device resolved every dependency hash to the index of the change that carries it (k_deps_resolve), addressed by the
    // dependency's place in the arena; the change headers say where those places are
    if (!c->h_dep_idx.ensure(4 * dep_words) || !c->h_metas.ensure(sizeof(ChangeMeta) * (size_t)std::max(n, 1u))) return fail(c, AM355_E_NOMEM, "host allocation failed");
    HIPCHK(c, hipMemcpyAsync(c->h_metas.p, c->d_metas.p, sizeof(ChangeMeta) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_dep_idx.p, c->d_dep_idx.p, 4 * dep_words, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const ChangeMeta* metas = c->h_metas.as<ChangeMeta>();
    const uint32_t* di = c->h_dep_idx.as<uint32_t>();
    c->dep_first.assign((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) c->dep_first[i + 1] = c->dep_first[i] + metas[i].n_deps;
    c->dep_index.resize(c->dep_first[n]);
    for (uint32_t i = 0; i < n; i++) {
      const size_t first = (size_t)((metas[i].base + metas[i].deps_off) >> 5);
      for (uint32_t k = 0; k < metas[i].n_deps; k++) c->dep_index[c->dep_first[i] + k] = di[first + k];
    }
    c->dep_graph_ready = true;
  }
  if (dep_first) *dep_first = c->dep_first.data();
  if (dep_index) *dep_index = c->dep_index.data();
  if (n_changes) *n_changes = c->n_changes;
  return AM355_OK;
}

static int sync_bloom_impl(am355_ctx* c, const uint32_t* idx, uint32_t n, uint32_t num_entries, uint32_t bits_per_entry, uint32_t num_probes, const uint8_t* probe_bits,
                           size_t probe_bytes, uint8_t* out, size_t out_cap, bool build) {
  if (!c || (n && !idx) || !out) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed || c->is_document || !c->d_hashes.p) return fail(c, AM355_E_STATE, "a replayed state of changes is needed");
  (void)hipSetDevice(c->device);
  for (uint32_t k = 0; k < n; k++)
    if (idx[k] >= c->n_changes) return fail(c, AM355_E_ARG, "change index %u out of range", idx[k]);
  hipStream_t st = c->stream;
  const uint64_t n_bits64 = build ? 8 * (((uint64_t)n * 10 + 7) / 8) : 8 * (uint64_t)probe_bytes;
  if (n_bits64 > 0xfffffff0ull) return fail(c, AM355_E_UNSUPPORTED, "Bloom filter beyond 2^32 bits");
  const uint32_t n_bits = (uint32_t)n_bits64;
  const size_t filter_bytes = n_bits / 8;
  if (build && out_cap < filter_bytes) return fail(c, AM355_E_ARG, "filter needs %zu bytes", filter_bytes);
  if (!build && (uint64_t)probe_bytes < ((uint64_t)num_entries * bits_per_entry + 7) / 8) return fail(c, AM355_E_ARG, "filter shorter than its header says");
  size_t o_bits = ((4 * (size_t)n + 255) & ~(size_t)255), o_flags = o_bits + ((filter_bytes + 8 + 255) & ~(size_t)255);
  if (!c->d_sync.ensure(o_flags + n + 256)) return fail(c, AM355_E_NOMEM, "device allocation failed");
  uint8_t* d = c->d_sync.as<uint8_t>();
  if (n) HIPCHK(c, hipMemcpyAsync(d, idx, 4 * (size_t)n, hipMemcpyHostToDevice, st));
  if (build) {
    launch_bloom_build(c->d_hashes.as<uint8_t>(), (const uint32_t*)d, n, (uint32_t*)(d + o_bits), n_bits, 7, st);
    if (filter_bytes) HIPCHK(c, hipMemcpyAsync(out, d + o_bits, filter_bytes, hipMemcpyDeviceToHost, st));
  } else {
    if (filter_bytes) HIPCHK(c, hipMemcpyAsync(d + o_bits, probe_bits, filter_bytes, hipMemcpyHostToDevice, st));
    // (an empty filter -- numEntries 0 -- contains nothing: sync.js:120)
    launch_bloom_probe(c->d_hashes.as<uint8_t>(), (const uint32_t*)d, n, d + o_bits, num_entries ? n_bits : 0, num_probes, d + o_flags, st);
    if (n) HIPCHK(c, hipMemcpyAsync(out, d + o_flags, n, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipStreamSynchronize(st));
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

// ---------------------------------------------------------------------------------------------------------
// Backend.save (new.js:2033-2055, columnar.js:983-1004)
// ---------------------------------------------------------------------------------------------------------
namespace {

struct HostOut : std::vector<uint8_t> {
  void uleb(uint64_t v) { while (v >= 0x80) { push_back((uint8_t)(v | 0x80)); v >>= 7; } push_back((uint8_t)v); }
  void sleb(int64_t v) {
    for (;;) {
      uint8_t b = (uint8_t)(v & 0x7f);
      v >>= 7;
      if ((v == 0 && !(b & 0x40)) || (v == -1 && (b & 0x40))) { push_back(b); return; }
      push_back(b | 0x80);
    }
  }
  void bytes(const void* p, size_t n) { insert(end(), (const uint8_t*)p, (const uint8_t*)p + n); }
};

// RLE of the (small) change-metadata columns: same output as the device encoders (am355_encode.hip), values never null
template <class T, class Put>
void host_rle(HostOut& out, const std::vector<T>& v, Put put) {
  size_t n = v.size(), i = 0;
  std::vector<size_t> lit;
  auto flush = [&]() {
    if (lit.empty()) return;
    out.sleb(-(int64_t)lit.size());
    for (size_t k : lit) put(out, v[k]);
    lit.clear();
  };
  while (i < n) {
    size_t j = i + 1;
    while (j < n && v[j] == v[i]) j++;
    if (j - i >= 2) { flush(); out.sleb((int64_t)(j - i)); put(out, v[i]); }
    else lit.push_back(i);
    i = j;
  }
  flush();
}
void host_rle_uint(HostOut& out, const std::vector<int64_t>& v) {
  host_rle(out, v, [](HostOut& o, int64_t x) { o.uleb((uint64_t)x); });
}
void host_delta(HostOut& out, const std::vector<int64_t>& v) {
  std::vector<int64_t> d(v.size());
  int64_t prev = 0;
  for (size_t i = 0; i < v.size(); i++) { d[i] = v[i] - prev; prev = v[i]; }
  host_rle(out, d, [](HostOut& o, int64_t x) { o.sleb(x); });
}
void host_rle_str(HostOut& out, const std::vector<std::string>& v) {
  host_rle(out, v, [](HostOut& o, const std::string& x) { o.uleb(x.size()); o.bytes(x.data(), x.size()); });
}

struct SaveColumn {
  uint32_t id;
  std::vector<uint8_t> data;
};

// columns of >= 256 bytes are stored DEFLATEd with bit 3 of the id set (columnar.js:1052-1057, DEFLATE_MIN_SIZE)
bool deflate_column(SaveColumn& col) {
  if (col.data.size() < 256) return true;
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  std::vector<uint8_t> out(deflateBound(&zs, (uLong)col.data.size()) + 64);
  zs.next_in = col.data.data(); zs.avail_in = (uInt)col.data.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
  int rc = deflate(&zs, Z_FINISH);
  size_t got = zs.total_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) return false;
  out.resize(got);
  col.data.swap(out);
  col.id |= 8;
  return true;
}

struct ChangeInfo {  // header fields of one applied change, parsed on the host (a few thousand small headers)
  std::vector<Hash32> deps;
  uint64_t seq = 0, start_op = 0;
  int64_t time = 0;
  std::string message, extra;
};

bool parse_change_info(const uint8_t* p, size_t len, ChangeInfo& ci) {
  size_t off = 9;
  uint64_t clen, v;
  if (len < 10 || !read_uleb_host(p, len, off, clen)) return false;
  if (!read_uleb_host(p, len, off, v) || v * 32 > len - off) return false;
  ci.deps.resize((size_t)v);
  for (auto& d : ci.deps) { memcpy(d.b, p + off, 32); off += 32; }
  if (!read_uleb_host(p, len, off, v) || v > len - off) return false;
  off += (size_t)v;  // actor
  if (!read_uleb_host(p, len, off, ci.seq) || !read_uleb_host(p, len, off, ci.start_op)) return false;
  {  // time: signed LEB128
    uint64_t u = 0;
    int shift = 0;
    for (;;) {
      if (off >= len || shift > 63) return false;
      uint8_t b = p[off++];
      u |= (uint64_t)(b & 0x7f) << shift;
      shift += 7;
      if (!(b & 0x80)) { if ((b & 0x40) && shift < 64) u |= ~0ull << shift; break; }
    }
    ci.time = (int64_t)u;
  }
  if (!read_uleb_host(p, len, off, v) || v > len - off) return false;
  ci.message.assign((const char*)p + off, (size_t)v);
  off += (size_t)v;
  if (!read_uleb_host(p, len, off, v)) return false;
  for (uint64_t k = 0; k < v; k++) {
    uint64_t l;
    if (!read_uleb_host(p, len, off, l) || l > len - off) return false;
    off += (size_t)l;
  }
  uint64_t ncols, total = 0;
  if (!read_uleb_host(p, len, off, ncols)) return false;
  for (uint64_t k = 0; k < ncols; k++) {
    uint64_t id, l;
    if (!read_uleb_host(p, len, off, id) || !read_uleb_host(p, len, off, l)) return false;
    total += l;
  }
  if (total > len - off) return false;
  off += (size_t)total;
  ci.extra.assign((const char*)p + off, len - off);  // extraBytes (columnar.js:757-760)
  return true;
}

__global__ __launch_bounds__(BLOCK) void k_save_identity(uint32_t n, uint32_t* __restrict__ v) {
  uint32_t i = gtid();
  if (i < n) v[i] = i;
}
// loaded document: rows are canonical already; only the actor fields change representation (rank -> document index)
__global__ __launch_bounds__(BLOCK) void k_save_doc_rows(OpCols in, uint32_t n, uint32_t n_succ, const uint32_t* __restrict__ doc_actor, OpCols out) {
  uint32_t f = gtid();
  if (f < n_succ) { out.pred_actor[f] = doc_actor[in.pred_actor[f]]; out.pred_ctr[f] = in.pred_ctr[f]; }
  if (f >= n) return;
  bool root = in.obj_actor[f] == NONE32;
  out.obj_actor[f] = root ? NONE32 : doc_actor[in.obj_actor[f]];
  out.obj_ctr[f] = root ? NONE32 : in.obj_ctr[f];
  out.key_actor[f] = in.key_actor[f] == NONE32 ? NONE32 : doc_actor[in.key_actor[f]];
  out.key_ctr[f] = in.key_ctr[f];
  out.key_off[f] = in.key_off[f];
  out.key_len[f] = in.key_len[f];
  out.id_actor[f] = doc_actor[in.id_actor[f]];
  out.id_ctr[f] = in.id_ctr[f];
  out.insert[f] = in.insert[f];
  out.action[f] = in.action[f];
  out.val_tl[f] = in.val_tl[f];
  out.val_off[f] = in.val_off[f];
  out.pred_num[f] = in.pred_num[f];
}

}  // namespace

static int save_impl(am355_ctx* c, uint32_t flags, const uint8_t** out_bytes, size_t* out_len) {
  if (!c || !out_bytes || !out_len) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must be called first");
  (void)hipSetDevice(c->device);
  if (c->is_document && !(flags & 1)) {  // unchanged document: the reference returns the bytes it was given (new.js:2034)
    *out_bytes = c->doc_bytes.data();
    *out_len = c->doc_bytes.size();
    return AM355_OK;
  }
  if (c->n_pending) return fail(c, AM355_E_UNSUPPORTED, "changes are queued: the document is saved by the JS path");
  if (!c->is_document && c->has_unknown_cols) return fail(c, AM355_E_UNSUPPORTED, "a change carries columns this engine does not model: the document is saved by the JS path");
  hipStream_t st = c->stream;
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "am355_save: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_start).count());
    t_start = now;
  };
  const uint32_t N = (uint32_t)c->n_ops, P = (uint32_t)c->n_preds, NA = (uint32_t)c->actors.size();
  const uint32_t n_obj = c->counts.n_objects, n_ins = c->counts.n_list_ins;
  // ---- actor table of the document: order of first appearance (new.js:1434-1441); documents keep theirs ----
  std::vector<uint32_t> doc_actor(std::max(NA, 1u), 0);  // rank -> document index
  std::vector<uint32_t> actor_by_doc(NA, 0);
  if (c->is_document) {
    for (uint32_t i = 0; i < NA; i++) { doc_actor[c->doc_actor_rank[i]] = i; actor_by_doc[i] = c->doc_actor_rank[i]; }
  } else {
    if (c->clock_actor.size() != NA) return fail(c, AM355_E_UNSUPPORTED, "actors without an applied change");
    for (uint32_t i = 0; i < NA; i++) { doc_actor[c->clock_actor[i]] = i; actor_by_doc[i] = c->clock_actor[i]; }
  }
  // ---- device buffers ----
  size_t n1 = (size_t)N + 2, p1 = (size_t)P + 2, o1 = (size_t)n_obj + 2;
  auto al = [](size_t b) { return carve_round(b); };
  size_t save_bytes = 10 * al(4 * n1) + 3 * al(4 * o1) + al(16 * o1) + 2 * al(8 * p1) + 2 * al(4 * p1) + al(64) + 13 * al(4 * n1) + al(n1) + 2 * al(4 * p1) + al(4 * (size_t)std::max(NA, 1u));
  if (!c->d_save.ensure(save_bytes)) return fail(c, AM355_E_NOMEM, "device allocation failed (save)");
  SaveBufs s;
  uint32_t* d_doc_actor;
  {
    uint8_t* p = c->d_save.as<uint8_t>();
    canary_scope("save buffers (d_save)");
    canary_forget(c->d_save.p, c->d_save.cap);
    auto take = [&](size_t bytes) { void* r = p; canary_note(p, bytes); p += al(bytes); return r; };
    uint32_t** a10[] = {&s.map_flag, &s.map_ex, &s.upd_flag, &s.upd_ex, &s.upd_cnt, &s.pos_of, &s.list_off, &s.final_pos, &s.src_of, &s.map_perm};
    for (uint32_t** a : a10) *a = (uint32_t*)take(4 * n1);
    s.obj_rank = (uint32_t*)take(4 * o1); s.rank_obj = (uint32_t*)take(4 * o1); s.base_by_rank = (uint32_t*)take(4 * o1);
    uint32_t* bounds = (uint32_t*)take(16 * o1);
    s.map_begin = bounds; s.map_end = bounds + (n_obj + 1); s.list_begin = bounds + 2 * (size_t)(n_obj + 1); s.list_end = bounds + 3 * (size_t)(n_obj + 1);
    s.succ_key_a = (uint64_t*)take(8 * p1); s.succ_key_b = (uint64_t*)take(8 * p1);
    s.succ_val_a = (uint32_t*)take(4 * p1); s.succ_val_b = (uint32_t*)take(4 * p1);
    s.words = (uint32_t*)take(64);
    OpCols& o = s.out;
    uint32_t** cols13[] = {&o.obj_actor, &o.obj_ctr, &o.key_actor, &o.key_ctr, &o.key_off, &o.key_len, &o.action, &o.val_tl, &o.val_off, &o.pred_first, &o.pred_num, &o.id_ctr, &o.id_actor};
    for (uint32_t** a : cols13) *a = (uint32_t*)take(4 * n1);
    o.insert = (uint8_t*)take(n1);
    o.pred_actor = (uint32_t*)take(4 * p1); o.pred_ctr = (uint32_t*)take(4 * p1);
    d_doc_actor = (uint32_t*)take(4 * (size_t)std::max(NA, 1u));
    canary_arm();
  }
  HIPCHK(c, hipMemcpyAsync(d_doc_actor, doc_actor.data(), 4 * (size_t)std::max(NA, 1u), hipMemcpyHostToDevice, st));
  // ---- rows in saved-document order ----
  uint32_t n_doc, n_succ = P;
  if (c->is_document) {
    n_doc = N;
    uint32_t n = std::max(N, P);
    if (n) AM355_LAUNCH_INDEPENDENT(k_save_doc_rows, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), st, c->cols, N, P, (const uint32_t*)d_doc_actor, s.out);
  } else {
    uint32_t words[8];
    save_phase1(c->mb, s, st);
    HIPCHK(c, hipMemcpyAsync(c->h_words.p, s.words, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    memcpy(words, c->h_words.p, 32);
    n_doc = words[0] + words[1] + n_ins;
    save_phase2(c->mb, c->ir, s, words, n_obj, n_ins, d_doc_actor, st);
  }
  // ---- encode the op columns on the device ----
  enum { E_OBJ_ACTOR, E_OBJ_CTR, E_KEY_ACTOR, E_KEY_CTR, E_KEY_STR, E_ID_ACTOR, E_ID_CTR, E_INSERT, E_ACTION, E_VAL_LEN, E_VAL_RAW, E_SUCC_NUM, E_SUCC_ACTOR, E_SUCC_CTR, E_NUM };
  static const uint32_t col_id[E_NUM] = {0x01, 0x02, 0x11, 0x13, 0x15, 0x21, 0x23, 0x34, 0x42, 0x56, 0x57, 0x80, 0x81, 0x83};
  uint32_t nmax = std::max(n_doc, n_succ);
  size_t raw_bound = c->raw.size() + 16;  // value bytes and key strings come out of the arena: never more than all of it
  size_t bound[E_NUM];
  for (int k = 0; k < E_NUM; k++) bound[k] = al(enc_numbers_bound(k >= E_SUCC_ACTOR ? n_succ : n_doc));
  bound[E_KEY_STR] = al(enc_numbers_bound(n_doc) + raw_bound);
  bound[E_VAL_RAW] = al(raw_bound);
  size_t out_total = 0;
  for (int k = 0; k < E_NUM; k++) out_total += bound[k];
  size_t enc_bytes = al(enc_work_bytes(nmax)) + al(4 * ((size_t)nmax + 2)) + al((size_t)nmax + 2) + al(4 * E_NUM);
  if (!c->d_enc.ensure(enc_bytes) || !c->d_encout.ensure(out_total + 256) || !c->h_words.ensure(256)) return fail(c, AM355_E_NOMEM, "device allocation failed (save columns)");
  EncWork w;
  canary_forget(c->d_enc.p, c->d_enc.cap);
  enc_carve(w, c->d_enc.p, nmax);
  canary_arm();
  uint32_t* deltas = (uint32_t*)(c->d_enc.as<uint8_t>() + al(enc_work_bytes(nmax)));
  uint8_t* nullmask = (uint8_t*)deltas + al(4 * ((size_t)nmax + 2));
  uint32_t* d_lens = (uint32_t*)(nullmask + al((size_t)nmax + 2));
  uint8_t* outp[E_NUM];
  {
    uint8_t* p = c->d_encout.as<uint8_t>();
    for (int k = 0; k < E_NUM; k++) { outp[k] = p; p += bound[k]; }
  }
  const OpCols& o = s.out;
  const uint8_t* arena = c->d_arena.as<uint8_t>();
  enc_rle_numbers(o.obj_actor, nullptr, n_doc, false, w, outp[E_OBJ_ACTOR], d_lens + E_OBJ_ACTOR, st);
  enc_rle_numbers(o.obj_ctr, nullptr, n_doc, false, w, outp[E_OBJ_CTR], d_lens + E_OBJ_CTR, st);
  enc_rle_numbers(o.key_actor, nullptr, n_doc, false, w, outp[E_KEY_ACTOR], d_lens + E_KEY_ACTOR, st);
  enc_delta_prepare(o.key_ctr, n_doc, deltas, nullmask, w, st);
  enc_rle_numbers(deltas, nullmask, n_doc, true, w, outp[E_KEY_CTR], d_lens + E_KEY_CTR, st);
  enc_rle_strings(arena, o.key_off, o.key_len, n_doc, w, outp[E_KEY_STR], d_lens + E_KEY_STR, st);
  enc_rle_numbers(o.id_actor, nullptr, n_doc, false, w, outp[E_ID_ACTOR], d_lens + E_ID_ACTOR, st);
  enc_delta_prepare(o.id_ctr, n_doc, deltas, nullmask, w, st);
  enc_rle_numbers(deltas, nullmask, n_doc, true, w, outp[E_ID_CTR], d_lens + E_ID_CTR, st);
  enc_boolean(o.insert, n_doc, w, outp[E_INSERT], d_lens + E_INSERT, st);
  enc_rle_numbers(o.action, nullptr, n_doc, false, w, outp[E_ACTION], d_lens + E_ACTION, st);
  enc_rle_numbers(o.val_tl, nullptr, n_doc, false, w, outp[E_VAL_LEN], d_lens + E_VAL_LEN, st);
  enc_raw_values(arena, o.val_off, o.val_tl, n_doc, w, outp[E_VAL_RAW], d_lens + E_VAL_RAW, st);
  enc_rle_numbers(o.pred_num, nullptr, n_doc, false, w, outp[E_SUCC_NUM], d_lens + E_SUCC_NUM, st);
  enc_rle_numbers(o.pred_actor, nullptr, n_succ, false, w, outp[E_SUCC_ACTOR], d_lens + E_SUCC_ACTOR, st);
  enc_delta_prepare(o.pred_ctr, n_succ, deltas, nullmask, w, st);
  enc_rle_numbers(deltas, nullmask, n_succ, true, w, outp[E_SUCC_CTR], d_lens + E_SUCC_CTR, st);
  HIPCHK(c, hipMemcpyAsync(c->h_words.p, d_lens, 4 * E_NUM, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  uint32_t lens[E_NUM];
  memcpy(lens, c->h_words.p, 4 * E_NUM);
  lap("row order + column encode");
  std::vector<SaveColumn> ops_cols;
  {
    size_t total = 0;
    for (int k = 0; k < E_NUM; k++) { if (lens[k] > bound[k]) return fail(c, AM355_E_UNSUPPORTED, "encoded column larger than its bound"); total += lens[k]; }
    if (!c->h_encout.ensure(total + 16)) return fail(c, AM355_E_NOMEM, "host allocation failed (save)");
    uint8_t* h = c->h_encout.as<uint8_t>();
    size_t at = 0;
    for (int k = 0; k < E_NUM; k++) {
      if (lens[k]) HIPCHK(c, hipMemcpyAsync(h + at, outp[k], lens[k], hipMemcpyDeviceToHost, st));
      at += lens[k];
    }
    HIPCHK(c, hipStreamSynchronize(st));
    at = 0;
    for (int k = 0; k < E_NUM; k++) {
      ops_cols.push_back(SaveColumn{col_id[k], std::vector<uint8_t>(h + at, h + at + lens[k])});
      at += lens[k];
    }
  }
  lap("columns to host");
  {
    // A key that starts with U+FEFF: the reference decodes keys on their way into a document (TextDecoder drops a leading byte
    // order mark, encoding.js:9-17) and writes the shortened key; such documents are saved by the JS path. (Walk of the encoded key
    // column: one step per run or literal.)
    const std::vector<uint8_t>& kc = ops_cols[E_KEY_STR].data;
    size_t o = 0;
    auto uleb = [&](uint64_t& v) { return read_uleb_host(kc.data(), kc.size(), o, v); };
    auto sleb = [&](int64_t& v) {
      uint64_t u = 0; int shift = 0;
      while (o < kc.size() && shift < 64) {
        uint8_t b = kc[o++];
        u |= (uint64_t)(b & 0x7f) << shift; shift += 7;
        if (!(b & 0x80)) { if ((b & 0x40) && shift < 64) u |= ~0ull << shift; v = (int64_t)u; return true; }
      }
      return false;
    };
    bool bom = false, ok = true;
    while (ok && o < kc.size() && !bom) {
      int64_t n;
      if (!(ok = sleb(n))) break;
      uint64_t strings = n > 0 ? 1 : n < 0 ? (uint64_t)-n : 0, l;
      if (n == 0) { ok = uleb(l); continue; }
      for (uint64_t k = 0; ok && k < strings; k++) {
        ok = uleb(l) && l <= kc.size() - o;
        if (ok) { bom = bom || (l >= 3 && kc[o] == 0xef && kc[o + 1] == 0xbb && kc[o + 2] == 0xbf); o += (size_t)l; }
      }
    }
    if (!ok) return fail(c, AM355_E_DEVICE, "internal: encoded key column does not parse");
    if (bom) return fail(c, AM355_E_UNSUPPORTED, "a map key starts with a byte order mark: the document is saved by the JS path");
  }
  // ---- change metadata columns (columnar.js:86-96, new.js:1680-1692) ----
  std::vector<SaveColumn> chg_cols;
  std::vector<uint8_t> tail;  // headsIndexes (+ extraBytes of a loaded document)
  if (c->is_document) {
    // diagnostic re-encode of a loaded document (flags & 1): its change metadata and trailer are kept as loaded
    if (c->doc_other_ops_cols) return fail(c, AM355_E_UNSUPPORTED, "document has op columns this engine does not model");
    for (auto& col : c->doc_chg_cols) chg_cols.push_back(SaveColumn{col.first, col.second});
    tail = c->doc_tail;
  } else {
    size_t na = c->applied_change.size();
    std::vector<int64_t> v_actor(na), v_seq(na), v_maxop(na), v_time(na), v_depsnum(na), v_depsidx, v_extralen(na);
    std::vector<std::string> v_msg(na);
    std::string extra_raw;
    std::unordered_map<Hash32, uint32_t, Hash32Hasher> index_of;
    const uint8_t* hs = c->h_hashes.as<uint8_t>();
    for (size_t i = 0; i < na; i++) {
      Hash32 h;
      memcpy(h.b, hs + 32 * (size_t)c->applied_change[i], 32);
      index_of.emplace(h, (uint32_t)i);
    }
    // author of each applied change: rank of its actor id
    std::unordered_map<std::string, uint32_t> rank_of;
    for (uint32_t r = 0; r < NA; r++) rank_of[c->actors[r]] = r;
    for (size_t i = 0; i < na; i++) {
      uint32_t ci = c->applied_change[i];
      const uint8_t* p = c->raw.data() + c->raw_off[ci];
      size_t len = (size_t)(c->raw_off[ci + 1] - c->raw_off[ci]);
      ChangeInfo info;
      if (!parse_change_info(p, len, info)) return fail(c, AM355_E_INVALID, "change %u: malformed header", ci);
      // actor id bytes: after the deps
      size_t off = 9;
      uint64_t clen, nd, al2;
      read_uleb_host(p, len, off, clen);
      read_uleb_host(p, len, off, nd);
      off += (size_t)nd * 32;
      read_uleb_host(p, len, off, al2);
      auto it = rank_of.find(std::string((const char*)p + off, (size_t)al2));
      if (it == rank_of.end()) return fail(c, AM355_E_STATE, "change %u: unknown author", ci);
      uint32_t n_ops_i = (i + 1 < na ? c->applied_op_base[i + 1] : N) - c->applied_op_base[i];
      v_actor[i] = doc_actor[it->second];
      v_seq[i] = (int64_t)info.seq;
      v_maxop[i] = (int64_t)(info.start_op + n_ops_i) - 1;
      v_time[i] = info.time;
      v_msg[i] = info.message;
      v_depsnum[i] = (int64_t)info.deps.size();
      for (const Hash32& d : info.deps) {
        auto di = index_of.find(d);
        if (di == index_of.end()) return fail(c, AM355_E_STATE, "change %u: dependency is not an applied change", ci);
        v_depsidx.push_back(di->second);
      }
      v_extralen[i] = (int64_t)(info.extra.size() << 4 | 7);  // VALUE_TYPE.BYTES
      extra_raw += info.extra;
    }
    HostOut a, sq, mo, tm, ms, dn, dx, el;
    host_rle_uint(a, v_actor); host_delta(sq, v_seq); host_delta(mo, v_maxop); host_delta(tm, v_time); host_rle_str(ms, v_msg);
    host_rle_uint(dn, v_depsnum); host_delta(dx, v_depsidx); host_rle_uint(el, v_extralen);
    chg_cols.push_back(SaveColumn{0x01, a}); chg_cols.push_back(SaveColumn{0x03, sq}); chg_cols.push_back(SaveColumn{0x13, mo});
    chg_cols.push_back(SaveColumn{0x23, tm}); chg_cols.push_back(SaveColumn{0x35, ms}); chg_cols.push_back(SaveColumn{0x40, dn});
    chg_cols.push_back(SaveColumn{0x43, dx}); chg_cols.push_back(SaveColumn{0x56, el});
    chg_cols.push_back(SaveColumn{0x57, std::vector<uint8_t>(extra_raw.begin(), extra_raw.end())});
    HostOut t;
    for (size_t k = 0; k + 32 <= c->heads.size(); k += 32) {
      Hash32 h;
      memcpy(h.b, &c->heads[k], 32);
      auto hi = index_of.find(h);
      if (hi == index_of.end()) return fail(c, AM355_E_STATE, "head is not an applied change");
      t.uleb(hi->second);
    }
    tail = t;
  }
  lap("change metadata");
  // ---- document chunk (columnar.js:983-1004): actors, heads, the two column directories, column data, head indexes ----
  {
    // DEFLATE is the one sequential codec the format imposes; columns are independent streams, so the big ones get a host
    // thread each (output per column is unchanged)
    std::vector<SaveColumn*> all;
    for (auto& col : chg_cols) all.push_back(&col);
    for (auto& col : ops_cols) all.push_back(&col);
    std::vector<std::thread> workers;
    std::vector<int> ok(all.size(), 1);
    for (size_t k = 0; k < all.size(); k++) {
      if (all[k]->data.size() >= (64u << 10)) workers.emplace_back([&, k]() { ok[k] = deflate_column(*all[k]) ? 1 : 0; });
      else ok[k] = deflate_column(*all[k]) ? 1 : 0;
    }
    for (auto& t : workers) t.join();
    for (int v : ok)
      if (!v) return fail(c, AM355_E_NOMEM, "deflate failed");
  }
  lap("deflate");
  HostOut body;
  body.uleb(NA);
  for (uint32_t i = 0; i < NA; i++) { const std::string& id = c->actors[actor_by_doc[i]]; body.uleb(id.size()); body.bytes(id.data(), id.size()); }
  body.uleb(c->heads.size() / 32);
  body.bytes(c->heads.data(), c->heads.size());
  auto directory = [&](const std::vector<SaveColumn>& cols) {
    size_t n = 0;
    for (auto& col : cols) n += col.data.empty() ? 0 : 1;
    body.uleb(n);
    for (auto& col : cols)
      if (!col.data.empty()) { body.uleb(col.id); body.uleb(col.data.size()); }
  };
  directory(chg_cols);
  directory(ops_cols);
  for (auto& col : chg_cols) body.bytes(col.data.data(), col.data.size());
  for (auto& col : ops_cols) body.bytes(col.data.data(), col.data.size());
  body.bytes(tail.data(), tail.size());
  HostOut chunk;  // [type][LEB len][body]: the part the checksum covers (columnar.js:664-686)
  chunk.push_back(0);
  chunk.uleb(body.size());
  chunk.bytes(body.data(), body.size());
  uint8_t digest[32];
  sha256_digest(chunk.data(), chunk.size(), digest);
  c->saved.clear();
  static const uint8_t magic[4] = {0x85, 0x6f, 0x4a, 0x83};
  c->saved.insert(c->saved.end(), magic, magic + 4);
  c->saved.insert(c->saved.end(), digest, digest + 4);
  c->saved.insert(c->saved.end(), chunk.begin(), chunk.end());
  lap("assembly + checksum");
  *out_bytes = c->saved.data();
  *out_len = c->saved.size();
  return AM355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// objectId sharding across GPUs. Every rank stages and decodes the whole batch (rows keep their global
// indexes, so op id -> row stays arithmetic) and merges only the objects it owns; the object table is identical on all
// ranks. What is exchanged is the OUTPUT: each rank's record tables (map records / edit records / values of its objects)
// as one contiguous fragment, all-gathered by the host binding over RCCL (xGMI) and stitched by object index.
// ---------------------------------------------------------------------------------------------------------
namespace {
struct FragmentHeader {
  uint32_t magic, world, rank, n_objects, n_map, n_erecs, n_values, reserved;
  uint64_t off_objects, off_map, off_edits, total;
};
constexpr uint32_t FRAGMENT_MAGIC = 0x46333535;  // "553F"
inline size_t frag_align(size_t x) { return (x + 255) & ~(size_t)255; }
FragmentHeader fragment_layout(uint32_t world, uint32_t rank, const Counts& k) {
  FragmentHeader h{};
  h.magic = FRAGMENT_MAGIC; h.world = world; h.rank = rank;
  h.n_objects = k.n_objects; h.n_map = k.n_map_emit; h.n_erecs = k.n_erecs; h.n_values = k.n_edits;
  h.off_objects = frag_align(sizeof(FragmentHeader));
  h.off_map = h.off_objects + frag_align((size_t)h.n_objects * sizeof(am355_ir_object));
  h.off_edits = h.off_map + frag_align((size_t)h.n_map * sizeof(am355_ir_map));
  h.total = h.off_edits + frag_align(((size_t)h.n_erecs + 1) * sizeof(am355_ir_edit));
  return h;
}
}  // namespace

extern "C" int am355_set_shard(am355_ctx* c, uint32_t rank, uint32_t world) {
  if (!c || world == 0 || rank >= world) return c ? fail(c, AM355_E_ARG, "bad shard (rank %u of %u)", rank, world) : AM355_E_ARG;
  c->shard_rank = rank;
  c->shard_world = world;
  c->replayed = c->ir_fetched = false;
  return AM355_OK;
}

extern "C" int am355_fragment_size(am355_ctx* c, size_t* bytes) {
  if (!c || !bytes) return AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  *bytes = (size_t)fragment_layout(c->shard_world, c->shard_rank, c->counts).total;
  return AM355_OK;
}

extern "C" int am355_export_fragment(am355_ctx* c, void* dst, size_t cap, int dst_is_device, size_t* len) {
  if (!c || !dst || !len) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  (void)hipSetDevice(c->device);
  FragmentHeader h = fragment_layout(c->shard_world, c->shard_rank, c->counts);
  if (h.total > cap) return fail(c, AM355_E_ARG, "fragment needs %llu bytes, buffer has %llu", (unsigned long long)h.total, (unsigned long long)cap);
  hipStream_t st = c->stream;
  uint8_t* d = (uint8_t*)dst;
  hipMemcpyKind from_dev = dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (!c->h_words.ensure(sizeof(FragmentHeader) + 64)) return fail(c, AM355_E_NOMEM, "host allocation failed");
  FragmentHeader* hh = (FragmentHeader*)(c->h_words.as<uint8_t>() + 64);  // (pinned; the first 64 bytes are the replay's words)
  *hh = h;
  HIPCHK(c, hipMemcpyAsync(d, hh, sizeof h, dst_is_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost, st));
  if (h.n_objects) HIPCHK(c, hipMemcpyAsync(d + h.off_objects, c->ir.obj, (size_t)h.n_objects * sizeof(am355_ir_object), from_dev, st));
  if (h.n_map) HIPCHK(c, hipMemcpyAsync(d + h.off_map, c->ir.map, (size_t)h.n_map * sizeof(am355_ir_map), from_dev, st));
  HIPCHK(c, hipMemcpyAsync(d + h.off_edits, c->ir.edit, ((size_t)h.n_erecs + 1) * sizeof(am355_ir_edit), from_dev, st));
  HIPCHK(c, hipStreamSynchronize(st));
  *len = (size_t)h.total;
  return AM355_OK;
}

// frags: `world` fragments back to back, fragment r = frags[offsets[r] .. offsets[r+1]) (host memory). Builds the patch IR of
// the whole document in this context (the one whose am355_patch_json / am355_fetch_ir the caller then uses: it must have
// replayed the same batch, its envelope and arena serve the stitched patch).
static int import_fragments_impl(am355_ctx* c, const uint8_t* frags, const uint64_t* offsets, uint32_t world) {
  if (!c || !frags || !offsets || !world) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  std::vector<FragmentHeader> hs(world);
  uint64_t n_map = 0, n_erecs = 0, n_values = 0;
  for (uint32_t r = 0; r < world; r++) {
    if (offsets[r + 1] < offsets[r] || offsets[r + 1] - offsets[r] < sizeof(FragmentHeader)) return fail(c, AM355_E_ARG, "fragment %u too short", r);
    memcpy(&hs[r], frags + offsets[r], sizeof(FragmentHeader));
    const FragmentHeader& h = hs[r];
    Counts k{};
    k.n_objects = h.n_objects; k.n_map_emit = h.n_map; k.n_erecs = h.n_erecs; k.n_edits = h.n_values;
    FragmentHeader want = fragment_layout(world, r, k);
    if (h.magic != FRAGMENT_MAGIC || h.world != world || h.rank != r || h.total != want.total || h.total > offsets[r + 1] - offsets[r] ||
        h.n_objects != hs[0].n_objects)
      return fail(c, AM355_E_ARG, "fragment %u is malformed or from another batch", r);
    n_map += h.n_map; n_erecs += (uint64_t)h.n_erecs + 1; n_values += h.n_values;
  }
  if (n_map >= 0xfffffff0ull || n_erecs >= 0xfffffff0ull || n_values >= 0xfffffff0ull) return fail(c, AM355_E_UNSUPPORTED, "stitched patch too large");
  const uint32_t NO = hs[0].n_objects;
  size_t o_obj = 0, o_map = o_obj + frag_align((size_t)NO * sizeof(am355_ir_object)), o_edit = o_map + frag_align(n_map * sizeof(am355_ir_map)),
         total = o_edit + frag_align(n_erecs * sizeof(am355_ir_edit));
  c->stitched.assign(total + 64, 0);
  uint8_t* base = c->stitched.data();
  base += (64 - ((uintptr_t)base & 63)) & 63;
  am355_ir_object* obj = (am355_ir_object*)(base + o_obj);
  am355_ir_map* map = (am355_ir_map*)(base + o_map);
  am355_ir_edit* edit = (am355_ir_edit*)(base + o_edit);
  std::vector<uint32_t> map_base(world), edit_base(world), val_base(world);
  uint32_t mb = 0, eb = 0, vb = 0;
  for (uint32_t r = 0; r < world; r++) {
    const FragmentHeader& h = hs[r];
    const uint8_t* f = frags + offsets[r];
    map_base[r] = mb; edit_base[r] = eb; val_base[r] = vb;
    if (h.n_map) memcpy(map + mb, f + h.off_map, (size_t)h.n_map * sizeof(am355_ir_map));
    memcpy(edit + eb, f + h.off_edits, ((size_t)h.n_erecs + 1) * sizeof(am355_ir_edit));
    for (uint32_t k = 0; k <= h.n_erecs; k++) edit[eb + k].first += vb;  // (the sentinel of rank r then points at rank r+1's first value)
    mb += h.n_map; eb += h.n_erecs + 1; vb += h.n_values;
  }
  // object table: identical on every rank but for the ranges, which the owner knows
  const am355_ir_object* obj0 = (const am355_ir_object*)(frags + offsets[0] + hs[0].off_objects);
  for (uint32_t oi = 0; oi < NO; oi++) {
    uint32_t owner = oi == 0 ? 0u : shard_owner(obj0[oi].id_actor, obj0[oi].id_ctr, world);
    const am355_ir_object& src = ((const am355_ir_object*)(frags + offsets[owner] + hs[owner].off_objects))[oi];
    if (src.id_ctr != obj0[oi].id_ctr || src.id_actor != obj0[oi].id_actor || src.map_end > hs[owner].n_map || src.edit_end > hs[owner].n_erecs)
      return fail(c, AM355_E_ARG, "fragments disagree on object %u", oi);
    obj[oi] = src;
    obj[oi].map_begin += map_base[owner]; obj[oi].map_end += map_base[owner];
    obj[oi].edit_begin += edit_base[owner]; obj[oi].edit_end += edit_base[owner];
  }
  int rc = fetch_ir_impl(c, nullptr);  // envelope (clock, heads, actors, arena) of this context
  if (rc) return rc;
  am355_patch_ir& h = c->hir;
  h.n_objects = NO; h.n_map = mb; h.n_edits = eb; h.n_values = vb;
  h.objects = obj; h.map = map; h.edits = edit;
  // (edit record eb is never read: the last fragment's own sentinel is record eb - 1)
  return AM355_OK;
}

// ---- C ABI entry points of the calls that allocate with the input size ----
// ---------------------------------------------------------------------------------------------------------
// history of a loaded document (am355_history.cpp does the host work; the rows come from the device decode)
// ---------------------------------------------------------------------------------------------------------
static int doc_changes_impl(am355_ctx* c, uint32_t flags, const uint8_t** arena, const uint64_t** offsets, uint32_t* n_changes, const uint8_t** hashes) {
  if (!c || !arena || !offsets || !n_changes || !hashes) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed || !c->is_document) return fail(c, AM355_E_STATE, "am355_load_document and am355_replay must be called first");
  (void)hipSetDevice(c->device);
  if (!(c->history_ok && c->history_flags == (flags & 1))) {
    if (c->doc_other_ops_cols) return fail(c, AM355_E_UNSUPPORTED, "document has op columns this engine does not model (child / link / unknown): history comes from the JS path");
    if (c->doc_col_rows.size() != BIG_NCOL) return fail(c, AM355_E_UNSUPPORTED, "history needs the parallel column decode (AM355_DOC_SERIAL is set)");
    const bool trace = getenv("AM355_TRACE") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!trace) return;
      auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "am355_doc_changes: %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t0).count());
      t0 = now;
    };
    const uint32_t N = (uint32_t)c->n_ops, P = (uint32_t)c->n_preds;
    hipStream_t st = c->stream;
    HistoryInput in;
    in.n_rows = N; in.n_succ = P;
    in.actors = &c->actors;
    in.change_columns = &c->doc_chg_cols;
    in.doc_actor_rank = &c->doc_actor_rank;
    in.heads = c->heads.data(); in.n_heads = (uint32_t)(c->heads.size() / 32);
    {
      // the reference reads rows until EVERY column is exhausted (columnar.js:577-590 decodeColumns): a column holding more values
      // than the action column makes extra, empty rows there. The per-row columns must hold N values or none.
      static const int per_row[] = {BC_OBJ_ACTOR, BC_OBJ_CTR, BC_KEY_ACTOR, BC_KEY_CTR, BC_ID_ACTOR, BC_ID_CTR, BC_INSERT, BC_ACTION, BC_VAL_LEN, BC_SUCC_NUM};
      for (int k : per_row)
        if (c->doc_col_rows[k] != N && c->doc_col_rows[k] != 0) return fail(c, AM355_E_UNSUPPORTED, "op columns of unequal length: the JS path decides");
      if (c->doc_col_rows[BC_SUCC_ACTOR] != P || c->doc_col_rows[BC_SUCC_CTR] != P) return fail(c, AM355_E_UNSUPPORTED, "succ columns do not match succNum: the JS path decides");
      in.key_column = c->raw.data() + c->doc_meta.col_off[C_KEY_STR];
      in.key_column_len = c->doc_meta.col_len[C_KEY_STR];
      in.val_raw_len = c->doc_meta.col_len[C_VAL_RAW];
    }
    // ---- host: the change metadata columns (a few thousand values) ----
    std::string err;
    HistoryMeta meta;
    int rc = history_metadata(in, meta, err);
    if (rc == HISTORY_INVALID) return fail(c, AM355_E_INVALID, "%s", err.c_str());
    if (rc) return fail(c, AM355_E_UNSUPPORTED, "%s", err.c_str());
    lap("change metadata");
    // ---- device, stage 1: ids -> slots, preds by slot, the changes' slot ranges (am355_hist.hip) ----
    const uint32_t NC = (uint32_t)meta.chg.size(), NA = (uint32_t)c->actors.size(), W = meta.word_base[NA];
    const size_t key_bytes = c->doc_meta.col_len[C_KEY_STR], val_bytes = c->doc_meta.col_len[C_VAL_RAW];
    if (!c->d_hist.ensure(hist_bytes(N, P, NC, NA, W, key_bytes, val_bytes))) return fail(c, AM355_E_NOMEM, "device allocation failed (history)");
    HistBufs hb;
    hist_bind(hb, c->d_hist.p, N, P, NC, NA, W, key_bytes, val_bytes);
    canary_arm();
    const size_t AW = hb.AW, c1 = (size_t)NC + 1;
    // pinned staging: [word_base | act_max | chg_actor | chg_prev_max | chg_max] up, [flags | chg_base | chg_nops] down, then
    // [sorted_base | sorted_chg] up and [flags | col_len | col_off | abits | column bytes] down
    size_t col_total_cap = 0;
    for (int k = 0; k < HIST_NCOL; k++) col_total_cap += hb.col_cap[k] + 256;
    const size_t up_words = 2 * ((size_t)NA + 1) + 5 * c1, down_words = 8 + 2 * c1 + HIST_NCOL + (size_t)HIST_NCOL * 2 * (c1) + c1 * AW;
    if (!c->h_rows.ensure(4 * (up_words + down_words) + col_total_cap + 4096)) return fail(c, AM355_E_NOMEM, "host allocation failed (history)");
    uint32_t* up = c->h_rows.as<uint32_t>();
    uint32_t *u_word_base = up, *u_act_max = up + NA + 1, *u_actor = u_act_max + NA + 1, *u_prev = u_actor + c1, *u_max = u_prev + c1, *u_sbase = u_max + c1, *u_schg = u_sbase + c1;
    uint32_t* down = up + up_words;
    uint32_t *d_flags = down, *d_base = down + 8, *d_nops = d_base + c1, *d_col_len = d_nops + c1, *d_col_off = d_col_len + HIST_NCOL, *d_abits = d_col_off + (size_t)HIST_NCOL * 2 * c1;
    uint8_t* d_cols = (uint8_t*)(down + down_words);
    memcpy(u_word_base, meta.word_base.data(), 4 * ((size_t)NA + 1));
    if (NA) memcpy(u_act_max, meta.act_max.data(), 4 * (size_t)NA);
    for (uint32_t k = 0; k < NC; k++) {
      const HistoryChange& ch = meta.chg[k];
      u_actor[k] = ch.actor;
      u_prev[k] = ch.prev_same_actor == NONE32 ? 0u : (uint32_t)meta.chg[ch.prev_same_actor].max_op;
      u_max[k] = (uint32_t)ch.max_op;
    }
    HIPCHK(c, hipMemcpyAsync(hb.word_base, u_word_base, 4 * ((size_t)NA + 1), hipMemcpyHostToDevice, st));
    if (NA) HIPCHK(c, hipMemcpyAsync(hb.act_max, u_act_max, 4 * (size_t)NA, hipMemcpyHostToDevice, st));
    if (NC) {
      HIPCHK(c, hipMemcpyAsync(hb.chg_actor, u_actor, 4 * (size_t)NC, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(hb.chg_prev_max, u_prev, 4 * (size_t)NC, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(hb.chg_max, u_max, 4 * (size_t)NC, hipMemcpyHostToDevice, st));
    }
    hist_stage1(c->cols, hb, st);
    HIPCHK(c, hipMemcpyAsync(d_flags, hb.flags, 16, hipMemcpyDeviceToHost, st));
    if (NC) {
      HIPCHK(c, hipMemcpyAsync(d_base, hb.chg_base, 4 * (size_t)NC, hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipMemcpyAsync(d_nops, hb.chg_nops, 4 * (size_t)NC, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    lap("ids -> slots, preds (device)");
    if (d_flags[0] & HF_INVALID) return fail(c, AM355_E_INVALID, "operation ids of the document contradict its change metadata");
    const uint32_t M = d_flags[2], PT = d_flags[3];
    // the changes that own slots, in slot order = (actor, seq) order; every slot must belong to one of them
    uint32_t n_sorted = 0;
    {
      std::vector<uint32_t> order;
      order.reserve(NC);
      for (uint32_t k = 0; k < NC; k++) if (d_nops[k]) order.push_back(k);
      std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return d_base[x] < d_base[y]; });
      uint64_t covered = 0;
      for (uint32_t k : order) {
        if (d_base[k] != covered) return fail(c, AM355_E_INVALID, "operation ids that no change of the document accounts for");
        covered += d_nops[k];
        u_sbase[n_sorted] = d_base[k];
        u_schg[n_sorted++] = k;
      }
      if (covered != M) return fail(c, AM355_E_INVALID, "operation ids that no change of the document accounts for");
    }
    if (n_sorted) {
      HIPCHK(c, hipMemcpyAsync(hb.sorted_base, u_sbase, 4 * (size_t)n_sorted, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(hb.sorted_chg, u_schg, 4 * (size_t)n_sorted, hipMemcpyHostToDevice, st));
    }
    // ---- device, stage 2: actor tables, the changes' op columns, the twelve column encodes segmented by change ----
    c->pool->prewake(c->pool->size(), 4000);   // (the host threads assemble and hash right behind it: they poll instead of sleeping until then)
    hb.P = PT;   // (pred entries = succ entries the slots account for)
    hist_stage2(c->cols, c->d_arena.as<uint8_t>(), c->raw.size(), hb, n_sorted, M, st);
    HIPCHK(c, hipMemcpyAsync(d_flags, hb.flags, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(d_col_len, hb.col_len, 4 * HIST_NCOL, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(d_col_off, hb.col_off, 4 * (size_t)HIST_NCOL * 2 * c1, hipMemcpyDeviceToHost, st));
    if (NC) HIPCHK(c, hipMemcpyAsync(d_abits, hb.abits, 4 * (size_t)NC * AW, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (d_flags[0] & HF_INVALID) return fail(c, AM355_E_INVALID, "the document's rows do not re-encode (an operation the reference throws on)");
    if (d_flags[0] & HF_UNSUPPORTED) return fail(c, AM355_E_UNSUPPORTED, "a value or key the reference does not re-encode byte for byte: the JS path decides");
    // (the value bytes of the rows must cover the valRaw column exactly: a longer column makes extra rows in the reference)
    if (d_col_len[8] != in.val_raw_len) return fail(c, AM355_E_UNSUPPORTED, "value bytes do not cover the valRaw column: the JS path decides");
    HistoryPieces pc;
    pc.chg_nops = d_nops; pc.abits = d_abits; pc.aw = (uint32_t)AW; pc.col_off = d_col_off;
    {
      uint8_t* q = d_cols;
      for (int k = 0; k < HIST_NCOL; k++) {
        pc.col_bytes[k] = q;
        if (d_col_len[k] > hb.col_cap[k]) return fail(c, AM355_E_DEVICE, "internal: encoded column larger than its bound");
        if (d_col_len[k]) HIPCHK(c, hipMemcpyAsync(q, hb.col_out[k], d_col_len[k], hipMemcpyDeviceToHost, st));
        q += ((size_t)d_col_len[k] + 255) & ~(size_t)255;
      }
      HIPCHK(c, hipStreamSynchronize(st));
    }
    lap("columns of all changes (device)");
    c->history = HistoryOutput{};
    rc = history_finish(in, meta, pc, (flags & 1) != 0, [&](unsigned k, const std::function<void(unsigned)>& fn) { c->pool->run(k, fn); }, c->history, err);
    lap("headers + hash chain");
    if (rc == HISTORY_INVALID) return fail(c, AM355_E_INVALID, "%s", err.c_str());
    if (rc) return fail(c, AM355_E_UNSUPPORTED, "%s", err.c_str());
    c->history_ok = true;
    c->history_flags = flags & 1;
  }
  *arena = c->history.arena.data();
  *offsets = c->history.offsets.data();
  *n_changes = (uint32_t)(c->history.offsets.size() - 1);
  *hashes = c->history.hashes.data();
  return AM355_OK;
}

extern "C" int am355_load_changes(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n) { return guarded(c, [&]() { return load_changes_impl(c, arena, offsets, n); }); }
extern "C" int am355_load_document(am355_ctx* c, const uint8_t* doc, size_t len) { return guarded(c, [&]() { return load_document_impl(c, doc, len); }); }
extern "C" int am355_replay(am355_ctx* c) { return guarded(c, [&]() { return replay_impl(c); }); }
extern "C" int am355_fetch_ir(am355_ctx* c, am355_patch_ir* out) { return guarded(c, [&]() { return fetch_ir_impl(c, out); }); }
extern "C" int am355_patch_json(am355_ctx* c, const char** json, size_t* len) { return guarded(c, [&]() { return patch_json_impl(c, json, len); }); }
extern "C" int am355_save(am355_ctx* c, uint32_t flags, const uint8_t** out_bytes, size_t* out_len) { return guarded(c, [&]() { return save_impl(c, flags, out_bytes, out_len); }); }
extern "C" int am355_doc_changes(am355_ctx* c, uint32_t flags, const uint8_t** arena, const uint64_t** offsets, uint32_t* n, const uint8_t** hashes) { return guarded(c, [&]() { return doc_changes_impl(c, flags, arena, offsets, n, hashes); }); }
extern "C" int am355_import_fragments(am355_ctx* c, const uint8_t* frags, const uint64_t* offsets, uint32_t world) { return guarded(c, [&]() { return import_fragments_impl(c, frags, offsets, world); }); }
extern "C" int am355_apply_changes(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n) { return guarded(c, [&]() { return apply_changes_impl(c, arena, offsets, n); }); }
extern "C" int am355_apply_patch_json(am355_ctx* c, const char** json, size_t* len) { return guarded(c, [&]() { return apply_patch_json_impl(c, json, len); }); }
extern "C" int am355_get_dep_graph(am355_ctx* c, const uint32_t** dep_first, const uint32_t** dep_index, uint32_t* n) { return guarded(c, [&]() { return get_dep_graph_impl(c, dep_first, dep_index, n); }); }
extern "C" int am355_sync_bloom_build(am355_ctx* c, const uint32_t* idx, uint32_t n, uint8_t* bits, size_t cap) {
  return guarded(c, [&]() { return sync_bloom_impl(c, idx, n, n, 10, 7, nullptr, 0, bits, cap, true); });
}
extern "C" int am355_sync_bloom_probe(am355_ctx* c, const uint32_t* idx, uint32_t n, uint32_t num_entries, uint32_t bits_per_entry, uint32_t num_probes, const uint8_t* bits,
                                      size_t n_bytes, uint8_t* contains) {
  return guarded(c, [&]() { return sync_bloom_impl(c, idx, n, num_entries, bits_per_entry, num_probes, bits, n_bytes, contains, n, false); });
}
