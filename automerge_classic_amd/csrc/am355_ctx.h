// Context of the C ABI (include/am355.h) and what its translation units share: device / pinned buffers, the host thread pool, struct
// am355_ctx, error helpers. am355_api.hip: entry points; am355_stage.hip: staging; am355_replay.hip: scheduler, plan, orchestration of
// the device stages; am355_calls.hip: patch IR, applyChanges, dependency graph, Bloom filters; am355_save.hip: save + history after
// load; am355_shard.hip: objectId sharding.
#pragma once
#include "../../include/am355.h"
#include "am355_decode.h"
#include "am355_merge.h"
#include "am355_bigcol.h"
#include "am355_encode.h"
#include "am355_prims.h"
#include "am355_render.h"
#include "am355_host.h"
#include "am355_pinflate.h"
#include "am355_history.h"
#include "am355_delta.h"
#include "am355_resorder.h"
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

namespace am355_host {   // (host-side helpers of the C ABI's translation units)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  uint32_t lost = 0;   // times the CONTENT was given up (ensure that had to grow, release): what a kept state checks (am355_replay.hip resident_mark)
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    lost++;
    canary_forget(p, cap);
    // (a buffer that grows AGAIN belongs to a document that is growing call by call -- Backend.applyChanges in a loop --: half as much
    // again, so that a reallocation, ~1 ms of hipFree + hipMalloc, comes once in dozens of calls instead of once in a few)
    const size_t slack = cap ? bytes / 2 : bytes / 8;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + slack + 256 + (canary_on() ? (128u << 10) : 0u);
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
    lost++;
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
  uint32_t lost = 0;   // as DevBuf::lost
  // grows like ensure() but carries the first `keep` bytes over
  bool ensure_keep(size_t bytes, size_t keep) {
    if (bytes <= cap) return true;
    size_t want = bytes + bytes / 2 + 256;
    void* q = nullptr;
    if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) return false;
    if (p && keep) memcpy(q, p, keep);
    if (p) (void)hipHostFree(p);
    p = q;
    cap = want;
    return true;
  }
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    lost++;
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

// One helper thread with a one-slot mailbox: the replay hands it the enqueueing of the hash stream's commands (ten HIP calls,
// ~40 us of host time) while the calling thread enqueues the critical stream -- the host's launch rate, not the device, sets the pace
// of a 0.4 ms replay. Polls for a few milliseconds after a job (replays come in bursts), sleeps otherwise.
class AsyncLane {
 public:
  AsyncLane() : t_([this]() { loop(); }) {}
  ~AsyncLane() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    t_.join();
  }
  void post(std::function<void()> job) {
    job_ = std::move(job);
    state_.store(1, std::memory_order_release);
    if (sleeping_.load(std::memory_order_acquire)) { std::lock_guard<std::mutex> l(m_); cv_.notify_one(); }
  }
  bool busy() const { return state_.load(std::memory_order_acquire) != 0; }
  void wait() {  // returns when the posted job has run (at once when nothing is posted)
    while (state_.load(std::memory_order_acquire) != 0) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
 private:
  void loop() {
    uint64_t hot_until = 0;
    for (;;) {
      if (state_.load(std::memory_order_acquire) == 1) {
        job_();
        job_ = nullptr;
        state_.store(0, std::memory_order_release);
        hot_until = now_us() + 3000;
        continue;
      }
      if (now_us() < hot_until) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        continue;
      }
      std::unique_lock<std::mutex> l(m_);
      sleeping_.store(true, std::memory_order_release);
      cv_.wait(l, [&]() { return stop_ || state_.load(std::memory_order_acquire) == 1; });
      sleeping_.store(false, std::memory_order_release);
      if (stop_) return;
    }
  }
  static uint64_t now_us() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  std::function<void()> job_;
  std::atomic<int> state_{0};
  std::atomic<bool> sleeping_{false};
  std::mutex m_;
  std::condition_variable cv_;
  bool stop_ = false;
  std::thread t_;  // (last member: the thread starts with everything above constructed)
};

struct Hash32 {
  uint8_t b[32];
  bool operator==(const Hash32& o) const { return memcmp(b, o.b, 32) == 0; }
};
struct Hash32Hasher {
  size_t operator()(const Hash32& h) const { size_t v; memcpy(&v, h.b, sizeof v); return v; }
};

}  // namespace am355_host
using namespace am355_host;

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
  std::unique_ptr<AsyncLane> lane; // enqueues the hash stream's commands beside the calling thread (created on first use)
  std::vector<uint64_t> raw_off;
  uint32_t n_changes = 0;
  // speculative decode launch (am355_internal.h decode_gate_open): op rows carved for a capacity before the totals are known
  DevBuf d_plan_totals;              // PlanTotals, device copy (k_plan_apply)
  bool spec_launched = false;        // this replay's decode kernels of the wave classes are enqueued behind the plan kernel
  bool spec_large_launched = false;  // ... the large class among them (only when the previous batch had such changes)
  uint32_t spec_cap_ops = 0, spec_cap_preds = 0;   // what c->cols is carved for while spec_launched
  uint32_t hint_ops = 0, hint_preds = 0, hint_large = 0;   // totals (and large-class changes) of the context's previous in-order replay
  bool staged = false, replayed = false, ir_fetched = false;
  int ir_copy_enqueued = 0;          // the IR tables are on their way to h_ir (1: without the edit table, 2: all three); reset by every replay
  bool prefetch_ir = false;          // am355_backend_load: the replay enqueues that copy itself
  bool has_unknown_cols = false;     // some change carries columns outside the modelled set (kept by the reference's save)
  bool staging_in_flight = false;    // am355_load_changes returned with its H2D copies still running on `stream`
  bool is_document = false;          // staged input is one saved document (am355_load_document) rather than changes
  ChangeMeta doc_meta{};             // column layout of the staged document inside `raw`
  std::vector<uint32_t> doc_actor_rank;  // document actor index -> lexicographic rank
  DevBuf d_arena, d_offsets, d_metas;
  // small host -> device copies put off until the next launch on `stream`, where they go out as ONE kernel (am355_prims.h CopyRanges;
  // sources: pinned memory that stays as it is until that kernel has run). flush_uploads() before anything that reads the destinations.
  uint64_t arena_epoch = 1;           // am355_arena_epoch: bumped whenever `raw` is rewritten rather than appended to
  CopyRanges pending_up;
  std::vector<uint32_t> breaks_dev;   // the delta stage's `breaks` table as am355_apply_changes queued it ahead of the replay (breaks_dev_ptr: in which allocation)
  void* breaks_dev_ptr = nullptr;
  HostBuf h_breaks_ahead;
  bool offsets_on_device = false;      // d_offsets holds raw_off of every staged change (a batch staged behind a kept state defers the copy)
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
  // Chunk checksum of the staged document (columnar.js:699-705: one SHA-256 over the whole chunk, ~20 ms for 44 MB on a core with SHA
  // extensions -- a dependent chain, the longest single item of Backend.load): a thread of its own hashes the context's COPY of the
  // document piece by piece as the copy tasks of the staging job finish them. am355_load_document waits for the verdict before it
  // returns; am355_backend_load lets the device stages of the load run beside it and asks at the end.
  struct DocSum {
    static constexpr size_t PIECE = (size_t)4 << 20;
    std::thread t;
    std::unique_ptr<std::atomic<uint8_t>[]> copied;  // per PIECE of doc_bytes: bytes are in place
    size_t n_pieces = 0;
    bool pending = false, ok = false;
    bool wait() {  // the verdict (joins the thread)
      if (t.joinable()) t.join();
      pending = false;
      return ok;
    }
    ~DocSum() { if (t.joinable()) t.join(); }
  } doc_sum;
  std::vector<uint8_t> doc_bytes;                          // the loaded document as given (Backend.save of an unchanged document returns it)
  std::vector<uint8_t> saved;                              // result of am355_save
  HistoryOutput history;                                   // result of am355_doc_changes
  bool history_ok = false; uint32_t history_flags = 0;
  std::vector<std::vector<uint8_t>> inflate_scratch;       // am355_load_document: inflated columns, longest first (capacity kept between loads)
  std::vector<std::unique_ptr<PInflateJob>> pinflate_jobs;  // am355_load_document: chunked decode of the long DEFLATE streams (symbol buffers kept between loads)
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
  bool key_unbounded = false;      // replay_document: second attempt, the key stream's walker follows literals without a bound
  bool device_scheduled = false;   // the last general-path replay was scheduled by the device (am355_sched.hip)
  HostBuf h_delta;
  DeltaBufs delta{};
  bool h_tables_current = false, h_tables_were_current = false;   // the object and map tables in h_ir (c->hir) are the device's (set by a completed fetch, dropped by every replay that rebuilds them)
  bool batch_list_only = false;      // the last replay merged a batch of plain list edits in place (replay_resident): no map row among the new rows
  void* apply_tail = nullptr;        // (am355_calls.hip ApplyTail of the delta stage that is running)
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
  // Lineage that began with Backend.load: the reference schedules against the hashes it KNOWS -- the document's heads and what later
  // calls applied -- until a round applies nothing; then it rebuilds the document's hash graph (new.js:1822-1841, computeHashGraph
  // :1887-1912) and, in that call, forgets the hashes of the changes the call had applied so far. graph_mode: 0 = the graph is known
  // (every other lineage), 1 = not rebuilt yet, 2 = the host replayed retained changes and does not know which (both are tried).
  uint32_t graph_mode = 0, doc_n_changes = 0, sched_prefix = 0;
  bool doc_head_index_known = true, sched_graph_after = true;
  bool doc_graph_known = false;  // document context: the host served a query that makes the reference rebuild the hash graph
  uint64_t doc_rows = 0;    // rows [0, doc_rows) of the state are the rebuilt history of a loaded document (valid while no_history)
  bool doc_rows_known = false;
  bool no_history = false;  // the staged changes are the rebuilt history of a LOADED document: the reference's objectMeta came from one pass over the document
  DevBuf d_breaks;
  HostBuf h_delta_tabs;  // pinned: stream breaks | pass rows of a delta stage on their way to the device (run_delta_stage)
  bool state_checked = false;  // the state was built by am355_apply_changes calls (each checked for what later patches depend on) or is empty
  std::string apply_json;

  // Resident state (am355_apply_changes onto a state this context holds; am355_replay.hip replay_resident): when the last replay left
  // the op rows, their per-row results and the per-change tables of exactly the applied changes in HBM -- in staged order, nothing
  // queued -- the next call parses, hashes, decodes and resolves the BATCH alone and appends. In apply mode the row arrays are carved
  // for a capacity (cols_cap_*) so that they stay where they are while the document grows.
  struct Keep { bool want = false; uint32_t n_changes = 0; uint64_t n_ops = 0, n_preds = 0; } keep;   // apply_changes_impl -> replay_impl
  uint32_t cols_cap_ops = 0, cols_cap_preds = 0;   // what c->cols / c->mb's per-row arrays are carved for (>= n_ops, n_preds)
  bool resident_valid = false;
  struct ResidentMark { uint32_t lost[8] = {}; uint32_t n_changes = 0; uint64_t n_ops = 0, n_preds = 0; } res_mark;
  uint32_t seed_list_inc = 0;                      // Counts.n_list_inc after the last replay
  std::vector<uint32_t> hash_index;                // open addressing over h_hashes: change index + 1 (0 empty); rebuilt when hash_index_n != applied changes
  uint32_t hash_index_n = 0;
  // actor ids -> ranks for the host's schedule of a batch: the document's table inverted, and per author the "other actors" table of its
  // last change with the ranks it gave (changes of one author nearly always carry the same table: one memcmp instead of 64 lookups);
  // both describe c->actors and are dropped with it (a full replay ranks the actors anew)
  std::vector<uint32_t> res_rank_of;   // actor id -> rank: open addressing over c->actors (rank + 1; keyed by the id's first bytes), built for res_rank_n actors
  uint32_t res_rank_n = 0;
  struct ActorMemo { std::vector<uint8_t> bytes; std::vector<uint32_t> ranks; };
  std::vector<ActorMemo> res_actor_memo;
  // resident list ORDER (am355_resorder.hip): the new elements of a small list-only batch are merged into the stored order; the
  // whole-document edit tables (ir.edit ...) are then stale until somebody asks for them (ensure_ir_fresh)
  DevBuf d_pos, d_order_alt, d_resorder;           // pos_of[row] | the other order buffer | the stage's scratch
  uint32_t* order_alt_ptr = nullptr;               // whichever of the two order arrays c->mb.order does NOT point to (null: not set up since the last carve)
  HostBuf h_resorder;                              // pinned: its verdict words + the merge counters' flag word
  bool pos_valid = false;                          // d_pos describes c->mb.order
  bool ir_stale = false;                           // rows / order are current, the whole-document patch tables are not
  uint64_t n_resorder_calls = 0;
  uint64_t n_maps_only_calls = 0;   // resident calls that ran merge_run_maps: plain map rows only, or beside list edits merged in place
  HostBuf h_res_metas;                             // pinned: the batch's ChangeMetas on their way to the host
  uint32_t res_dep_base = 0;                       // changes >= this were applied by resident calls: their dependency indexes live in ...
  std::vector<uint32_t> res_dep_first, res_dep_index;   // ... CSR over (change - res_dep_base)
  uint64_t n_resident_calls = 0, n_resident_fallbacks = 0;
  std::string resident_why;                        // why the last attempt fell back (diagnostics, AM355_TRACE)

  // objectId sharding (am355_set_shard): this context merges the objects rank `shard_rank` of `shard_world` owns
  uint32_t shard_rank = 0, shard_world = 1;
  std::vector<uint8_t> stitched;  // am355_import_fragments: the combined record tables
  // am355_shard_init: this context's RCCL communicator (one rank per GPU, one process per rank) and the collective's buffers
  void* shard_comm = nullptr;       // ncclComm_t
  DevBuf d_shard_send, d_shard_recv, d_shard_sizes;
  HostBuf h_shard_sizes, h_shard_frags;
  std::vector<uint64_t> shard_fragment_bytes;   // of the last am355_sharded_replay, per rank
};

static inline int fail(am355_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  c->err = buf;
  c->pending_up.n = 0;   // (uploads a failed call queued and never launched must not go out later, into buffers that may have moved)
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
static inline int guarded(am355_ctx* c, F body) {
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

// ---- shared by the translation units of the C ABI (am355_api / _stage / _replay / _calls / _save / _shard .hip) ----
static inline int bits_for64(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b)) b++;
  return b;
}
template <class T>
static inline T* carve(uint8_t*& p, size_t count) {
  T* r = (T*)p;
  canary_note(p, count * sizeof(T));
  p += carve_round(count * sizeof(T));
  return r;
}

static inline size_t carve_size(size_t count, size_t elem) { return carve_round(count * elem); }
static inline uint32_t pow2_at_least(uint64_t v) {
  uint32_t p = 64;
  while (p < v) p <<= 1;
  return p;
}
// words shared with the device: [0] flags of the critical-path kernels, [1] fast-path word (stream A part),
// [2] total actor-table entries, [3] flags of the hash stream, [4] fast-path word (stream B part)
enum { W_FLAGS_A = 0, W_FAST_A = 1, W_TOTAL_ENTRIES = 2, W_FLAGS_B = 3, W_FAST_B = 4, W_NUM = 8 };
static inline int error_for_flags(am355_ctx* c, uint32_t f, const char* what) {
  c->flags |= f;
  uint32_t hard = f & ~(uint32_t)(F_OVERFLOW | F_UNSUPPORTED);
  return fail(c, hard ? AM355_E_INVALID : AM355_E_UNSUPPORTED, "%s (flags 0x%x)", what, f);
}

// staging (am355_stage.hip)
bool read_uleb_host(const uint8_t* p, size_t len, size_t& off, uint64_t& out);
int load_changes_impl(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n, bool keep_staged = false);
// dst (HBM) <- src (pinned) on c->stream: queued when small and room is left, else copied at once (after what is queued)
int queue_upload(am355_ctx* c, void* dst, const void* src, size_t bytes);
int flush_uploads(am355_ctx* c);
int upload_offsets(am355_ctx* c);   // the staged changes' offsets table to HBM, if it is not there (am355_stage.hip)
int load_document_impl(am355_ctx* c, const uint8_t* doc, size_t len, bool defer_checksum = false);
int backend_load_impl(am355_ctx* c, const uint8_t* doc, size_t len);
int ir_copy_enqueue(am355_ctx* c, bool with_edits);
// replay: host scheduler / plan, device buffers, orchestration of the device stages (am355_replay.hip)
int setup_buffers(am355_ctx* c, uint32_t NA);
int replay_impl(am355_ctx* c);
int ensure_ir_fresh(am355_ctx* c);   // rebuilds the whole-document patch tables when a resident call left them stale (ir_stale)
// the calls on a replayed state: patch IR to the host, Backend.applyChanges, dependency graph, Bloom filters (am355_calls.hip)
int fetch_ir_impl(am355_ctx* c, am355_patch_ir* out, bool with_edits = true);
int patch_json_impl(am355_ctx* c, const char** json, size_t* len);
int apply_changes_impl(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n);
int apply_patch_json_impl(am355_ctx* c, const char** json, size_t* len);
int get_dep_graph_impl(am355_ctx* c, const uint32_t** dep_first, const uint32_t** dep_index, uint32_t* n_changes);
int sync_bloom_impl(am355_ctx* c, const uint32_t* idx, uint32_t n, uint32_t num_entries, uint32_t bits_per_entry, uint32_t num_probes, const uint8_t* bits, size_t n_bytes,
                    uint8_t* out, size_t cap, bool build);
// documents out: Backend.save, history after load (am355_save.hip); objectId sharding (am355_shard.hip)
int save_impl(am355_ctx* c, uint32_t flags, const uint8_t** out_bytes, size_t* out_len);
int doc_changes_impl(am355_ctx* c, uint32_t flags, const uint8_t** arena, const uint64_t** offsets, uint32_t* n_changes, const uint8_t** hashes);
int import_fragments_impl(am355_ctx* c, const uint8_t* frags, const uint64_t* offsets, uint32_t world);
