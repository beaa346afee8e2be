// C ABI of the replay engine (include/am355.h): context, staging, host-side causal scheduler, orchestration of
// the device stages, patch-IR download.
//
// Host-side logic restated from the reference (paths relative to the reference tree):
//   inflate of DEFLATEd changes    backend/columnar.js:813-823 inflateChange (zlib raw inflate)
//   causal scheduling              backend/new.js:1550-1597 applyChanges, :1822-1841 retry loop
//   actor table                    backend/new.js:1434-1451 getActorTable (first-applied order; the engine
//                                  additionally ranks actors lexicographically for numeric op-id comparison)
//   envelope                       backend/new.js:1870-1873, 2064-2067 (maxOp, clock, deps, pendingChanges)
#include "am355_ctx.h"

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
  if (c->shard_comm) (void)am355_shard_finalize(c);   // (ncclCommDestroy)
  c->d_shard_send.release(); c->d_shard_recv.release(); c->d_shard_sizes.release(); c->h_shard_sizes.release(); c->h_shard_frags.release();
  c->d_delta.release(); c->d_delta_edit.release(); c->d_sched.release(); c->h_sched.release(); c->d_hist.release(); c->d_pass.release(); c->d_breaks.release(); c->h_delta_tabs.release(); c->h_breaks_ahead.release(); c->h_delta.release(); c->d_sync.release();
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

extern "C" int am355_arena_epoch(const am355_ctx* c, uint64_t* epoch) {
  if (!c || !epoch) return AM355_E_ARG;
  *epoch = c->arena_epoch;
  return AM355_OK;
}

extern "C" int am355_get_hashes_range(const am355_ctx* c, uint32_t first, uint32_t count, uint8_t* out) {
  if (!c || (!out && count)) return AM355_E_ARG;
  if (!c->replayed || !c->h_hashes.p || c->is_document) return AM355_E_STATE;
  if (first > c->n_changes || count > c->n_changes - first) return AM355_E_ARG;
  if (count) memcpy(out, (const uint8_t*)c->h_hashes.p + 32 * (size_t)first, 32 * (size_t)count);
  return AM355_OK;
}

extern "C" int am355_applied_in_input_order(const am355_ctx* c, int* yes) {
  if (!c || !yes) return AM355_E_ARG;
  if (!c->replayed || c->is_document) return AM355_E_STATE;
  bool ok = c->pending_change.empty() && c->n_pending == 0 && c->applied_change.size() == c->n_changes;
  for (uint32_t i = 0; ok && i < c->n_changes; i++) ok = c->applied_change[i] == i;
  *yes = ok ? 1 : 0;
  return AM355_OK;
}

extern "C" int am355_resident_counters(const am355_ctx* c, uint64_t out[3]) {
  if (!c || !out) return AM355_E_ARG;
  out[0] = c->n_resident_calls;
  out[1] = c->n_resident_fallbacks;
  out[2] = c->n_resorder_calls;
  return AM355_OK;
}

extern "C" int am355_resident_maps_only_calls(const am355_ctx* c, uint64_t* out) {
  if (!c || !out) return AM355_E_ARG;
  *out = c->n_maps_only_calls;
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
  if (c->shard_world > 1) return fail(c, AM355_E_UNSUPPORTED, "am355_get_rows on a sharded context (foreign rows are decoded as far as their object columns only)");
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

extern "C" int am355_load_changes(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n) { return guarded(c, [&]() { return load_changes_impl(c, arena, offsets, n); }); }
extern "C" int am355_load_document(am355_ctx* c, const uint8_t* doc, size_t len) { return guarded(c, [&]() { return load_document_impl(c, doc, len); }); }
extern "C" int am355_backend_load(am355_ctx* c, const uint8_t* doc, size_t len) { return guarded(c, [&]() { return backend_load_impl(c, doc, len); }); }
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

