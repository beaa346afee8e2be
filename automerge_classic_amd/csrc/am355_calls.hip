// Calls on a replayed state (include/am355.h): patch IR to the host, Backend.applyChanges (delta stage orchestration), dependency graph,
// Bloom filters of the sync protocol. See am355_ctx.h.
#include "am355_ctx.h"

// The three record tables, device -> pinned host memory, enqueued on the context's stream (no wait). am355_backend_load starts the
// copy right behind the device stages, beside the tail of the checksum thread; am355_fetch_ir otherwise.
int ir_copy_enqueue(am355_ctx* c, bool with_edits) {
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
  c->ir_copy_enqueued = with_edits ? 2 : 1;
  return AM355_OK;
}

int fetch_ir_impl(am355_ctx* c, am355_patch_ir* out, bool with_edits) {
  if (!c) return AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  (void)hipSetDevice(c->device);
  if (with_edits && c->ir_stale) {   // (the object and map tables of a state are current after an in-place list merge; its edit tables are not)
    int frc = ensure_ir_fresh(c);
    if (frc) return frc;
  }
  if (!c->ir_fetched) {
    hipStream_t st = c->stream;
    am355_patch_ir& h = c->hir;
    if (c->ir_copy_enqueued < (with_edits ? 2 : 1)) {
      int erc = ir_copy_enqueue(c, with_edits);
      if (erc) return erc;
    }
    c->ir_copy_enqueued = 0;
    // (a ~0.1 ms copy: polled, not slept on -- a blocking wait adds an interrupt wake-up of tens of microseconds to a call of 140)
    {
      hipError_t q;
      while ((q = hipStreamQuery(st)) == hipErrorNotReady) {}
      if (q != hipSuccess) HIPCHK(c, q);
    }
    c->h_tables_current = true;
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

int patch_json_impl(am355_ctx* c, const char** json, size_t* len) {
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
// (tail: the host copies of the stage's tables, enqueued by the stage itself in front of its last wait -- apply_tail_enqueue below)
struct ApplyTail {
  am355_ctx* c = nullptr;
  uint32_t n_obj = 0, n_dmap = 0;
  size_t edit_records = 0;       // records of d.edit on their way (>= n_erecs + 1 when the stage ends well)
  ObjLink* h_link = nullptr;
  am355_ir_map* h_map = nullptr;
  am355_ir_edit* h_edit = nullptr;
  bool enqueued = false;
};
static void apply_tail_enqueue(am355_ctx* c, ApplyTail* t, const DeltaCounts* mid, size_t rec_bound) {
  // a bounded guess of the edit table is cheap for the small batches this is for: larger ones copy the exact size after the last wait
  if (!t) return;
  t->enqueued = false;   // (a second call of one stage: what the first put on its way is not what the stage ends with)
  if (rec_bound * sizeof(am355_ir_edit) > ((size_t)256 << 10)) return;
  hipStream_t st = c->stream;
  DeltaBufs& d = c->delta;
  const uint32_t NO = t->n_obj, n_dmap = mid->n_kept + mid->n_place;
  const size_t n_rec = std::min<size_t>(rec_bound, d.edit_cap);
  const size_t b_link = carve_size(NO, sizeof(ObjLink)), b_map = carve_size(n_dmap, sizeof(am355_ir_map)), b_edit = carve_size(n_rec, sizeof(am355_ir_edit));
  if (!c->h_delta.ensure(b_link + b_map + b_edit + 256)) return;
  uint8_t* hp = c->h_delta.as<uint8_t>();
  t->h_link = (ObjLink*)hp;
  t->h_map = (am355_ir_map*)(hp + b_link);
  t->h_edit = (am355_ir_edit*)(hp + b_link + b_map);
  if (hipMemcpyAsync(t->h_link, d.link, sizeof(ObjLink) * (size_t)NO, hipMemcpyDeviceToHost, st) != hipSuccess) return;
  if (n_dmap && hipMemcpyAsync(t->h_map, d.map, sizeof(am355_ir_map) * (size_t)n_dmap, hipMemcpyDeviceToHost, st) != hipSuccess) return;
  if (hipMemcpyAsync(t->h_edit, d.edit, sizeof(am355_ir_edit) * n_rec, hipMemcpyDeviceToHost, st) != hipSuccess) return;
  t->n_dmap = n_dmap;
  t->edit_records = n_rec;
  t->enqueued = true;
}

static int run_delta_stage(am355_ctx* c, uint32_t T0, DeltaCounts* hc, bool check_only, ApplyTail* tail = nullptr) {
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
  const bool breaks_there = c->breaks_dev_ptr && c->breaks_dev_ptr == c->d_breaks.p && c->breaks_dev == breaks;   // (am355_apply_changes sent them ahead)
  if (!breaks_there && !c->d_breaks.ensure(4 * (breaks.size() + 1))) return fail(c, AM355_E_NOMEM, "device allocation failed (delta)");
  d.n_breaks = (uint32_t)breaks.size();
  d.breaks = c->d_breaks.as<uint32_t>();
  d.breaks_exact = c->breaks_exact ? 1u : 0u;
  d.T_doc = c->no_history && c->doc_rows_known ? (uint32_t)c->doc_rows : 0u;
  // (from pinned memory: no bounce through the driver and no wait -- the tables are read by kernels enqueued behind these copies, and
  // the pinned words are only rewritten by the next stage, which starts after this one has drained the stream)
  if (!c->h_delta_tabs.ensure(4 * (breaks.size() + pass_rows.size() + 2))) return fail(c, AM355_E_NOMEM, "host allocation failed (delta)");
  uint32_t* h_breaks = c->h_delta_tabs.as<uint32_t>();
  uint32_t* h_pass = h_breaks + breaks.size();
  if (!breaks.empty() && !breaks_there) {
    memcpy(h_breaks, breaks.data(), 4 * breaks.size());
    HIPCHK(c, hipMemcpyAsync(c->d_breaks.p, h_breaks, 4 * breaks.size(), hipMemcpyHostToDevice, st));
    c->breaks_dev_ptr = nullptr;   // (what the device holds is this call's: nothing sent ahead)
  }
  if (d.n_pass) {
    memcpy(h_pass, pass_rows.data(), 4 * pass_rows.size());
    HIPCHK(c, hipMemcpyAsync(c->d_pass.p, h_pass, 4 * pass_rows.size(), hipMemcpyHostToDevice, st));
  }
  auto grow = [](void* user, size_t records) -> am355_ir_edit* {
    am355_ctx* cx = (am355_ctx*)user;
    return cx->d_delta_edit.ensure(sizeof(am355_ir_edit) * records) ? cx->d_delta_edit.as<am355_ir_edit>() : nullptr;
  };
  d.sig = c->h_sig.as<HostSignals>();
  d.sig_seq = ++c->sig_seq;
  d.list_only = T0 && c->batch_list_only && !check_only ? 1u : 0u;
  d.host_link = nullptr;
  d.host_edit = nullptr;
  if (tail && d.list_only && NN <= 1024 && NO <= 4096 && !getenv("AM355_DELTA_COPY_TABLES")) {
    // a batch kd_edit_small may serve: it writes the two small tables the assembly reads straight into pinned memory
    const size_t b_link = carve_size(NO, sizeof(ObjLink)), b_edit = carve_size((size_t)d.edit_cap + 1, sizeof(am355_ir_edit));
    if (c->h_delta.ensure(b_link + b_edit + 256)) {
      d.host_link = (ObjLink*)c->h_delta.as<uint8_t>();
      d.host_edit = (am355_ir_edit*)(c->h_delta.as<uint8_t>() + b_link);
    }
  }
  c->apply_tail = tail;
  auto before_end = [](void* user, const DeltaCounts* mid, size_t rec_bound) {
    am355_ctx* cx = (am355_ctx*)user;
    apply_tail_enqueue(cx, (ApplyTail*)cx->apply_tail, mid, rec_bound);
  };
  delta_run(c->mb, c->ir, d, hc, st, check_only, grow, c, tail ? (DeltaBeforeEnd)before_end : nullptr);
  c->apply_tail = nullptr;
  if (tail && d.host_edit && !hc->flags) {   // (kd_edit_small handed the tables over itself)
    tail->h_link = d.host_link; tail->h_map = nullptr; tail->h_edit = d.host_edit;
    tail->n_dmap = 0; tail->edit_records = (size_t)hc->n_erecs + 1;
    tail->enqueued = true;
  }
  HIPCHK(c, hipGetLastError());
  return AM355_OK;
}

int apply_changes_impl(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n) {
  if (!c || (!arena && n) || !offsets) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  (void)hipSetDevice(c->device);
  c->apply_ready = false;
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "apply_changes: %-26s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  if (c->shard_world > 1) return fail(c, AM355_E_UNSUPPORTED, "am355_apply_changes on a sharded context");
  if (c->staged && c->is_document) {
    // Backend.applyChanges onto a loaded document. The reference rebuilds the document's changes first (computeHashGraph,
    // new.js:1887-1912, called from applyChanges :1809): the batch is scheduled against THEIR hashes. So does the engine
    // (am355_doc_changes: the device regroups rows into changes and encodes them, am355_hist.hip), then replays the rebuilt changes as
    // the state the batch goes onto. What differs from a state that calls built is the reference's objectMeta: it came from one pass
    // over the document's rows (new.js:1604-1635), which the delta stage is told by the number of rows the document had (T_doc).
    if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must be called after am355_load_document");
    const uint8_t *ha = nullptr, *hh = nullptr;
    const uint64_t* ho = nullptr;
    uint32_t hn = 0;
    const bool head_index_known = !c->doc_tail.empty() || c->heads.size() <= 32;  // (heads.length === headsIndexes.length, or one head: new.js:1719-1729)
    // (the form the host last asked for, if it did: the history is kept per context and DEFLATEd changes stage like any others)
    int hrc = doc_changes_impl(c, c->history_ok ? c->history_flags : 0, &ha, &ho, &hn, &hh);
    if (hrc) return hrc;
    if (hn == 0) {  // (an empty document: Backend.init())
      int rrc = am355_reset(c);
      if (rrc) return rrc;
    } else {
    std::vector<uint8_t> hist(ha + ho[0], ha + ho[hn]);
    std::vector<uint64_t> hoff(hn + 1);
    for (uint32_t i = 0; i <= hn; i++) hoff[i] = ho[i] - ho[0];
    lap("document history rebuilt");
    hrc = load_changes_impl(c, hist.data(), hoff.data(), hn, false);
    if (!hrc) hrc = replay_impl(c);
    if (hrc) { c->staged = false; return hrc; }
    bool in_order = c->pending_change.empty() && c->applied_change.size() == hn;
    for (uint32_t i = 0; in_order && i < hn; i++) in_order = c->applied_change[i] == i;
    if (!in_order) { c->staged = c->replayed = false; return fail(c, AM355_E_DEVICE, "internal: the rebuilt changes of the document did not apply in document order"); }
    c->stream_breaks.clear();
    c->breaks_exact = true;
    c->children_hazard = false;
    c->state_checked = true;  // (no merge call built this state: nothing was skipped)
    c->no_history = true;
    c->doc_rows = c->n_ops;
    c->doc_rows_known = true;
    c->graph_mode = c->doc_graph_known ? 0 : 1;  // (the reference has the document's heads only, until a round of its retry loop applies nothing or a query rebuilds the graph)
    c->doc_n_changes = hn;
    c->doc_head_index_known = head_index_known;
    lap("document history replayed");
    }
  }
  const bool have_state = c->staged;
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
  const uint64_t old_ops = have_state ? c->n_ops : 0, old_preds = have_state ? c->n_preds : 0;
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
  if (!have_state) { c->stream_breaks.clear(); c->breaks_exact = true; c->children_hazard = false; c->no_history = false; c->doc_rows_known = false; c->graph_mode = 0; }
  c->in_apply = true;
  c->sched_prefix = n_old_applied;
  // the state the context holds stays where it is and the batch is merged into it (am355_replay.hip replay_resident) when the staged
  // changes are exactly the applied ones; AM355_NO_RESIDENT=1: the full replay for every call (rounds 3-5; A/B and tests)
  // (measured, profiles/r06_s3_apply_seq_big.txt: the batch's host-side work -- headers, SHA-256, schedule: ~2 us per change -- and the
  // chunk-by-chunk list merge meet what the skipped device stages cost at about 150 changes of 250 ops -- 100 changes 0.55 against
  // 0.67 ms, 144 changes 0.65 against 0.68-0.75, 200 changes the same --; larger batches take the full replay, whose stage 1 runs on
  // the device. AM355_RESIDENT_MAX)
  static const uint32_t resident_max = []() { const char* e = getenv("AM355_RESIDENT_MAX"); return e && atol(e) > 0 ? (uint32_t)atol(e) : 144u; }();
  c->keep.want = append && n > 0 && n <= resident_max && !getenv("AM355_NO_RESIDENT");
  c->keep.n_changes = n_old_applied;
  c->keep.n_ops = old_ops;
  c->keep.n_preds = old_preds;
  c->breaks_dev_ptr = nullptr;
  if (c->keep.want && old_ops) {
    // the rows at which the op streams so far began, as the delta stage of this call will want them when the batch applies in one
    // pass: queued with the replay's own small uploads (one launch for all of them), recognised by run_delta_stage
    std::vector<uint32_t> br;
    for (uint32_t r : c->stream_breaks) if (r < old_ops) br.push_back(r);
    br.push_back((uint32_t)old_ops);
    if (c->d_breaks.ensure(4 * (br.size() + 1)) && c->h_breaks_ahead.ensure(4 * br.size())) {
      memcpy(c->h_breaks_ahead.p, br.data(), 4 * br.size());
      if (queue_upload(c, c->d_breaks.p, c->h_breaks_ahead.p, 4 * br.size()) == AM355_OK) { c->breaks_dev.swap(br); c->breaks_dev_ptr = c->d_breaks.p; }
    }
  }
  rc = replay_impl(c);
  c->in_apply = false;
  if (!rc && c->graph_mode == 1 && c->sched_graph_after) c->graph_mode = 0;  // (this call made the reference rebuild the hash graph)
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
  // (the whole-document object / map tables setupPatches reads are final since the replay: their copy goes in front of the stage and is
  // covered by the stage's first wait; the stage's own tables follow behind its last kernel -- ApplyTail)
  if (!c->ir_fetched && c->ir_copy_enqueued < 1) { int erc = ir_copy_enqueue(c, false); if (erc) return drop_state(erc); }
  ApplyTail tail;
  tail.c = c;
  tail.n_obj = NO;
  rc = run_delta_stage(c, (uint32_t)old_ops, &hc, false, &tail);
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
  ObjLink* h_link;
  am355_ir_map* h_map;
  am355_ir_edit* h_edit;
  if (tail.enqueued && tail.n_dmap == n_dmap && (size_t)n_dedits + 1 <= tail.edit_records) {
    h_link = tail.h_link; h_map = tail.h_map; h_edit = tail.h_edit;   // (on the host since the stage's last wait)
  } else {
    size_t b_link = carve_size(NO, sizeof(ObjLink)), b_map = carve_size(n_dmap, sizeof(am355_ir_map)), b_edit = carve_size((size_t)n_dedits + 1, sizeof(am355_ir_edit));
    if (!c->h_delta.ensure(b_link + b_map + b_edit + 256)) return fail(c, AM355_E_NOMEM, "host allocation failed");
    uint8_t* hp = c->h_delta.as<uint8_t>();
    h_link = (ObjLink*)hp;
    h_map = (am355_ir_map*)(hp + b_link);
    h_edit = (am355_ir_edit*)(hp + b_link + b_map);
    HIPCHK(c, hipMemcpyAsync(h_link, d.link, sizeof(ObjLink) * (size_t)NO, hipMemcpyDeviceToHost, st));
    if (n_dmap) HIPCHK(c, hipMemcpyAsync(h_map, d.map, sizeof(am355_ir_map) * (size_t)n_dmap, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h_edit, d.edit, sizeof(am355_ir_edit) * ((size_t)n_dedits + 1), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
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
    if (rc != AM355_E_UNSUPPORTED || need.empty() || round == 16 || need.size() > 256 || (c->no_history && !c->doc_rows_known)) break;
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
  c->resident_valid = false;
  c->arena_epoch++;
  c->state_checked = true;
  c->stream_breaks.clear();
  c->breaks_exact = true;
  c->children_hazard = false;
  c->no_history = false;
  c->doc_rows_known = false;
  c->graph_mode = 0;
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
  if (from_document) {
    // the first `from_document` staged changes are the document's: its rows are the ops of those changes, when they were applied first
    // and in their order (the rebuilt history is a causal order: they are, unless the host staged something else)
    c->no_history = true;
    c->doc_rows_known = false;
    c->graph_mode = 2;  // (whether a later call made the reference rebuild the hash graph is not known here: both are tried)
    c->doc_n_changes = (uint32_t)from_document;
    c->doc_head_index_known = true;
    const uint32_t nd = (uint32_t)from_document;
    bool ok = c->replayed && !c->is_document && nd <= c->applied_change.size();
    for (uint32_t i = 0; ok && i < nd; i++) ok = c->applied_change[i] == i;
    if (ok) {
      c->doc_rows = nd < c->applied_change.size() ? c->applied_op_base[nd] : c->n_ops;
      c->doc_rows_known = true;
    }
  }
  return AM355_OK;
}

extern "C" int am355_hash_graph_known(am355_ctx* c, int set, int* known) {
  if (!c) return AM355_E_ARG;
  if (!c->staged) return fail(c, AM355_E_STATE, "the context holds no state");
  if (c->is_document) {
    if (set >= 0) c->doc_graph_known = set != 0;
    if (known) *known = c->doc_graph_known ? 1 : 0;
    return AM355_OK;
  }
  if (set >= 0 && c->graph_mode != 0) c->graph_mode = set ? 0 : 1;  // (a lineage without a document has no such state)
  if (known) *known = c->graph_mode == 0 ? 1 : 0;
  return AM355_OK;
}

extern "C" int am355_get_pending(const am355_ctx* c, uint32_t* out, uint32_t* n_pending) {
  if (!c || !n_pending) return AM355_E_ARG;
  if (!c->replayed || c->is_document) return AM355_E_STATE;
  *n_pending = (uint32_t)c->pending_change.size();
  if (out && !c->pending_change.empty()) memcpy(out, c->pending_change.data(), 4 * c->pending_change.size());
  return AM355_OK;
}

int apply_patch_json_impl(am355_ctx* c, const char** json, size_t* len) {
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
int get_dep_graph_impl(am355_ctx* c, const uint32_t** dep_first, const uint32_t** dep_index, uint32_t* n_changes) {
  if (!c) return AM355_E_ARG;
  if (!c->replayed || c->is_document) return fail(c, AM355_E_STATE, "a replayed state of changes is needed");
  (void)hipSetDevice(c->device);
  if (!c->dep_graph_ready) {
    const uint32_t n = c->n_changes;
    const size_t dep_words = std::min(c->raw.size() / 32 + 2, c->d_dep_idx.cap / 4);   // (resident calls grow the arena without touching this table)
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
    // (changes appended by resident calls, replay_resident: their dependencies were resolved on the host)
    const uint32_t nr = std::min(n, c->res_dep_base);
    for (uint32_t i = 0; i < nr; i++) {
      const size_t first = (size_t)((metas[i].base + metas[i].deps_off) >> 5);
      for (uint32_t k = 0; k < metas[i].n_deps; k++) c->dep_index[c->dep_first[i] + k] = di[first + k];
    }
    for (uint32_t i = nr; i < n; i++) {
      const uint32_t j = i - c->res_dep_base;
      if (j + 1 >= c->res_dep_first.size() || c->res_dep_first[j + 1] - c->res_dep_first[j] != metas[i].n_deps) return fail(c, AM355_E_DEVICE, "internal: dependency record of an appended change is missing");
      for (uint32_t k = 0; k < metas[i].n_deps; k++) c->dep_index[c->dep_first[i] + k] = c->res_dep_index[c->res_dep_first[j] + k];
    }
    c->dep_graph_ready = true;
  }
  if (dep_first) *dep_first = c->dep_first.data();
  if (dep_index) *dep_index = c->dep_index.data();
  if (n_changes) *n_changes = c->n_changes;
  return AM355_OK;
}

int sync_bloom_impl(am355_ctx* c, const uint32_t* idx, uint32_t n, uint32_t num_entries, uint32_t bits_per_entry, uint32_t num_probes, const uint8_t* probe_bits,
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
