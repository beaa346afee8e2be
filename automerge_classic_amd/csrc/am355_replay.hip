// am355_replay of the C ABI (include/am355.h): host scheduler (fallback of the device scheduler am355_sched.hip), host half of the plan,
// device buffers, orchestration of the device stages for change logs (replay_impl) and saved documents (replay_document). See am355_ctx.h.
#include "am355_ctx.h"

// ---------------------------------------------------------------------------------------------------------
// host scheduler
// ---------------------------------------------------------------------------------------------------------

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
    const bool lineage = c->in_apply && c->graph_mode != 0 && c->sched_prefix <= n;
    if (lineage) {
      // ---- a lineage that began with Backend.load: the retry loop itself, round by round (see am355_ctx.h graph_mode) ----
      const uint32_t na0 = c->sched_prefix, nd = std::min(c->doc_n_changes, na0);
      auto author_of = [&](uint32_t ci) { return rank[local_ids[local_off[ci]]]; };
      auto group_of = [&](uint32_t ci) { return self[ci] < n ? self[ci] : ci; };
      std::vector<uint8_t> doc_has_dependent(nd, 0);
      for (uint32_t j = 0; j < nd; j++)
        for (uint32_t k = 0; k < dep_count[j]; k++) { uint32_t dd = dep_of(j, k); if (dd < nd) doc_has_dependent[dd] = 1; }
      struct Outcome { std::vector<uint32_t> pass, gpass, gpos; bool graph_after = true; uint32_t flags = 0; };
      auto simulate = [&](bool graph_known, Outcome& o) {
        o.pass.assign(n, NEVER); o.gpass.assign(n, NEVER); o.gpos.assign(n, 0); o.flags = 0;
        std::vector<uint8_t> vis_dep(n, 0), vis_dup(n, 0), in_round(n, 0);  // by group: satisfies a dependency / makes a copy a duplicate
        std::vector<uint64_t> clk(na, 0);
        for (uint32_t i = 0; i < na0; i++) { clk[author_of(i)] = metas[i].seq; o.pass[i] = 0; o.gpass[group_of(i)] = 0; o.gpos[group_of(i)] = i; }
        auto all_prior = [&]() {
          std::fill(vis_dep.begin(), vis_dep.end(), 0); std::fill(vis_dup.begin(), vis_dup.end(), 0);
          for (uint32_t i = 0; i < na0; i++) vis_dep[group_of(i)] = vis_dup[group_of(i)] = 1;
        };
        if (graph_known) all_prior();
        else {
          for (uint32_t i = nd; i < na0; i++) vis_dep[group_of(i)] = vis_dup[group_of(i)] = 1;
          for (uint32_t i = 0; i < nd; i++)
            if (!doc_has_dependent[i]) { vis_dup[i] = 1; vis_dep[i] = c->doc_head_index_known ? 1 : 0; }  // (a head without index: -1, new.js:1727-1729, 1563)
        }
        std::vector<uint32_t> queue, app, enq;
        for (uint32_t ci = na0; ci < n; ci++) queue.push_back(ci);
        bool graph = graph_known;
        uint32_t round_pass = 0;
        while (!queue.empty()) {
          app.clear(); enq.clear();
          std::vector<uint64_t> clk2 = clk;
          bool aborted = false;
          for (uint32_t ci : queue) {
            const uint32_t F = group_of(ci);
            if (vis_dup[F] || in_round[F]) continue;  // (new.js:1566)
            bool ready = true;
            for (uint32_t k = 0; k < dep_count[ci] && ready; k++) { const uint32_t dd = dep_of(ci, k); ready = dd < n && (vis_dep[dd] || in_round[dd]); }
            const uint32_t a = author_of(ci);
            const uint64_t expected = clk2[a] + 1;
            if (!ready) enq.push_back(ci);
            else if (metas[ci].seq < expected) {
              if (graph) { o.flags |= AM355_F_BAD_SEQ; return; }  // "Reuse of sequence number"
              aborted = true;  // (new.js:1581: nothing of this round is applied, the whole queue waits for the hash graph)
              break;
            } else if (metas[ci].seq > expected) { o.flags |= AM355_F_BAD_SEQ; return; }
            else { clk2[a] = metas[ci].seq; in_round[F] = 1; app.push_back(ci); }
          }
          for (uint32_t ci : app) in_round[group_of(ci)] = 0;
          if (aborted) { app.clear(); enq = queue; }
          if (!app.empty()) {
            for (uint32_t ci : app) { const uint32_t F = group_of(ci); o.pass[ci] = round_pass; o.gpass[F] = round_pass; o.gpos[F] = ci; vis_dep[F] = vis_dup[F] = 1; }
            clk = clk2;
            round_pass++;
          }
          queue = enq;
          if (queue.empty()) break;
          if (app.empty()) {
            if (graph) break;
            graph = true;  // computeHashGraph: the index is rebuilt from the document as it was BEFORE this call -- what the call applied so far is not in it
            all_prior();
          }
        }
        o.graph_after = graph;
      };
      Outcome a, b;
      simulate(c->graph_mode != 1, a);
      if (c->graph_mode == 2) {
        simulate(false, b);
        if (a.flags != b.flags || a.pass != b.pass || a.gpass != b.gpass) {
          c->flags |= AM355_F_UNSUPPORTED;
          return fail(c, AM355_E_UNSUPPORTED, "the schedule depends on whether the reference had rebuilt the loaded document's hash graph, which the replayed state does not tell (JS path)");
        }
      }
      if (a.flags) sched_flags |= a.flags;
      pass = a.pass; gpass = a.gpass; gpos = a.gpos;
      c->sched_graph_after = a.graph_after;
    }
    for (uint32_t root = 0; root < n && !lineage; root++) {
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

// The op rows (13 word columns + the insert byte) and the flattened pred lists, for N rows / P preds.
static int carve_cols(am355_ctx* c, uint32_t N, uint32_t P, bool estimate = false) {
  canary_scope("replay buffers (carve_cols: op rows, preds)");
  static const bool no_cap = getenv("AM355_NO_ROW_CAP") != nullptr;   // (A/B: rows carved exactly, as in rounds 1-5; no resident state then)
  if (c->in_apply && !no_cap && !estimate) {   // (estimate: N, P are the speculative launch's upper estimate already, several times the rows)
    // am355_apply_changes: room for the document to grow without the rows moving (replay_resident appends to them)
    N = (uint32_t)std::min<uint64_t>((uint64_t)N + N / 4 + 65536, 0x7ffffff0u);
    P = (uint32_t)std::min<uint64_t>((uint64_t)P + P / 4 + 65536, 0xfffffff0u);
  }
  c->cols_cap_ops = N;
  c->cols_cap_preds = P;
  c->resident_valid = false;   // (the rows are about to be written anew, maybe somewhere else)
  size_t Nc = (size_t)N + 1;
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
  return AM355_OK;
}

// Device buffers for N op rows / P preds, decode, merge, patch IR. `slot_rank` != null: actor tables are the
// device-interned slots (fast path); null: c->amap holds ranks (general path).
// Device buffers for N op rows / P preds (op rows, merge scratch, sort scratch, patch IR), carved from a few arenas.
int setup_buffers(am355_ctx* c, uint32_t NA) {
  uint32_t N = (uint32_t)c->n_ops, P = (uint32_t)c->n_preds;
  int bits_ctr = bits_for64(c->max_op), bits_actor = bits_for64(NA ? NA - 1 : 0), bits_row = bits_for64(N);
  if (1 + bits_row + bits_ctr + bits_actor > 64) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "sort key wider than 64 bits"); }
  size_t Nc = (size_t)N + 1;
  c->resident_valid = false;
  c->order_alt_ptr = nullptr;
  c->pos_valid = false;
  c->ir_stale = false;
  canary_scope("replay buffers (setup_buffers: op rows, preds, merge scratch, sort scratch, patch IR)");
  // (rows the speculative decode launch of this replay is writing keep their place: they are carved for a capacity >= N, P)
  if (!(c->spec_launched && N <= c->spec_cap_ops && P <= c->spec_cap_preds)) {
    c->spec_launched = false;
    int rcc = carve_cols(c, N, P);
    if (rcc) return rcc;
  }
  canary_scope("replay buffers (setup_buffers: merge scratch, sort scratch, patch IR)");
  // am355_apply_changes: the per-row arrays of the merge stage are carved with the stride of the op rows' capacity, so that the kept
  // rows' results stay where they are from call to call (replay_resident); their fills go by N (merge_prepare)
  static const bool no_cap = getenv("AM355_NO_ROW_CAP") != nullptr;
  const bool by_cap = c->in_apply && c->cols_cap_ops >= N && !no_cap;
  // (the op rows of a speculative launch are carved for several times N; the merge arrays take a quarter more than N: ~1.3 x the exact size)
  if (by_cap) Nc = std::min<size_t>(c->cols_cap_ops, (size_t)N + N / 4 + 65536) + 1;
  {
    size_t cw = carry_words((uint32_t)(Nc - 1));
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
    static const bool fill_kernel = getenv("AM355_FILL_KERNEL") != nullptr;   // (A/B: the fused fill launch with exactly carved rows too)
    b.first_row = 0; b.seed_list_inc = 0; b.row_stride = (by_cap || fill_kernel) ? (uint32_t)Nc : 0u;
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
  // (c->spec_launched survived setup_buffers: the rows are where the speculative launch writes them; its gate was open iff this holds)
  const bool spec_done = c->spec_launched && decode_gate_open(tot, c->spec_cap_ops, c->spec_cap_preds, distinct_capacity()) &&
                         c->counts_zeroed_at == c->d_counts.p && c->mb.counts_bytes <= c->counts_zeroed;
  if (trace && c->spec_launched && !spec_done)
    fprintf(stderr, "  planned: speculative decode not usable: ops %u / cap %u, preds %u / cap %u, distinct %u, fast_a %x flags_a %x fallback %u, counts %p/%p %zu/%zu\n", tot.n_ops,
            c->spec_cap_ops, tot.n_preds, c->spec_cap_preds, tot.n_distinct, tot.fast_a, tot.flags_a, tot.fallback, c->counts_zeroed_at, c->d_counts.p, c->mb.counts_bytes, c->counts_zeroed);
  if (c->spec_launched && !spec_done) {
    // the launch did nothing, or wrote rows that have just been carved anew: wait for it, then decode as if it had not been there
    c->spec_launched = false;
    HIPCHK(c, hipStreamSynchronize(c->stream3));
    HIPCHK(c, hipStreamSynchronize(st));
    int rc2 = setup_buffers(c, n_distinct);
    if (rc2) return rc2;
  }
  if (!spec_done && !(c->counts_zeroed_at == c->d_counts.p && c->mb.counts_bytes <= c->counts_zeroed)) HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, c->mb.counts_bytes, st));
  c->counts_zeroed_at = nullptr;  // (one replay's worth: the merge kernels are about to write it)
  c->hint_large = tot.n_large;
  if (spec_done) {
    c->hint_ops = std::max(c->hint_ops, tot.n_ops);
    c->hint_preds = std::max(c->hint_preds, tot.n_preds);
    // the two wave classes are running (small on `st`, large on stream3, which was ordered behind the plan kernel then); what is
    // left for the host to launch is the lane-serial class, beside them on stream3
    const uint32_t host_large = c->spec_large_launched ? 0u : tot.n_large;  // (no large-class change last time: that launch was left out)
    if (tot.n_serial || host_large) {
      if (!c->spec_large_launched) {  // stream3 has not been ordered behind the plan kernel yet
        HIPCHK(c, hipEventRecord(c->ev_fork, st));
        HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
      }
      launch_decode_planned(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), c->d_plans.as<ChangePlan>(), c->d_plans.as<ChangePlan>() + std::max(n, 1u), n, 0,
                            host_large, tot.n_serial, c->d_amap_prov.as<uint32_t>(), c->d_slot_rank.as<uint32_t>(), c->cols, &c->d_counts.as<Counts>()->flags,
                            c->stream3, nullptr, c->shard_rank, c->shard_world);
    }
    lap("decode was launched behind the plan kernel");
  } else {
    c->hint_ops = tot.n_ops;
    c->hint_preds = tot.n_preds;
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
  }
  c->spec_launched = false;  // (used up: a second merge run of this replay -- the general path after a late dependency -- decodes in its own order)
  // The merge stage's fills and, behind them on the SAME stream, the host-built span tables (below): one stream that is busy from here
  // on, one event for k_resolve to wait for. (Rounds 2-4 sent the tables over stream4 alone, which the copy of the digests had used a
  // moment earlier; with the digests written into host memory by the kernels that copy was the stream's first command, and k_resolve
  // waited 60 us for it in most replays -- same-box A/B, profiles/r05_ab_libs.txt.)
  const bool second_class = tot.n_small && (tot.n_large || tot.n_serial);
  // (stream3 carries the second decoder class -- 0.1-0.27 ms for a batch of fat changes --: the fills, which depend on nothing, would
  // start behind it and k_resolve would wait for them; they go to stream4 then)
  hipStream_t fill_stream = second_class ? c->stream4 : c->stream3;
  merge_prepare(c->mb, fill_stream);
  if (second_class) {
    HIPCHK(c, hipEventRecord(c->ev_join, c->stream3));
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
  }
  lap("fills enqueued");
  // ---- host half of the plan, beside the decode kernels (the digests were copied right behind k_plan) ----
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
    // behind the fills, beside the decode kernels (in the main stream the copy would start when the decode kernels end)
    HIPCHK(c, hipMemcpyAsync(d_tables, h, o_tab + b_tab, hipMemcpyHostToDevice, fill_stream));
    HIPCHK(c, hipEventRecord(c->ev_tables, fill_stream));
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
static int replay_document_stages(am355_ctx* c) {
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
    lap("actor ranks enqueued");
    HIPCHK(c, hipEventRecord(c->ev_b0, st));
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_b0, 0));
    KeyStage ks;
    ks.max_jumps = c->key_unbounded ? 0u : 64u;
    canary_forget(c->d_ks.p, c->d_ks.cap);
    keystr_index_begin(c->d_arena.as<uint8_t>(), m.col_off[C_KEY_STR], m.col_len[C_KEY_STR], c->d_ks.p, ks, d_words + W_TOTAL_ENTRIES, d_words + W_FAST_B,
                       c->stream2);
    // (the key stream needs no host decision any more: both halves are enqueued at once and run beside the token index)
    keystr_index_finish(ks, false, &ks_start, &ks_off, &ks_len, d_words + W_FLAGS_B, c->stream2);
    HIPCHK(c, hipEventRecord(c->ev_b1, c->stream2));
    lap("key stream enqueued");
    BigColInfo* hi = c->h_biginfo.as<BigColInfo>();
    bigcol_index_tokens(c->d_arena.as<uint8_t>(), d, w, st);
    HIPCHK(c, hipMemcpyAsync(hi, w.info, sizeof(BigColInfo), hipMemcpyDeviceToHost, st));
    lap("token ends enqueued");
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
    HIPCHK(c, hipMemcpyAsync(c->h_words.as<uint32_t>() + W_FLAGS_B, d_words + W_FLAGS_B, 16, hipMemcpyDeviceToHost, st));  // (+ the key stream's "cut off" word)
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
  if (!c->doc_serial && c->h_words.as<uint32_t>()[W_FLAGS_B + 3] && !c->key_unbounded) {
    // the true parse of the key column reached a literal longer than the continuation walker follows (64 windows: tens of thousands of
    // strings): once more, without the bound (the walker then also follows every garbage "literal" to the end of the column)
    c->key_unbounded = true;
    int rc2 = replay_document_stages(c);
    c->key_unbounded = false;
    return rc2;
  }
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

// The device stages of a document, then the verdict of the checksum thread when am355_backend_load left it running beside them: the
// reference verifies the checksum before anything else (columnar.js:699-705), so a mismatch outranks whatever the stages found.
static int replay_document(am355_ctx* c) {
  int rc = replay_document_stages(c);
  if (rc == AM355_OK && c->prefetch_ir) (void)ir_copy_enqueue(c, true);  // (255 MB for the config-5 document: on its way while the checksum finishes)
  if (c->doc_sum.pending && !c->doc_sum.wait()) {
    c->replayed = false;
    c->staged = false;
    c->flags = AM355_F_BAD_CHECKSUM;
    return fail(c, AM355_E_INVALID, "checksum does not match data");
  }
  return rc;
}

// Backend.load(bytes) in one call (backend.js:104-107): staging with the checksum verdict deferred, the device stages, the IR copy.
int backend_load_impl(am355_ctx* c, const uint8_t* doc, size_t len) {
  int rc = load_document_impl(c, doc, len, true);
  if (rc) return rc;
  c->prefetch_ir = true;
  rc = replay_impl(c);
  c->prefetch_ir = false;
  return rc;
}

// ---------------------------------------------------------------------------------------------------------
// Resident state: Backend.applyChanges onto the state the context holds (new.js:1797-1879 with mergeDocChangeOps :1052-1290 merging
// the batch INTO the stored op set). When the last replay left the op rows, their per-row results and the per-change tables of
// exactly the applied changes in HBM (am355_ctx.h resident_valid), only the BATCH is parsed (host: the device's parser compiled for
// the host, am355_decode.hip parse_changes_host -- a launch, a copy back and the wait for them cost ~25 us, the walk a microsecond),
// hashed (host: SHA extensions, ~2 us per 3 KB change against ~130 us of chain latency on the device), scheduled (host: the
// in-order case of new.js:1550-1597 -- every dependency applied, next sequence number, known actors; anything else falls back to
// the full replay, which has the general scheduler), decoded (the batch's plans) and resolved (k_resolve over the new rows, onto
// the kept accumulators). A batch of plain list edits is then merged into the stored order (am355_resorder.hip); for any other the
// whole-document order / patch tables are rebuilt by the kernels of merge_run as in a full replay. What a call no longer pays is
// stage 1, the decode and the resolution of the old rows -- and, for list edits, the ordering of the old elements.
// Returns AM355_OK, an error, or RESIDENT_FALLBACK: nothing of the state was touched, replay_impl goes on with the full replay.
// ---------------------------------------------------------------------------------------------------------
enum { RESIDENT_FALLBACK = 1 };

static void resident_mark(am355_ctx* c) {
  am355_ctx::ResidentMark& m = c->res_mark;
  const uint32_t lost[8] = {c->d_cols.lost, c->d_pred.lost, c->d_merge.lost, c->d_metas.lost, c->d_hashes.lost, c->d_ir.lost, c->d_sort.lost, c->h_hashes.lost};
  for (int k = 0; k < 8; k++) m.lost[k] = lost[k];
  m.n_changes = c->n_changes; m.n_ops = c->n_ops; m.n_preds = c->n_preds;
}

// none of the buffers the kept state lives in has given its content up since the mark (growing WITH the content -- ensure_keep -- is fine)
static bool resident_mark_holds(const am355_ctx* c) {
  const am355_ctx::ResidentMark& m = c->res_mark;
  const uint32_t lost[8] = {c->d_cols.lost, c->d_pred.lost, c->d_merge.lost, c->d_metas.lost, c->d_hashes.lost, c->d_ir.lost, c->d_sort.lost, c->h_hashes.lost};
  for (int k = 0; k < 8; k++)
    if (m.lost[k] != lost[k]) return false;
  return c->d_cols.p && c->d_merge.p && c->d_metas.p && c->d_hashes.p && c->h_hashes.p;
}

// hash -> change index over c->h_hashes (open addressing, change index + 1; keyed by eight bytes of the hash, verified by full comparison)
static inline size_t hash_slot(const uint8_t* h, size_t mask) { uint64_t v; memcpy(&v, h, 8); return (size_t)((v * 0x9e3779b97f4a7c15ull) >> 20) & mask; }
static void hash_index_add(am355_ctx* c, uint32_t ci) {
  const uint8_t* hs = c->h_hashes.as<uint8_t>();
  const size_t mask = c->hash_index.size() - 1;
  size_t i = hash_slot(hs + 32 * (size_t)ci, mask);
  while (c->hash_index[i]) i = (i + 1) & mask;
  c->hash_index[i] = ci + 1;
}
static uint32_t hash_index_find(const am355_ctx* c, const uint8_t* h) {
  const uint8_t* hs = (const uint8_t*)c->h_hashes.p;
  const size_t mask = c->hash_index.size() - 1;
  for (size_t i = hash_slot(h, mask); c->hash_index[i]; i = (i + 1) & mask)
    if (memcmp(hs + 32 * (size_t)(c->hash_index[i] - 1), h, 32) == 0) return c->hash_index[i] - 1;
  return NONE32;
}

static int replay_resident(am355_ctx* c) {
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "resident: %-30s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  auto fallback = [&](const char* why) {
    c->resident_why = why;
    c->n_resident_fallbacks++;
    if (trace) fprintf(stderr, "resident: full replay instead (%s)\n", why);
    return (int)RESIDENT_FALLBACK;
  };
  // (once changes of the batch have entered the hash index it no longer describes the applied changes: dropped, rebuilt by the next attempt)
  auto fallback_dirty = [&](const char* why) { c->hash_index_n = 0; c->hash_index.clear(); return fallback(why); };
  const uint32_t K = c->keep.n_changes, n = c->n_changes;
  const uint64_t old_ops = c->keep.n_ops, old_preds = c->keep.n_preds;
  if (n <= K) return fallback("empty batch");
  const uint32_t nb = n - K;
  if (!c->resident_valid || !resident_mark_holds(c) || c->res_mark.n_changes != K || c->res_mark.n_ops != old_ops || c->res_mark.n_preds != old_preds)
    return fallback("the context's arrays are not the kept state");
  if (c->shard_world != 1 || c->phase_events || c->graph_mode != 0 || !c->mb.sig) return fallback("mode");
  if (c->d_metas.cap < sizeof(ChangeMeta) * (size_t)n) return fallback("per-change tables full");   // (am355_load_changes grows it with its content)
  if (!c->d_hashes.ensure_keep(32 * (size_t)n, 32 * (size_t)K) || !c->h_hashes.ensure_keep(32 * (size_t)n, 32 * (size_t)K))
    return fail(c, AM355_E_NOMEM, "allocation failed (per-change tables)");
  if (!c->h_res_metas.ensure(sizeof(ChangeMeta) * (size_t)nb)) return fail(c, AM355_E_NOMEM, "host allocation failed");
  hipStream_t st = c->stream;
  const uint8_t* raw = c->raw.data();

  // ---- host: header / column directory / row counts of the batch's changes (the device's own parser, am355_decode.hip parse_change,
  //      compiled for the host: no launch, no copy back, no wait), and their hashes ----
  {
    std::vector<uint32_t> ne(nb);
    parse_changes_host(raw, c->raw_off.data() + K, nb, c->h_res_metas.as<ChangeMeta>(), ne.data());
    { int qrc = queue_upload(c, c->d_metas.as<ChangeMeta>() + K, c->h_res_metas.p, sizeof(ChangeMeta) * (size_t)nb); if (qrc) return qrc; }
  }
  uint8_t* hs = c->h_hashes.as<uint8_t>();
  std::atomic<int> bad_sum{0};
  auto hash_one = [&](uint32_t i) {
    const uint64_t off = c->raw_off[K + i], len = c->raw_off[K + i + 1] - off;
    if (len < 9) { bad_sum.store(1); return; }
    sha256_digest(raw + off + 8, (size_t)len - 8, hs + 32 * (size_t)(K + i));     // columnar.js:693-705: over the chunk without magic + checksum
    if (memcmp(hs + 32 * (size_t)(K + i), raw + off + 4, 4) != 0) bad_sum.store(1);
  };
  // (run BEHIND the enqueue of the batch's device work, see below: ~1 us per change of the calling thread's time that the device spends decoding)
  auto hash_batch = [&]() {
    if (nb >= 32 && c->pool->size() >= 2) {
      const unsigned parts = std::min<unsigned>(c->pool->size() + 1, 16u);
      c->pool->run(parts, [&](unsigned t) { for (uint32_t i = t; i < nb; i += parts) hash_one(i); });
    } else {
      for (uint32_t i = 0; i < nb; i++) hash_one(i);
    }
    return bad_sum.load() == 0;
  };
  lap("batch parsed (host)");
  // the hash index of the applied changes (rebuilt when it does not describe exactly them: after a full replay, a reset, a fallback)
  if (c->hash_index_n != K || c->hash_index.empty() || c->hash_index.size() < 4 * (size_t)n) {
    size_t cap = 64;
    while (cap < 4 * (size_t)n + 64) cap <<= 1;
    c->hash_index.assign(cap, 0);
    for (uint32_t i = 0; i < K; i++) hash_index_add(c, i);
    c->hash_index_n = K;
  }
  const ChangeMeta* metas = c->h_res_metas.as<ChangeMeta>();

  // ---- host: the in-order schedule of the batch, on copies (committed only when every change passes) ----
  const uint32_t NA = (uint32_t)c->actors.size();
  // actor id -> rank without building a std::string per lookup (a change names dozens of other actors: their table is a lookup each)
  auto actor_slot = [](const uint8_t* a, size_t len, size_t mask) {
    uint64_t v = 0;
    memcpy(&v, a, len < 8 ? len : 8);
    return (size_t)(((v ^ len) * 0x9e3779b97f4a7c15ull) >> 24) & mask;
  };
  if (c->res_rank_n != NA || c->res_rank_of.empty()) {
    size_t cap = 64;
    while (cap < 4 * (size_t)NA + 16) cap <<= 1;
    c->res_rank_of.assign(cap, 0);
    for (uint32_t r = 0; r < NA; r++) {
      size_t i = actor_slot((const uint8_t*)c->actors[r].data(), c->actors[r].size(), cap - 1);
      while (c->res_rank_of[i]) i = (i + 1) & (cap - 1);
      c->res_rank_of[i] = r + 1;
    }
    c->res_rank_n = NA;
    c->res_actor_memo.assign(NA, am355_ctx::ActorMemo{});
  }
  auto rank_of = [&](const uint8_t* a, size_t len) -> uint32_t {
    const size_t mask = c->res_rank_of.size() - 1;
    for (size_t i = actor_slot(a, len, mask); c->res_rank_of[i]; i = (i + 1) & mask) {
      const std::string& s = c->actors[c->res_rank_of[i] - 1];
      if (s.size() == len && memcmp(s.data(), a, len) == 0) return c->res_rank_of[i] - 1;
    }
    return NONE32;
  };
  std::vector<uint64_t> clock(NA, 0);
  for (size_t k = 0; k < c->clock_actor.size(); k++) clock[c->clock_actor[k]] = c->clock_seq[k];
  std::vector<uint32_t> new_clock_actors;
  // heads as a mark per change index: the document's heads now, minus what the batch depends on, plus the batch
  std::vector<uint8_t> is_head(n, 0);
  for (size_t k = 0; k + 32 <= c->heads.size(); k += 32) {
    const uint32_t hi = hash_index_find(c, &c->heads[k]);
    if (hi == NONE32) return fallback("a head that is not an applied change");
    is_head[hi] = 1;
  }
  const uint8_t* prev_author_bytes = nullptr;
  uint32_t prev_author_len = 0, prev_author = 0;
  std::vector<ChangePlan> plans;
  std::vector<uint32_t> amap, dep_first(1, 0), dep_index, op_base(nb);
  plans.reserve(nb); dep_first.reserve(nb + 1); dep_index.reserve(2 * (size_t)nb); amap.reserve(4 * (size_t)nb);
  std::vector<std::vector<ActorSpan>> add_spans(NA);
  uint64_t ops = old_ops, preds = old_preds, max_op = c->max_op;
  for (uint32_t i = 0; i < nb; i++) {
    const ChangeMeta& m = metas[i];
    const uint32_t ci = K + i;
    if (m.flags || (m.pad & 1)) return fallback_dirty("a change the parser flags");
    const uint8_t* p = raw + m.base;
    // actor table: author + the others, all known to the document (a new actor changes the ranks of the kept rows: full replay)
    // (a run of changes by one author -- a peer's backlog, a typing session -- looks its rank up once)
    uint32_t author;
    if (prev_author_bytes && prev_author_len == m.actor_len && memcmp(prev_author_bytes, p + m.actor_off, m.actor_len) == 0) author = prev_author;
    else {
      author = rank_of(p + m.actor_off, m.actor_len);
      if (author == NONE32) return fallback_dirty("new actor");
      prev_author_bytes = p + m.actor_off; prev_author_len = m.actor_len; prev_author = author;
    }
    ChangePlan pl{ci, (uint32_t)ops, (uint32_t)preds, (uint32_t)amap.size(), author, 1 + m.n_other};
    if (pl.n_actors != m.n_entries) return fallback_dirty("actor table");
    amap.push_back(author);
    {
      // the table of the other actors: the same bytes as in the author's last change -> the same ranks
      am355_ctx::ActorMemo& memo = c->res_actor_memo[author];
      size_t off = m.others_off, end = off;
      for (uint32_t k = 0; k < m.n_other; k++) {
        // (actor ids are 16 bytes in practice: a one-byte length, no general LEB128 walk)
        if (end < m.len && p[end] < 0x80 && (size_t)p[end] < m.len - end) { end += 1 + (size_t)p[end]; continue; }
        uint64_t l;
        if (!read_uleb_host(p, m.len, end, l) || l > m.len - end) return fallback_dirty("actor table");
        end += (size_t)l;
      }
      const size_t tlen = end - off;
      if (memo.ranks.size() == m.n_other && memo.bytes.size() == tlen && (tlen == 0 || memcmp(memo.bytes.data(), p + off, tlen) == 0)) {
        amap.insert(amap.end(), memo.ranks.begin(), memo.ranks.end());
      } else {
        std::vector<uint32_t> ranks;
        ranks.reserve(m.n_other);
        size_t o = off;
        for (uint32_t k = 0; k < m.n_other; k++) {
          uint64_t l;
          (void)read_uleb_host(p, m.len, o, l);
          const uint32_t rk = l <= m.len - o ? rank_of(p + o, (size_t)l) : NONE32;
          if (rk == NONE32) return fallback_dirty("new actor");
          ranks.push_back(rk);
          o += (size_t)l;
        }
        amap.insert(amap.end(), ranks.begin(), ranks.end());
        memo.bytes.assign(p + off, p + end);
        memo.ranks.swap(ranks);
      }
    }
    if (m.seq != clock[author] + 1) return fallback_dirty("sequence number");
    if (clock[author] == 0) new_clock_actors.push_back(author);
    clock[author] = m.seq;
    op_base[i] = (uint32_t)ops;
    if (m.n_ops) {
      // the change's op ids lie behind every id of its author so far (ascending, disjoint spans: what plan_fast verifies)
      const uint32_t a0 = c->actor_tab_off[author], a1 = c->actor_tab_off[author + 1];
      uint64_t last_end = a1 > a0 ? (uint64_t)c->spans[a1 - 1].start_op + c->spans[a1 - 1].n_ops : 0;
      if (!add_spans[author].empty()) last_end = (uint64_t)add_spans[author].back().start_op + add_spans[author].back().n_ops;
      if (m.start_op < last_end || m.start_op + m.n_ops > 0xfffffff0ull) return fallback_dirty("op id range");
      add_spans[author].push_back(ActorSpan{(uint32_t)m.start_op, m.n_ops, (uint32_t)ops});
      max_op = std::max<uint64_t>(max_op, m.start_op + m.n_ops - 1);
      plans.push_back(pl);
    }
    ops += m.n_ops;
    preds += m.n_preds;
    if (ops >= 0x7ffffff0ull || preds >= 0xfffffff0ull) return fallback_dirty("size");
  }
  const uint32_t N = (uint32_t)ops, P = (uint32_t)preds;
  if (N > c->cols_cap_ops || P > c->cols_cap_preds || (size_t)N + 1 > c->mb.row_stride) return fallback_dirty("row capacity");
  const int bits_ctr = bits_for64(max_op), bits_actor = bits_for64(NA ? NA - 1 : 0), bits_row = bits_for64(N);
  if (1 + bits_row + bits_ctr + bits_actor > 64) return fallback_dirty("sort key width");
  lap("batch scheduled (host)");

  // ---- the batch's hashes, and what depends on them: not a duplicate, every dependency an applied change (a change of this batch in
  //      front counts), the heads. Run BEHIND the enqueue of the batch's device work (decode, resolution, list order): SHA-256 of a
  //      3 KB change is ~1.5 us of the calling thread -- 40 changes 60 us with the lookups -- which the device spends on the batch anyway.
  //      A batch that fails here has changed the kept arrays: the full replay that follows starts from the staged bytes, as it does
  //      after any failure. Returns the reason, nullptr when the batch passes. ----
  auto hashes_and_dependencies = [&]() -> const char* {
    if (!hash_batch()) return "checksum";
    const uint8_t* prev_deps = nullptr;   // the dependency block of the change in front (a round of synced peers shares it): resolved once
    uint32_t prev_n_deps = 0, prev_first = 0;
    for (uint32_t i = 0; i < nb; i++) {
      const ChangeMeta& m = metas[i];
      const uint32_t ci = K + i;
      const uint8_t* p = raw + m.base;
      if (hash_index_find(c, hs + 32 * (size_t)ci) != NONE32) return "duplicate change";
      const uint8_t* deps = p + m.deps_off;
      if (prev_deps && prev_n_deps == m.n_deps && m.n_deps && memcmp(prev_deps, deps, 32 * (size_t)m.n_deps) == 0) {
        for (uint32_t k = 0; k < m.n_deps; k++) dep_index.push_back(dep_index[prev_first + k]);   // (their head marks are already down)
      } else {
        for (uint32_t k = 0; k < m.n_deps; k++) {
          const uint32_t di = hash_index_find(c, deps + 32 * (size_t)k);
          if (di == NONE32) return "dependency not applied yet";
          dep_index.push_back(di);
          is_head[di] = 0;
        }
      }
      prev_deps = deps; prev_n_deps = m.n_deps; prev_first = dep_first.back();
      dep_first.push_back((uint32_t)dep_index.size());
      is_head[ci] = 1;
      hash_index_add(c, ci);   // (undone by dropping the index when a later change fails)
    }
    return nullptr;
  };
  // the per-actor op-id spans with the batch's (what the device tables hold from this call on; the context's copy follows with the commit)
  std::vector<ActorSpan> spans_new;
  std::vector<uint32_t> tab_new(NA + 1, 0);
  spans_new.reserve(c->spans.size() + plans.size());
  for (uint32_t a = 0; a < NA; a++) {
    tab_new[a] = (uint32_t)spans_new.size();
    spans_new.insert(spans_new.end(), c->spans.begin() + c->actor_tab_off[a], c->spans.begin() + c->actor_tab_off[a + 1]);
    spans_new.insert(spans_new.end(), add_spans[a].begin(), add_spans[a].end());
  }
  tab_new[NA] = (uint32_t)spans_new.size();
  std::vector<ChangePlan> plans_dev = plans;   // (in the decoder's class order below)
  // ---- commit the host state (once the batch has passed every check) ----
  auto commit_host_state = [&]() {
    c->hash_index_n = n;
    for (uint32_t a : new_clock_actors) c->clock_actor.push_back(a);
    c->clock_seq.clear();
    for (uint32_t a : c->clock_actor) c->clock_seq.push_back(clock[a]);
    {
      std::vector<const uint8_t*> hv;
      for (uint32_t i = 0; i < n; i++)
        if (is_head[i]) hv.push_back(hs + 32 * (size_t)i);
      std::sort(hv.begin(), hv.end(), [](const uint8_t* x, const uint8_t* y) { return memcmp(x, y, 32) < 0; });
      std::vector<uint8_t> heads_new(hv.size() * 32);
      for (size_t k = 0; k < hv.size(); k++) memcpy(&heads_new[32 * k], hv[k], 32);
      c->heads.swap(heads_new);
    }
    c->spans.swap(spans_new);
    c->actor_tab_off.swap(tab_new);
    for (uint32_t i = 0; i < nb; i++) { c->applied_change.push_back(K + i); c->applied_op_base.push_back(op_base[i]); }
    if (c->res_dep_first.empty()) c->res_dep_first.assign(1, 0);
    for (uint32_t i = 0; i < nb; i++) {
      c->res_dep_index.insert(c->res_dep_index.end(), dep_index.begin() + dep_first[i], dep_index.begin() + dep_first[i + 1]);
      c->res_dep_first.push_back((uint32_t)c->res_dep_index.size());
    }
    c->n_applied = n; c->n_pending = 0;
    c->pending_change.clear();
    c->pass_first_row.clear();
    c->n_ops = N; c->n_preds = P; c->max_op = max_op;
    c->has_unknown_cols = false;
    c->plans.swap(plans_dev);
    c->amap = amap;
  };

  // ---- device: tables, the batch's rows, their resolution, then the whole-document order / patch tables ----
  const size_t np = plans.size();
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_plans = sizeof(ChangePlan) * np, b_spans = sizeof(ActorSpan) * spans_new.size(), b_tab = 4 * tab_new.size(), b_amap = 4 * amap.size();
  const size_t o_spans = al(b_plans + 16), o_tab = o_spans + al(b_spans + 16), o_x = o_tab + al(b_tab + 16), tables_bytes = o_x + al(b_amap + 16);
  if (!c->d_tables.ensure(tables_bytes) || !c->h_stage.ensure(tables_bytes)) return fail(c, AM355_E_NOMEM, "device allocation failed");
  uint8_t* d_tables = c->d_tables.as<uint8_t>();
  c->p_spans = (ActorSpan*)(d_tables + o_spans);
  c->p_tab_off = (uint32_t*)(d_tables + o_tab);
  c->sig_seq++;
  uint32_t n_small = 0, n_large = 0;
  {
    std::vector<ChangePlan> large, serial;
    size_t w = 0;
    for (size_t i = 0; i < np; i++) {
      const int wc = change_wave_class(metas[plans_dev[i].change - K]);
      if (wc == 2) plans_dev[w++] = plans_dev[i];
      else if (wc == 1) large.push_back(plans_dev[i]);
      else serial.push_back(plans_dev[i]);
    }
    n_small = (uint32_t)w;
    n_large = (uint32_t)large.size();
    for (auto& pl : large) plans_dev[w++] = pl;
    for (auto& pl : serial) plans_dev[w++] = pl;
  }
  {
    uint8_t* h = c->h_stage.as<uint8_t>();
    if (b_plans) memcpy(h, plans_dev.data(), b_plans);
    if (b_spans) memcpy(h + o_spans, spans_new.data(), b_spans);
    memcpy(h + o_tab, tab_new.data(), b_tab);
    if (b_amap) memcpy(h + o_x, amap.data(), b_amap);
    { int qrc = queue_upload(c, d_tables, h, o_x + b_amap); if (qrc) return qrc; }
  }
  MergeBufs& b = c->mb;
  b.arena = c->d_arena.as<uint8_t>();
  b.ops = c->cols;
  b.n_ops = N; b.n_preds = P; b.n_actors = NA;
  b.sig = c->h_sig.as<HostSignals>(); b.sig_seq = c->sig_seq;
  b.actor_tab_off = c->p_tab_off; b.spans = c->p_spans;
  b.bits_ctr = (uint32_t)bits_ctr; b.bits_actor = (uint32_t)bits_actor;
  b.first_row = (uint32_t)old_ops; b.seed_list_inc = c->seed_list_inc;   // (b.row_stride: as the last full replay carved the arrays)
  if (!c->d_counts.ensure(merge_counts_bytes(N))) return fail(c, AM355_E_NOMEM, "device allocation failed (merge)");
  merge_bind_counts(b, c->d_counts.p);
  // ---- list order: a batch of plain list edits is merged into the STORED order (am355_resorder.hip); anything else -- map rows, new
  //      objects, a new element with two new children -- orders every list anew with the kernels of merge_run. Its buffers are bound here:
  //      their clears ride with the fill of the new rows' accumulators ----
  static const bool no_resorder = getenv("AM355_NO_RESORDER") != nullptr;
  const uint32_t NN = N - (uint32_t)old_ops, NL_old = c->counts.n_list_ins, NO = c->counts.n_objects;
  const bool try_resorder = !no_resorder && NN && NN <= resorder_chunk_rows() * (resorder_chunk_rows() < RESORDER_ROWS_MAX ? 64u : RESORDER_CHUNKS_MAX) &&   // (tests' small chunks: many of them)
                             NL_old && c->mb.row_stride;
  ResOrderBufs ro{};
  if (try_resorder) {
    const size_t cap_rows = c->mb.row_stride;
    // (the order lives in one of two arrays of row_stride + 2 words -- the one carved with the merge arrays and c->d_order_alt --, and
    // every in-place merge writes the other one; a new carve, setup_buffers, starts over)
    if (!c->order_alt_ptr) {
      if (!c->d_order_alt.ensure(4 * (cap_rows + 2))) return fail(c, AM355_E_NOMEM, "device allocation failed (resident list order)");
      c->order_alt_ptr = c->d_order_alt.as<uint32_t>();
    }
    if (!c->d_pos.ensure_keep(4 * cap_rows, c->pos_valid ? 4 * (size_t)old_ops : 0) || !c->d_resorder.ensure(resorder_bytes(NN, NO)) || !c->h_resorder.ensure(64))
      return fail(c, AM355_E_NOMEM, "device allocation failed (resident list order)");
    resorder_bind(ro, c->d_resorder.p, NN, NO);
    canary_arm();   // (AM355_CANARY=1 only)
    ro.T0 = (uint32_t)old_ops; ro.n_new = NN; ro.n_list = NL_old; ro.n_obj = NO;
    ro.pos_of = c->d_pos.as<uint32_t>();
    ro.order_new = c->order_alt_ptr;
    ro.sig = b.sig; ro.sig_seq = b.sig_seq;
    ro.allow_maps = getenv("AM355_NO_MAPS_ONLY") ? 0u : 1u;   // (tests, A/B: read per call)
  }
  c->resident_valid = false;   // (from here on the kept arrays change: a failure leaves no state behind)
  { int frc = flush_uploads(c); if (frc) return frc; }   // (the batch's bytes, its records, the tables, the delta stage's breaks: one launch; its hashes follow)
  {
    FillRanges extra;
    extra.add(c->d_counts.p, b.counts_bytes, 0);
    if (try_resorder) extra.add(ro.obj_add, (size_t)((uint8_t*)(ro.words + 8) - (uint8_t*)ro.obj_add), 0);   // obj_add | words (neighbours in the block, resorder_bind)
    merge_prepare(b, st, MERGE_FILL_ROWS, &extra);   // (the new rows' accumulators; in this stream: the decode of a small batch is too short to hide a second stream's join)
  }
  // (launch_decode_columns puts a class on the second stream only when the small class and the lane-serial one are both there and no
  // large one: a handful of changes otherwise decode in ONE launch on `st`, and the fork / join would be four runtime calls for nothing)
  const uint32_t n_serial = (uint32_t)np - n_small - n_large;
  const bool second_stream = n_small && n_serial && !(n_large && n_small + n_large <= 1024);
  if (second_stream) {
    HIPCHK(c, hipEventRecord(c->ev_fork, st));
    HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
  }
  launch_decode_columns(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), (const ChangePlan*)d_tables, n_small, n_large, n_serial,
                        (const uint32_t*)(d_tables + o_x), nullptr, c->cols, &c->d_counts.as<Counts>()->flags, st, second_stream ? c->stream3 : nullptr, 0, 1);
  if (second_stream) {
    HIPCHK(c, hipEventRecord(c->ev_join, c->stream3));
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
  }
  lap("decode enqueued");
  Counts* hc = c->h_counts.as<Counts>();
  merge_resolve(b, st);
  bool merged_in_place = false, maps_only = false, final_in_new = true;
  if (try_resorder) {
    if (!c->pos_valid) resorder_positions(b, NL_old, ro.pos_of, st);
    resorder_run(b, ro, st, &final_in_new);
  }
  lap("resolution / list order enqueued");
  // ---- host, while the device works on the batch: hashes, duplicates, dependencies; then the commit ----
  if (const char* why = hashes_and_dependencies()) {
    (void)hipStreamSynchronize(st);   // (the copy kernel reads the pinned arena; the kernels behind it signal into this call's words)
    c->staging_in_flight = false;
    c->pos_valid = false;
    return fallback_dirty(why);
  }
  commit_host_state();
  { int qrc = queue_upload(c, c->d_hashes.as<uint8_t>() + 32 * (size_t)K, hs + 32 * (size_t)K, 32 * (size_t)nb); if (!qrc) qrc = flush_uploads(c); if (qrc) return qrc; }
  lap("batch hashed, dependencies checked, committed (host)");
  if (try_resorder) {
    // its verdict and the flags of the resolution: signalled into pinned words by a launch behind it (two copy dispatches and their wait otherwise)
    uint32_t* hw = c->h_resorder.as<uint32_t>();
    if (wait_host_signal(&b.sig->resorder_seq, b.sig_seq, st)) {
      memcpy(hw, (const void*)b.sig->resorder, 36);
      c->staging_in_flight = false;   // (the kernel that signalled ran behind everything that read the pinned arena)
    } else {
      HIPCHK(c, hipMemcpyAsync(hw, ro.words, 32, hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipMemcpyAsync(hw + 8, &b.counts->flags, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipStreamSynchronize(st));
    }
    if (hw[8]) return error_for_flags(c, hw[8], "op set rejected");
    if (hw[0] == 0) {
      // the order after the batch is the state's order from here on: in the other buffer after an odd number of chunks
      if (final_in_new) {
        uint32_t* old_order = b.order;
        b.order = ro.order_new;
        c->order_alt_ptr = old_order;   // (the previous order array is what the next in-place merge writes)
      }
      c->counts.n_list_ins = hw[2] + hw[1];
      c->pos_valid = true;
      c->ir_stale = true;
      c->ir_fetched = false;
      // the object and map tables are what they were (the batch has list rows only and makes no object): a host copy of them that was
      // current before the call still is, and no copy is enqueued for the call's patch; the edit table it may have come with is not
      c->h_tables_current = c->h_tables_were_current;
      c->ir_copy_enqueued = c->h_tables_current ? 1 : 0;
      c->hir.edits = nullptr;
      c->n_resorder_calls++;
      c->batch_list_only = true;
      merged_in_place = true;
      lap("list order merged in place");
      if (hw[4]) {
        // plain map rows beside the list edits (text typed and a key assigned in one change): the map half of the merge behind the
        // in-place list merge -- the map records and the object table's map ranges change, the delta stage runs its map kernels
        merge_run_maps(b, c->ir, hc, st);
        lap("map half of the merge done");
        if (hc->flags) return error_for_flags(c, hc->flags, "op set rejected");
        c->counts.n_map_emit = hc->n_map_emit;
        c->counts.max_key_len = hc->max_key_len;
        c->h_tables_current = false;
        c->ir_copy_enqueued = 0;
        c->batch_list_only = false;
        c->n_maps_only_calls++;
      }
    } else if (hw[3] == 0 && !getenv("AM355_NO_MAPS_ONLY")) {
      // a batch of plain map rows (`set` / `del` on string keys): no list changes -- the stored order, positions and element counts stay,
      // the map half of the merge runs alone (visibility, object table, map records in patch order); the whole-document edit tables are
      // stale from here on, as after an in-place list merge
      merge_run_maps(b, c->ir, hc, st);
      lap("map half of the merge done");
      if (hc->flags) return error_for_flags(c, hc->flags, "op set rejected");
      c->counts.n_map_emit = hc->n_map_emit;
      c->counts.max_key_len = hc->max_key_len;
      c->ir_stale = true;
      c->ir_fetched = false;
      c->n_maps_only_calls++;
      c->pos_valid = true;   // (the positions this call read or rebuilt: the order did not change)
      maps_only = true;
    } else {
      c->pos_valid = false;
      lap("list order: not a batch for the in-place merge");
    }
  }
  if (!merged_in_place && !maps_only) {
    merge_prepare(b, st, MERGE_FILL_TABLES);
    merge_run(b, c->ir, hc, st, nullptr, c->ev_runs, true);
    lap("merge_run done");
    if (hc->flags) return error_for_flags(c, hc->flags, "op set rejected");
    c->counts = *hc;
    c->counts.n_objects += 1;  // + _root
    c->pos_valid = false;
    c->ir_stale = false;
  }
  c->n_resident_calls++;
  return AM355_OK;
}

// The whole-document patch tables of a state whose last calls merged their list edits in place (replay_resident: c->ir_stale): the
// kernels of merge_run from k_emit on, over the rows as they stand. AM355_RESORDER_VERIFY=1: the order they compute must be the one
// the in-place merges left (tests).
int ensure_ir_fresh(am355_ctx* c) {
  if (!c || !c->ir_stale) return AM355_OK;
  (void)hipSetDevice(c->device);
  hipStream_t st = c->stream;
  MergeBufs& b = c->mb;
  const uint32_t N = b.n_ops;
  std::vector<uint32_t> kept;
  const bool verify = getenv("AM355_RESORDER_VERIFY") != nullptr;
  if (verify) {
    kept.resize(c->counts.n_list_ins);
    HIPCHK(c, hipMemcpy(kept.data(), b.order, 4 * kept.size(), hipMemcpyDeviceToHost));
  }
  c->sig_seq++;
  b.sig_seq = c->sig_seq;
  b.first_row = N;                      // (nothing to resolve: every row has been)
  b.seed_list_inc = c->seed_list_inc;
  if (!c->d_counts.ensure(merge_counts_bytes(N))) return fail(c, AM355_E_NOMEM, "device allocation failed (merge)");
  merge_bind_counts(b, c->d_counts.p);
  HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, b.counts_bytes, st));
  merge_prepare(b, st);
  Counts* hc = c->h_counts.as<Counts>();
  merge_run(b, c->ir, hc, st, nullptr, c->ev_runs);
  if (hc->flags) { c->staged = c->replayed = false; return error_for_flags(c, hc->flags, "op set rejected"); }
  c->counts = *hc;
  c->counts.n_objects += 1;
  c->ir_stale = false;
  c->ir_fetched = false;
  c->ir_copy_enqueued = 0;
  c->h_tables_current = false;
  c->pos_valid = false;
  c->stats.n_edits = c->counts.n_edits;
  c->stats.ir_bytes = (uint64_t)c->counts.n_objects * sizeof(am355_ir_object) + (uint64_t)c->counts.n_map_emit * sizeof(am355_ir_map) +
                      ((uint64_t)c->counts.n_erecs + 1) * sizeof(am355_ir_edit);
  if (verify) {
    std::vector<uint32_t> now(c->counts.n_list_ins);
    HIPCHK(c, hipMemcpy(now.data(), b.order, 4 * now.size(), hipMemcpyDeviceToHost));
    if (now != kept) return fail(c, AM355_E_DEVICE, "internal: the list order merged in place differs from the order computed from scratch");
  }
  return AM355_OK;
}

int replay_impl(am355_ctx* c) {
  if (!c) return AM355_E_ARG;
  if (!c->staged) return fail(c, AM355_E_STATE, "am355_load_changes must be called first");
  (void)hipSetDevice(c->device);
  c->replayed = c->ir_fetched = false;
  c->ir_copy_enqueued = 0;
  c->dep_graph_ready = false;
  c->flags = 0;
  c->spec_launched = false;
  c->batch_list_only = false;
  c->h_tables_were_current = c->h_tables_current;   // (of the state before this replay: replay_resident may find them unchanged)
  c->h_tables_current = false;
  if (c->is_document) return replay_document(c);
  auto t_begin = std::chrono::steady_clock::now();
  if (c->in_apply && c->keep.want) {
    // Backend.applyChanges onto the state this context holds: the batch alone (replay_resident), unless it needs the general path
    c->keep.want = false;
    const int rr = replay_resident(c);
    if (rr != RESIDENT_FALLBACK) {
      if (rr != AM355_OK) return rr;
      am355_stats& s = c->stats;
      s.n_changes = c->n_changes; s.n_applied = c->n_applied; s.n_pending = 0; s.n_actors = (uint32_t)c->actors.size(); s.n_objects = c->counts.n_objects;
      s.n_heads = (uint32_t)(c->heads.size() / 32); s.n_ops = c->n_ops; s.max_op = c->max_op; s.raw_bytes = c->raw.size();
      s.n_map_values = c->counts.n_map_emit; s.n_list_elems = c->counts.n_list_ins; s.n_edits = c->counts.n_edits;
      s.ir_bytes = (uint64_t)c->counts.n_objects * sizeof(am355_ir_object) + (uint64_t)c->counts.n_map_emit * sizeof(am355_ir_map) +
                   ((uint64_t)c->counts.n_erecs + 1) * sizeof(am355_ir_edit);
      // (no wait here: what the call goes on with -- the delta stage, a fetch -- is enqueued behind the batch on the same stream)
      s.ms_parse = s.ms_decode = s.ms_merge = s.ms_order = s.ms_hash_stream = s.ms_host_schedule = 0;
      s.fast_path = 1;
      s.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
      c->used_fast_path = true;
      c->device_scheduled = false;
      c->replayed = true;
      c->seed_list_inc = c->counts.n_list_inc;
      resident_mark(c);
      c->resident_valid = true;
      return AM355_OK;
    }
  }
  c->keep.want = false;
  { int orc = flush_uploads(c); if (!orc) orc = upload_offsets(c); if (orc) return orc; }
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
  // (more than 4 KB per change on average: the parse kernel with the wavefront-parallel row / pred counts, k_parse_changes<true>; the
  // headline log's 3.2 KB changes keep the lean kernel -- 20 us against 36 with the bigger LDS footprint, profiles/r05_kernel_table_fat_parse.txt)
  const bool fat_changes = n && c->raw.size() / n > 4096 && !getenv("AM355_PARSE_LEAN");
  if (c->inline_fills) {
    // (cleared by the parse kernel's workgroups on their way in: no second stream, no event wait in front of the next kernel)
    ParseFills f{};
    auto add = [&](void* q, size_t bytes, uint32_t v) { f.p[f.n] = (uint32_t*)q; f.n_words[f.n] = (bytes + 3) / 4; f.value[f.n] = v; f.n++; };
    add(d_words, 4 * W_NUM, 0);
    add(d_wa, s1_distinct + 16, 0);
    add(c->d_slots.p, 8 * (size_t)(c->slot_mask + 1), 0);
    add(c->d_first_idx.p, 4 * (size_t)(c->slot_mask + 1), 0xffffffffu);
    if (counts_too) add(c->d_counts.p, cb, 0);
    launch_parse_changes(c->d_arena.as<uint8_t>(), c->d_offsets.as<uint64_t>(), n, c->d_metas.as<ChangeMeta>(), c->d_entries.as<uint32_t>(), f, sa, fat_changes);
    HIPCHK(c, hipEventRecord(c->ev_parse, sa));
  } else {
    launch_parse_changes(c->d_arena.as<uint8_t>(), c->d_offsets.as<uint64_t>(), n, c->d_metas.as<ChangeMeta>(), c->d_entries.as<uint32_t>(), ParseFills{}, sa, fat_changes);
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
  // AM355_HASH_ENQUEUE=main: by the calling thread, behind the stage-1 launches (rounds 2-4). Default: by the helper thread, beside them.
  static const bool enqueue_thread = AM355_STREAMS_ORDER_ACROSS_THREADS && []() { const char* e = getenv("AM355_HASH_ENQUEUE"); return !e || !strcmp(e, "thread"); }();
  std::atomic<int> lane_rc{AM355_OK};
  struct LaneJoin {   // (whatever way this function is left, the helper is not enqueueing into the context any more)
    AsyncLane* l = nullptr;
    ~LaneJoin() { if (l) l->wait(); }
  } lane_join;
  bool b_enqueued = false;
  if (hash_after_parse && enqueue_early) { int rb = enqueue_stream_b(); if (rb) return rb; b_enqueued = true; }
  else if (hash_after_parse && enqueue_thread && n >= 512) {
    if (!c->lane) c->lane.reset(new AsyncLane);
    c->lane->post([&]() { (void)hipSetDevice(c->device); lane_rc.store(enqueue_stream_b()); });
    lane_join.l = c->lane.get();
    b_enqueued = true;
  }
  exclusive_scan_u32(c->d_entries.as<uint32_t>(), c->d_amap_base.as<uint32_t>(), n, d_wa + W_TOTAL_ENTRIES, c->d_scan1.p, sa);
  // (AM355_SPEC_DECODE=0: the decode kernels wait for the host to read the totals, as in rounds 2-4)
  static const bool spec_env = []() { const char* e = getenv("AM355_SPEC_DECODE"); return !(e && *e == '0'); }();
  const bool spec_possible = spec_env && n && c->inline_fills && counts_too && hash_after_parse && !getenv("AM355_HOST_PLAN") && !(c->in_apply && c->graph_mode != 0) &&
                             c->d_plan_totals.ensure(sizeof(PlanTotals));
  for (int attempt = 0;; attempt++) {
    if (attempt) {
      HIPCHK(c, hipMemsetAsync(c->d_slots.p, 0, 8 * (size_t)(c->slot_mask + 1), sa));
      HIPCHK(c, hipMemsetAsync(c->d_first_idx.p, 0xff, 4 * (size_t)(c->slot_mask + 1), sa));
      HIPCHK(c, hipMemsetAsync(d_distinct, 0, 4, sa));
    }
    launch_actor_intern(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), n, c->d_amap_base.as<uint32_t>(), c->d_amap_prov.as<uint32_t>(), c->amap_cap,
                        c->d_slots.as<unsigned long long>(), c->slot_mask, c->d_first_idx.as<uint32_t>(), d_wa + W_FLAGS_A, d_wa + W_FAST_A,
                        d_distinct, c->d_rank_ids.p, d_briefs, c->d_slot_rank.as<uint32_t>(), c->d_plan_sums.as<unsigned long long>(), d_wa + 8, sa, c->hp_briefs);
    // device half of the in-order plan (actor ranks, per-change bases, decoder classes): the decode kernels start from it
    // (its totals, and the stage-1 words the host decides on, reach the host through HostSignals: no copy, no blocking wait)
    c->sig_seq++;
    launch_plan(d_briefs, n, d_distinct, c->d_slot_rank.as<uint32_t>(), c->slot_mask, c->d_plan_sums.as<unsigned long long>(), c->d_plans.as<ChangePlan>(),
                c->d_plans.as<ChangePlan>() + n1, d_wa, d_wa + 8, sig, c->sig_seq, sa, spec_possible ? c->d_plan_totals.as<PlanTotals>() : nullptr,
                c->h_s1.as<uint32_t>());
    // the host's own half of the plan needs a 32-byte digest per change and the handful of distinct actor ids: k_actor_check and
    // k_plan_apply write them into the host's pinned mirror themselves (rounds 2-4: an event record in this stream, a copy on
    // stream4 and a blocking wait for it -- the wait was the longest item between the plan and k_resolve)
    const bool want_large_spec = spec_possible && attempt == 0 && c->hint_large != 0;
    if (want_large_spec) HIPCHK(c, hipEventRecord(c->ev_plan, sa));  // (the large class decodes on stream3, ordered behind the plan kernel)
    if (spec_possible && attempt == 0) {
      // the decode kernels of the wave classes right behind the plan kernel, before the host knows the totals (decode_gate_open,
      // am355_internal.h): rows carved for a capacity -- the context's previous in-order replay, or one row per four encoded bytes
      const uint64_t est = c->raw.size() / 2 + 1024;
      c->spec_cap_ops = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(c->hint_ops, est), 0x7ffffff0u);
      c->spec_cap_preds = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(c->hint_preds, est), 0x7ffffff0u);
      // (the merge stage's counter block, cleared by the parse kernel above, must not move when the real totals arrive)
      if (trace) fprintf(stderr, "replay: speculative decode: cap %u ops / %u preds (counter block %zu of %zu bytes)\n", c->spec_cap_ops, c->spec_cap_preds, merge_counts_bytes(c->spec_cap_ops), cb);
      if (merge_counts_bytes(c->spec_cap_ops) <= cb && carve_cols(c, c->spec_cap_ops, c->spec_cap_preds, true) == AM355_OK) {
        canary_arm();
        if (c->phase_events) { HIPCHK(c, hipEventRecord(c->ev[1], sa)); HIPCHK(c, hipEventRecord(c->ev[2], sa)); }
        if (want_large_spec) HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_plan, 0));
        launch_decode_speculative(c->d_arena.as<uint8_t>(), c->d_metas.as<ChangeMeta>(), c->d_plans.as<ChangePlan>(), n, c->d_plan_totals.as<PlanTotals>(), c->spec_cap_ops,
                                  c->spec_cap_preds, distinct_capacity(), c->d_amap_prov.as<uint32_t>(), c->d_slot_rank.as<uint32_t>(), c->cols,
                                  &c->d_counts.as<Counts>()->flags, sa, c->stream3, want_large_spec, c->shard_rank, c->shard_world);
        c->spec_large_launched = want_large_spec;
        if (c->phase_events) HIPCHK(c, hipEventRecord(c->ev[3], sa));
        c->spec_launched = true;
      }
    }
    if ((c->phase_events && !c->spec_launched) || !hash_after_parse) HIPCHK(c, hipEventRecord(c->ev[1], sa));
    if (attempt == 0 && !b_enqueued) { int rb = enqueue_stream_b(); if (rb) return rb; b_enqueued = true; }
    lap("stage 1 enqueued");
    if (!wait_host_signal(&sig->plan_seq, c->sig_seq, sa)) {
      (void)hipStreamSynchronize(sb);
      (void)hipStreamSynchronize(c->stream3);
      return fail(c, AM355_E_DEVICE, "the device did not report the plan of this replay (%s)", hipGetErrorString(hipGetLastError()));
    }
    memcpy(&tot, (const void*)&sig->plan, sizeof tot);
    lap("stage 1 totals read");
    if (!(tot.fast_a & FF_CAPACITY) || attempt) break;
    c->spec_launched = false;  // (the decode kernels behind the first plan saw its capacity flag and did nothing; the second plan gets an ordinary launch)
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
  const bool lineage_schedule = c->in_apply && c->graph_mode != 0;  // (a lineage that began with Backend.load: the host's round-by-round scheduler, see schedule())
  if (lineage_schedule) fast = false;
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
    // (measured and taken out again, profiles/r05_ab_libs.txt: starting the copy of the record tables to the host right here, beside the
    // join of the hash stream, instead of in am355_fetch_ir -- no gain for the host-to-host time within the spread of the runs, and a
    // caller that replays without fetching pays 0.1 ms for 5 MB it did not ask for)
    lap("run_device (device plan) done");
    opt_flags = c->flags;
    opt_err = c->err;
  } else {
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
  if (lane_join.l) { lane_join.l->wait(); lane_join.l = nullptr; if (lane_rc.load()) return lane_rc.load(); }
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
    c->ir_copy_enqueued = 0;   // (tables of a discarded optimistic run may be on their way: what am355_fetch_ir hands out is enqueued behind the new ones)
    // the device's actor tables, when its list of distinct ids holds them all (else the host interns)
    const bool dev_actors = tot.n_distinct <= distinct_capacity() && !(tot.fast_a & FF_CAPACITY) && tot.total_entries <= c->amap_cap;
    bool served = false;
    const bool host_schedule = getenv("AM355_HOST_SCHEDULE") != nullptr || lineage_schedule;  // (A/B and tests: the host's scheduler for every batch)
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
  // what a later am355_apply_changes may keep (replay_resident): every staged change applied, in staged order, rows carved for a capacity
  c->seed_list_inc = c->counts.n_list_inc;
  c->res_dep_base = n;
  c->res_dep_first.clear();
  c->res_dep_index.clear();
  c->hash_index_n = 0;
  c->res_rank_of.clear();
  c->res_rank_n = 0;
  c->res_actor_memo.clear();
  c->resident_valid = fast && c->in_apply && c->shard_world == 1 && c->mb.row_stride != 0 && !c->has_unknown_cols;
  if (c->resident_valid) resident_mark(c);
  if (!c->in_apply) {  // (one call of Backend.loadChanges: its scheduling passes are the op streams)
    c->stream_breaks = c->pass_first_row;
    c->breaks_exact = true;
    c->children_hazard = false;
    c->no_history = false;
    c->doc_rows_known = false;
    c->graph_mode = 0;
  }
  return AM355_OK;
}

