// Incremental patch of Backend.applyChanges(state, changes) as data-parallel kernels.
//
// The reference builds that patch op by op while it merges the batch into its document (backend/new.js:1052-1290
// mergeDocChangeOps calling :884-1040 updatePatchProperty in its incremental mode, :747-782 appendEdit, :1461-1528 setupPatches):
// list edits carry the index the element had AT THE MOMENT the op was applied, map properties list what is visible after the last
// op of the batch that touched the key.  Here the whole log (earlier changes + the batch) has just been replayed, so the final
// state is known -- every element's position in its list (MergeBufs.order), every row's successor count -- and rows are numbered in
// application order, which makes "time" a row number: rows >= T0 are the batch.  The patch follows from the final state and time:
//
//   list object   insert edit for every new element e at time t(e) = its row; for every new row that deletes from or assigns to an
//                 element an edit at its time: remove (the element was visible and is not any more), insert (the other way round),
//                 update (it stays visible; weight 0).  index(x, t) = visible elements in front of x at time t
//                   = V0(x)                                  elements in front that were visible before the batch (prefix sum)
//                   + #{new elements in front of x inserted before t} - #{elements in front of x removed before t}
//                 -- a dominance count over (position, time), done for all edits at once by stable binary partitions on the bits of
//                 the time (most significant first) of the edits laid out in position order: at every level an edit whose bit is 1
//                 adds the (signed) number of edits of its group with bit 0 in front of it.  The partitions END with the edits of
//                 each object in time order, which is the order of the reference's `edits` array; runs of consecutive inserts
//                 become multi-insert records and runs of removes at one index one record (appendEdit, new.js:754-777).
//   map object    for every key a new row names: its visible values in op id order (`props[key] = {}` when none is left).  The one
//                 order-dependent rule of the merge loop that shows in a patch is reproduced: when the call that handled the
//                 key's last op went on to a greater key of the same object (new.js:1125-1129), the document ops of the key with
//                 a greater id than that last op are never looked at (new.js:1140-1149 keeps the stale `changeOp`), and their
//                 values are missing from the patch.
//   object links  am355_calls.hip / am355_apply.cpp (host): setupPatches over the object table, from ObjLink.
//
// Served subset (anything else raises F_UNSUPPORTED and the call is served by the JS path): list elements hold plain values -- inserted,
// deleted and assigned to (`list[i] = v`: kd_events; the edit rewriting of appendUpdate and the index lag inside one merge call are
// restated in kd_edit_runs / kd_events) --; elements that hold child objects or counters are inserted and deleted but not assigned to;
// no objectId sharding.
#include "am355_delta.h"
#include "am355_prims.h"
#include "am355_rows.h"
#include "am355_scan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace am355 {

// why a call is refused (DeltaCounts.reason keeps the smallest code raised: the host names it in the error text)
__device__ __forceinline__ uint32_t refuse(const DeltaBufs& d, uint32_t reason) {
  atomicMin(&d.counts->reason, reason);
  return (uint32_t)F_UNSUPPORTED;
}

static inline dim3 dgrid(uint32_t n) { return dim3((n + BLOCK - 1) / BLOCK); }
static size_t al256(size_t b) { return carve_round(b); }

static uint32_t key_table_cap(uint32_t n_new) {
  uint32_t cap = 64;
  while (cap < 2 * (uint64_t)n_new + 16) cap <<= 1;
  return cap;
}

size_t delta_bytes(uint32_t N, uint32_t NN, uint32_t NM, uint32_t NO, uint32_t NL) {
  size_t cap = key_table_cap(NN);
  size_t b = al256(sizeof(DeltaCounts)) + al256(sizeof(ObjLink) * ((size_t)NO + 1));
  b += 5 * al256(4 * ((size_t)N + 1)) + 3 * al256(4 * ((size_t)N + 2));
  b += 5 * al256(4 * ((size_t)NL + 2));
  b += 3 * al256(4 * ((size_t)NN + 1));
  b += 10 * al256(4 * ((size_t)NN + 2)) + 4 * al256(4 * ((size_t)NN + 3)) + 6 * al256(4 * ((size_t)NN + 2)) + al256(sizeof(am355_ir_edit) * ((size_t)NN + 2));
  b += 9 * al256(4 * (cap + 1)) + al256(8 * cap);
  b += 3 * al256(4 * ((size_t)NM + 1)) + 2 * al256(8 * ((size_t)NM + cap + 1)) + 2 * al256(4 * ((size_t)NM + cap + 1)) + al256(sizeof(am355_ir_map) * ((size_t)NM + cap + 1));
  size_t biggest = std::max<size_t>({(size_t)N + 2, (size_t)NM + cap + 2, (size_t)NN + 4});
  b += al256(scan_workspace_bytes((uint32_t)biggest)) + al256(sort_workspace_bytes((uint32_t)((size_t)NM + cap + 1)));
  return b + 4096;
}

template <class T>
static T* dcarve(uint8_t*& p, size_t count) {
  T* r = (T*)p;
  canary_note(p, count * sizeof(T));
  p += al256(count * sizeof(T));
  return r;
}

void delta_bind(DeltaBufs& d, void* block, uint32_t N, uint32_t NN, uint32_t NM, uint32_t NO, uint32_t NL) {
  canary_scope("delta stage (delta_bind)");
  canary_forget(block, delta_bytes(N, NN, NM, NO, NL));  // (the layout moves with the sizes: zones of the last call lie inside this call's arrays)
  uint8_t* p = (uint8_t*)block;
  size_t cap = key_table_cap(NN);
  d.key_mask = (uint32_t)cap - 1;
  d.counts = dcarve<DeltaCounts>(p, 1);
  d.link = dcarve<ObjLink>(p, (size_t)NO + 1);
  // (first_del | first_kill are filled with ones, new_succ | upd_n | upd_cur cleared: neighbours, one fill each -- delta_run)
  d.first_del = dcarve<uint32_t>(p, (size_t)N + 1); d.first_kill = dcarve<uint32_t>(p, (size_t)N + 1);
  d.new_succ = dcarve<uint32_t>(p, (size_t)N + 1); d.upd_n = dcarve<uint32_t>(p, (size_t)N + 1); d.upd_cur = dcarve<uint32_t>(p, (size_t)N + 2);
  d.pos_of = dcarve<uint32_t>(p, (size_t)N + 1);
  d.upd_off = dcarve<uint32_t>(p, (size_t)N + 2); d.upd_rows = dcarve<uint32_t>(p, (size_t)N + 2);
  canary_allow(d.first_del, (size_t)((uint8_t*)(d.first_kill + N + 1) - (uint8_t*)d.first_del));
  canary_allow(d.new_succ, (size_t)((uint8_t*)(d.upd_cur + N + 2) - (uint8_t*)d.new_succ));
  d.ev_kind = dcarve<uint32_t>(p, (size_t)NN + 1); d.ev_before = dcarve<uint32_t>(p, (size_t)NN + 1); d.ev_nafter = dcarve<uint32_t>(p, (size_t)NN + 1);
  d.v0 = dcarve<uint32_t>(p, (size_t)NL + 2); d.icnt = dcarve<uint32_t>(p, (size_t)NL + 2);
  d.v0_ex = dcarve<uint32_t>(p, (size_t)NL + 2); d.item_ex = dcarve<uint32_t>(p, (size_t)NL + 2); d.icur = dcarve<uint32_t>(p, (size_t)NL + 2);
  for (int k = 0; k < 2; k++) {
    d.tk[k] = dcarve<uint32_t>(p, (size_t)NN + 2); d.elem[k] = dcarve<uint32_t>(p, (size_t)NN + 2); d.acc[k] = dcarve<uint32_t>(p, (size_t)NN + 2);
    d.lo[k] = dcarve<uint32_t>(p, (size_t)NN + 2); d.hi[k] = dcarve<uint32_t>(p, (size_t)NN + 2);
  }
  d.zf = dcarve<uint32_t>(p, (size_t)NN + 3); d.zw = dcarve<uint32_t>(p, (size_t)NN + 3);
  d.zf_ex = dcarve<uint32_t>(p, (size_t)NN + 3); d.zw_ex = dcarve<uint32_t>(p, (size_t)NN + 3);
  d.e_index = dcarve<uint32_t>(p, (size_t)NN + 2); d.e_flags = dcarve<uint32_t>(p, (size_t)NN + 2);
  d.e_head = dcarve<uint32_t>(p, (size_t)NN + 2); d.e_head_ex = dcarve<uint32_t>(p, (size_t)NN + 2);
  d.e_val = dcarve<uint32_t>(p, (size_t)NN + 2); d.e_val_ex = dcarve<uint32_t>(p, (size_t)NN + 2);
  d.edit = dcarve<am355_ir_edit>(p, (size_t)NN + 2);
  d.edit_cap = NN + 2;
  // (slot_rep | slot_last | slot_cnt | slot_child | slot_drop are cleared by one fill: neighbours)
  d.slot_rep = dcarve<uint32_t>(p, cap + 1); d.slot_last = dcarve<uint32_t>(p, cap + 1); d.slot_cnt = dcarve<uint32_t>(p, cap + 1);
  d.slot_child = dcarve<uint32_t>(p, cap + 1); d.slot_drop = dcarve<uint32_t>(p, cap + 1);
  canary_allow(d.slot_rep, (size_t)((uint8_t*)(d.slot_drop + cap + 1) - (uint8_t*)d.slot_rep));
  d.slot_first = dcarve<uint32_t>(p, cap + 1); d.slot_cont = dcarve<uint32_t>(p, cap + 1);
  d.place = dcarve<uint32_t>(p, cap + 1); d.place_ex = dcarve<uint32_t>(p, cap + 1);
  d.slot_L = dcarve<unsigned long long>(p, cap);
  d.keep = dcarve<uint32_t>(p, (size_t)NM + 1); d.keep_ex = dcarve<uint32_t>(p, (size_t)NM + 1); d.rec_slot = dcarve<uint32_t>(p, (size_t)NM + 1);
  for (int k = 0; k < 2; k++) { d.pair_key[k] = dcarve<uint64_t>(p, (size_t)NM + cap + 1); d.pair_val[k] = dcarve<uint32_t>(p, (size_t)NM + cap + 1); }
  d.map = dcarve<am355_ir_map>(p, (size_t)NM + cap + 1);
  size_t biggest = std::max<size_t>({(size_t)N + 2, (size_t)NM + cap + 2, (size_t)NN + 4});
  d.scan_ws = p;
  canary_note(p, scan_workspace_bytes((uint32_t)biggest));
  p += al256(scan_workspace_bytes((uint32_t)biggest));
  d.sort_ws = p;
  canary_note(p, sort_workspace_bytes((uint32_t)((size_t)NM + cap + 1)));
}

// ---------------------------------------------------------------------------------------------------------
// objects: parent link of every object (setupPatches walks them on the host, new.js:1461-1528)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void kd_objects(MergeBufs b, PatchIR ir, DeltaBufs d) {
  uint32_t oi = gtid();
  if (oi >= d.n_obj) return;
  ObjLink L{NONE32, 0, 0, 0, 0, 0, NONE32, 0, 0, 0, 0, 0};
  if (oi > 0) {
    const OpCols& o = b.ops;
    uint32_t m = ir.obj[oi].make_row;
    L.parent = obj_index_of(b, b.obj_row[m]);
    if (b.succ_cnt[m] == 0) L.flags |= OL_VISIBLE;
    if (o.key_len[m] != NONE32) { L.key_off = o.key_off[m]; L.key_len = o.key_len[m]; }
    else {
      uint32_t el = o.insert[m] ? m : b.ref_row[m];
      L.flags |= OL_LIST_PARENT;
      if (el != NONE32) { L.elem_ctr = o.id_ctr[el]; L.elem_actor = o.id_actor[el]; if (el >= d.T0) L.flags |= OL_ELEM_NEW; }
    }
  }
  d.link[oi] = L;
}

// ---------------------------------------------------------------------------------------------------------
// rows: which elements hold assignments; per new row: the object it touches, the map key it touches, the element it deletes
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t key_hash(const MergeBufs& b, uint32_t g) {
  const uint8_t* p = b.arena + b.ops.key_off[g];
  uint32_t len = b.ops.key_len[g];
  unsigned long long h = 0xcbf29ce484222325ull ^ b.obj_row[g];
  for (uint32_t k = 0; k < len; k++) h = (h ^ p[k]) * 0x100000001b3ull;
  // (FNV-1a leaves the last bytes of the key in the low bits only: keys that differ in their last characters -- "k0001", "k0002" -- would
  // share a few slots of the table and probe linearly through each other; measured 1.4 ms per 80 k rows. Finish with an avalanche.)
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  return (uint32_t)h;
}

__device__ __forceinline__ uint32_t key_slot(const MergeBufs& b, const DeltaBufs& d, uint32_t g, bool insert) {
  uint32_t i = key_hash(b, g) & d.key_mask;
  for (uint32_t probes = 0; probes <= d.key_mask; probes++) {
    uint32_t v = d.slot_rep[i];
    if (v == 0) {
      if (!insert) return NONE32;
      v = atomicCAS(&d.slot_rep[i], 0u, g + 1);
      if (v == 0) return i;
    }
    uint32_t r = v - 1;
    if (b.obj_row[r] == b.obj_row[g] && same_key(b, r, g)) return i;
    i = (i + 1) & d.key_mask;
  }
  return NONE32;
}

// first new row that names each object. A batch names few objects -- often ONE, and a million atomics on one word execute one
// after another (~12 ns each: 12 ms for the headline log). The lanes of a wavefront that name the same object as its first live lane
// share one atomic (the lowest row of a wavefront is its lowest lane), and a wavefront whose rows come after the recorded minimum
// issues none.  (Its own kernel: folded into kd_rows, the compiler -- ROCm 7.2 -- left the row number of the map-key lanes in
// an undefined register pair.)
__global__ __launch_bounds__(BLOCK) void kd_touch(MergeBufs b, DeltaBufs d) {
  uint32_t g = d.T0 + gtid();
  uint8_t kind = g < b.n_ops ? b.kind[g] : (uint8_t)K_NONE;
  const bool live = kind != K_NONE && kind != K_FOREIGN;
  uint32_t oi = live ? obj_index_of(b, b.obj_row[g]) : NONE32;
  unsigned long long m = __ballot(live);
  uint32_t lane = threadIdx.x & (WAVE - 1);
  uint32_t leader = m ? (uint32_t)__ffsll(m) - 1 : 0;
  uint32_t loi = __shfl(oi, (int)leader);
  if (live && (lane == leader || oi != loi)) {
    uint32_t t = g - d.T0;
    if (d.link[oi].touch > t) atomicMin(&d.link[oi].touch, t);
  }
}

// The counter an increment feeds: among its preds the counter `set` with the greatest id (counterStates[succOp] = counterState, the
// later assignment wins: new.js:944-950; k_resolve counts the increment for that row in inc_cnt / inc_sum). NONE32: none.
__device__ __forceinline__ uint32_t inc_counter_of(const MergeBufs& b, uint32_t g) {
  const OpCols& o = b.ops;
  uint32_t fed = NONE32;
  unsigned long long best = 0;
  for (uint32_t k = 0; k < o.pred_num[g]; k++) {
    const uint32_t pa = o.pred_actor[o.pred_first[g] + k], pc = o.pred_ctr[o.pred_first[g] + k];
    const uint32_t r = row_of(b, pa, pc);
    if (r == NONE32 || r >= g || o.action[r] != 1 || (o.val_tl[r] & 15) != 8) continue;
    const unsigned long long id = pack_id(pc, pa);
    if (fed == NONE32 || id > best) { fed = r; best = id; }
  }
  return fed;
}

__global__ __launch_bounds__(BLOCK) void kd_rows(MergeBufs b, DeltaBufs d) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  uint8_t kind = b.kind[g];
  if (kind == K_LIST_UPD) {
    uint32_t el = b.ref_row[g];
    if (el != NONE32) atomicAdd(&d.upd_n[el], 1u);
  }
  if (g < d.T0) return;
  const OpCols& o = b.ops;
  uint32_t err = 0;
  if (kind == K_FOREIGN) err |= refuse(d, DR_FOREIGN_ROW);
  // a new list row that is neither a value, a child object, a deletion nor an increment of a counter (a link, an unknown action): not
  // restated here (the whole-document patch serves them, k_quirk_rows)
  if (kind == K_LIST_INS || kind == K_LIST_INS_VIS || kind == K_LIST_UPD) {
    const uint32_t a = o.action[g];
    if (!(a == 1 || ((a & 1u) == 0 && a < 7) || (a == 5 && kind == K_LIST_UPD))) err |= refuse(d, DR_ELEM_NOT_PLAIN);  // (increments: kd_events)
  }
  if (kind != K_NONE && kind != K_FOREIGN) {
    if (kind == K_MAP || (kind == K_DEL && o.key_len[g] != NONE32)) {
      uint32_t s = key_slot(b, d, g, true);
      if (s == NONE32) err |= refuse(d, DR_KEY_TABLE);
      else { atomicMin(&d.slot_first[s], g); atomicMax(&d.slot_last[s], g); }
    } else if (kind == K_DEL || kind == K_LIST_UPD) {
      // every value row this op overwrites or deletes: how many rows of the batch do so, and which is the first. An increment does
      // not take the counter it feeds away (first_kill: the first successor that does); first_del / new_succ count every successor,
      // as the reference's succNum does
      if (b.ref_row[g] == NONE32) err |= F_BAD_ELEM;
      const uint32_t fed = o.action[g] == 5 ? inc_counter_of(b, g) : NONE32;
      for (uint32_t k = 0; k < o.pred_num[g]; k++) {
        uint32_t r = row_of(b, o.pred_actor[o.pred_first[g] + k], o.pred_ctr[o.pred_first[g] + k]);
        if (r == NONE32 || r >= g) { err |= F_BAD_ELEM; continue; }
        atomicAdd(&d.new_succ[r], 1u);
        atomicMin(&d.first_del[r], g);
        if (r != fed) atomicMin(&d.first_kill[r], g);
      }
    }
  }
  if (err) atomicOr(&d.counts->flags, err);
}

__global__ __launch_bounds__(BLOCK) void kd_upd_scatter(MergeBufs b, DeltaBufs d) {
  uint32_t g = gtid();
  if (g >= b.n_ops || b.kind[g] != K_LIST_UPD) return;
  uint32_t el = b.ref_row[g];
  if (el != NONE32) d.upd_rows[d.upd_off[el] + atomicAdd(&d.upd_cur[el], 1u)] = g;
}

// ---------------------------------------------------------------------------------------------------------
// lists
// ---------------------------------------------------------------------------------------------------------
// A list element shows the values of its insert row and of the rows that assign to it (K_LIST_UPD), as far as they have no
// successor. With rows numbered in application order that is a question of time: a value row r exists from time r on, and is
// overwritten at the time of its first successor -- before the batch if it has successors among the earlier rows, else at
// first_del[r].
__device__ __forceinline__ bool alive_at_T0(const MergeBufs& b, const DeltaBufs& d, uint32_t r) { return r >= d.T0 || b.succ_cnt[r] == d.new_succ[r]; }
constexpr uint32_t ELEM_ROWS_MAX = 1u << 16;  // value rows of one element this stage walks, one thread per op on it (more: refused)

// the element's value rows visible just before / just after row g was applied
__device__ __forceinline__ void elem_state(const MergeBufs& b, const DeltaBufs& d, uint32_t e, uint32_t g, uint32_t& before, uint32_t& after) {
  before = after = 0;
  const uint32_t nu = d.upd_n[e], base = d.upd_off[e];
  for (uint32_t k = 0; k <= nu; k++) {
    uint32_t r = k == 0 ? e : d.upd_rows[base + k - 1];
    if (r > g || !alive_at_T0(b, d, r)) continue;
    uint32_t dies = d.first_del[r];
    if (r < g && dies >= g) before++;
    if (dies > g) after++;
  }
}

// ---- elements that hold counters (new.js:937-965) ----------------------------------------------------------------------------
// A counter `set` stays a value of its element while every successor it has is an increment that feeds it; it then shows its
// total, and stands among the element's values where its LAST increment stands (the reference lists it when it visits that row).
// Times are row numbers: "at tau" = rows < tau, as they stand when row tau is about to be applied.
constexpr uint32_t QUIRK_ROWS_MAX = 64;  // rows of an element with increments this stage walks (quadratic: more are refused)

__device__ __forceinline__ bool is_value_action(uint32_t a) { return a == 1 || ((a & 1u) == 0 && a < 7); }
__device__ __forceinline__ uint32_t elem_row(const DeltaBufs& d, uint32_t e, uint32_t q) { return q == 0 ? e : d.upd_rows[d.upd_off[e] + q - 1]; }

// does the element hold an increment (at any time)?  (its rows: the insert row and the K_LIST_UPD rows, kd_upd_scatter)
__device__ __forceinline__ bool elem_has_inc(const MergeBufs& b, const DeltaBufs& d, uint32_t e) {
  const uint32_t nu = d.upd_n[e];
  for (uint32_t q = 0; q <= nu; q++)
    if (b.ops.action[elem_row(d, e, q)] == 5) return true;
  return false;
}

// value row r of element e at tau: still a value?  *key: where it stands among the element's values (op id; a counter: the id of its
// last increment before tau), *total: a counter's total at tau (valid when *is_total)
__device__ bool quirk_value_at(const MergeBufs& b, const DeltaBufs& d, uint32_t e, uint32_t r, uint32_t tau, unsigned long long* key, long long* total,
                               bool* is_total) {
  const OpCols& o = b.ops;
  if (r >= tau || !is_value_action(o.action[r])) return false;
  if (key) *key = pack_id(o.id_ctr[r], o.id_actor[r]);
  if (is_total) *is_total = false;
  if (o.action[r] != 1 || (o.val_tl[r] & 15) != 8 || b.succ_cnt[r] == 0) return alive_at_T0(b, d, r) && d.first_del[r] >= tau;
  // a counter with successors: the increments that feed it, old and new (the element's rows hold them all)
  const uint32_t nu = d.upd_n[e];
  uint32_t new_incs = 0, incs_before = 0;
  long long sum = 0;
  unsigned long long last = 0;
  for (uint32_t q = 1; q <= nu; q++) {
    const uint32_t r2 = elem_row(d, e, q);
    if (o.action[r2] != 5 || inc_counter_of(b, r2) != r) continue;
    if (r2 >= d.T0) new_incs++;
    if (r2 < tau) {
      incs_before++;
      long long v = 0;
      (void)int_value(b, r2, v);
      sum += v;
      const unsigned long long id = pack_id(o.id_ctr[r2], o.id_actor[r2]);
      last = id > last ? id : last;
    }
  }
  // successors before the batch that are not its increments: all of them, less the increments, less what the batch adds
  const uint32_t old_other = (b.succ_cnt[r] - d.new_succ[r]) - (b.inc_cnt[r] - new_incs);
  if (old_other != 0 || d.first_kill[r] < tau) return false;
  if (incs_before) {
    if (key && last > *key) *key = last;
    if (total) { long long base = 0; (void)int_value(b, r, base); *total = base + sum; }
    if (is_total) *is_total = true;
  }
  return true;
}

// values the element shows at tau, whether any of its rows is without successor at tau (what the reference counts as a visible
// element, new.js:1626 / seekToOp), whether it holds a row this stage does not model (a link, an unknown action)
__device__ void quirk_elem_state(const MergeBufs& b, const DeltaBufs& d, uint32_t e, uint32_t tau, uint32_t& n_vals, bool& raw_vis, bool& weird) {
  n_vals = 0; raw_vis = false; weird = false;
  const uint32_t nu = d.upd_n[e];
  for (uint32_t q = 0; q <= nu; q++) {
    const uint32_t r = elem_row(d, e, q);
    const uint32_t a = b.ops.action[r];
    if (a != 5 && !is_value_action(a)) weird = true;
    if (r >= tau) continue;
    if (alive_at_T0(b, d, r) && d.first_del[r] >= tau) raw_vis = true;
    if (quirk_value_at(b, d, e, r, tau, nullptr, nullptr, nullptr)) n_vals++;
  }
}

// per list position: was the element visible before the batch; one edit item for a new element
__global__ __launch_bounds__(BLOCK) void kd_positions(MergeBufs b, DeltaBufs d) {
  uint32_t p = gtid();
  if (p > d.n_list) return;
  uint32_t v0 = 0, c = 0;
  if (p < d.n_list) {
    uint32_t e = b.order[p];
    d.pos_of[e] = p;
    bool is_new = e >= d.T0;
    if (!is_new) {
      const uint32_t nu = d.upd_n[e], base = d.upd_off[e];
      bool vis = alive_at_T0(b, d, e);
      for (uint32_t k = 0; k < nu && !vis; k++) {
        uint32_t r = d.upd_rows[base + k];
        vis = r < d.T0 && alive_at_T0(b, d, r);
      }
      v0 = vis ? 1u : 0u;
    }
    c = is_new ? 1u : 0u;
  }
  d.v0[p] = v0;
  d.icnt[p] = c;
}

__device__ __forceinline__ bool list_elem_op(const MergeBufs& b, uint32_t g) {
  uint8_t k = b.kind[g];
  return k == K_LIST_UPD || (k == K_DEL && b.ops.key_len[g] == NONE32);
}

// per new row that deletes from or assigns to a list element: the edit it gives rise to (new.js:984-1033 taken over all the
// rows of the element the merge call visits: insert when nothing was visible and something is, remove the other way round, update
// when the element stays visible). Served: elements whose rows are plain `set`s.
//
// The index lag of the reference: a merge call takes the following ops of the pass -- the ops of all changes one scheduling pass
// applies are one stream -- while they are by the same actor and continue on the NEXT element of the document (new.js:1111-1123),
// and it reports the insert row of such a next element at an index that does not yet count the previous element
// (updatePatchProperty at new.js:1203 runs before the increment at :1236-1239). It shows when the previous element is visible after
// its op and the insert row of this one held a visible value: `list[1] = x` followed by the same actor's deletion of element 2
// gives update@1, remove@1. Reproduced for removes (ev_lag); an update whose first value would sit at the lagging index and the
// others not is refused.
constexpr uint32_t GAP_WALK_MAX = 4096;  // later insertions between two elements this stage walks over to decide whether they were neighbours

// Several ops of one change on ONE list element share a merge call -- and one visit of the element, one edit -- unless a later one
// overwrites an earlier one of them (new.js:1092-1118): `del x; set x` arrives as one `update`, not as remove + insert. Such a RUN of
// consecutive rows reports through its last row, with the element's state before its first row and after its last.
constexpr uint32_t RUN_ROWS_MAX = 64;  // rows of one run this stage walks (more: refused)

__device__ __forceinline__ bool first_row_of_pass(const DeltaBufs& d, uint32_t g) {
  bool first = g == d.T0;
  for (uint32_t k = 0; k < d.n_pass; k++) first = first || d.pass_rows[k] == g;
  return first;
}
// row x could continue the merge call of row x - 1 on the same element (same stream, actor, object, element)
__device__ __forceinline__ bool same_elem_follows(const MergeBufs& b, const DeltaBufs& d, uint32_t x) {
  if (x <= d.T0 || x >= b.n_ops || first_row_of_pass(d, x)) return false;
  if (!list_elem_op(b, x) || !list_elem_op(b, x - 1)) return false;
  const OpCols& o = b.ops;
  return o.id_actor[x - 1] == o.id_actor[x] && same_obj(b, x - 1, x) && b.ref_row[x] != NONE32 && b.ref_row[x] == b.ref_row[x - 1];
}
// does row x name one of the rows [s, x) as pred?
__device__ __forceinline__ bool overwrites_one_of(const MergeBufs& b, uint32_t x, uint32_t s) {
  const OpCols& o = b.ops;
  for (uint32_t k = 0; k < o.pred_num[x]; k++) {
    const uint32_t pc = o.pred_ctr[o.pred_first[x] + k], pa = o.pred_actor[o.pred_first[x] + k];
    for (uint32_t r = s; r < x; r++)
      if (o.id_ctr[r] == pc && o.id_actor[r] == pa) return true;
  }
  return false;
}

__global__ __launch_bounds__(BLOCK) void kd_events(MergeBufs b, DeltaBufs d) {
  uint32_t t = gtid();
  if (t >= d.n_new) return;
  uint32_t g = d.T0 + t;
  uint32_t ev = EV_NONE, before = 0, after = 0, err = 0, lag = 0;
  if (list_elem_op(b, g) && b.ref_row[g] != NONE32) {
    const OpCols& o = b.ops;
    uint32_t e = b.ref_row[g];
    const uint32_t nu = d.upd_n[e];
    const bool quirk = nu <= ELEM_ROWS_MAX && elem_has_inc(b, d, e);  // the element holds increments: counter rules (quirk_elem_state)
    // ---- the run of the merge call this row belongs to: [g0, g] so far; does row g + 1 go on with it? ----
    uint32_t chain = g, steps = 0;
    while (same_elem_follows(b, d, chain) && steps <= RUN_ROWS_MAX) { chain--; steps++; }
    uint32_t g0 = chain;
    for (uint32_t x = chain + 1; x <= g; x++)
      if (overwrites_one_of(b, x, g0)) g0 = x;  // (a row that overwrites one of the call's rows starts a call of its own, new.js:1094-1101)
    const bool goes_on = same_elem_follows(b, d, g + 1) && !overwrites_one_of(b, g + 1, g0);
    if (steps > RUN_ROWS_MAX) err |= refuse(d, DR_SAME_ELEM_CALL);
    if (nu > ELEM_ROWS_MAX || (quirk && nu + 1 > QUIRK_ROWS_MAX)) err |= refuse(d, DR_ELEM_ROWS);
    else if (!goes_on) {
      if (quirk) {
        // Served when the two notions of "the element is visible" agree before and after the op -- the reference counts an element as
        // visible when any of its rows has no successor (list indexes), and reports insert / remove / update by the values it lists;
        // they part when a deleted counter still has increments, or when every increment of a counter has been deleted (a `remove`
        // edit for an element that stays, a value of an element that does not count: left to the JS path)
        bool raw_b, raw_a, weird_b, weird_a;
        quirk_elem_state(b, d, e, g0, before, raw_b, weird_b);
        quirk_elem_state(b, d, e, g + 1, after, raw_a, weird_a);
        if (weird_b || weird_a || raw_b != (before != 0) || raw_a != (after != 0)) err |= refuse(d, DR_ELEM_NOT_PLAIN);
      } else {
        uint32_t unused;
        elem_state(b, d, e, g, unused, after);
        if (g0 == g) before = unused;
        else elem_state(b, d, e, g0, before, unused);
      }
      ev = before ? (after ? EV_UPDATE : EV_REMOVE) : (after ? EV_INSERT : EV_NONE);
      if (nu > 0 && !quirk) {  // (the element holds assignment rows, old or new)
        // values are `set` rows and make rows (child objects); anything else (link, unknown actions) is not restated
        bool plain = is_value_action(o.action[e]);
        for (uint32_t k = 0; k < nu; k++) plain = plain && is_value_action(o.action[d.upd_rows[d.upd_off[e] + k]]);
        if (!plain) err |= refuse(d, DR_ELEM_NOT_PLAIN);
      }
      // ---- does the call continue the merge call of the previous op of the stream (on another element)? ----
      if (!first_row_of_pass(d, g0) && list_elem_op(b, g0 - 1) && o.id_actor[g0 - 1] == o.id_actor[g0] && same_obj(b, g0 - 1, g0) && b.ref_row[g0 - 1] != NONE32) {
        uint32_t a = b.ref_row[g0 - 1];
        if (a == e) {
          // (the same element: row g0 overwrites a row of the call in front -- a call of its own, no index lag)
        } else if (d.upd_n[a] <= ELEM_ROWS_MAX) {
          uint32_t a_before = 0, a_after = 0;
          if (elem_has_inc(b, d, a)) {
            // (an element with increments: its values after row g0 - 1; the thread of that row has checked that they say the same as
            // the reference's own count of visible elements)
            bool raw_vis, weird;
            if (d.upd_n[a] + 1 > QUIRK_ROWS_MAX) err |= refuse(d, DR_ELEM_ROWS);
            else quirk_elem_state(b, d, a, g0, a_after, raw_vis, weird);
          } else elem_state(b, d, a, g0 - 1, a_before, a_after);
          const bool ins_row_visible = e < g0 && alive_at_T0(b, d, e) && d.first_del[e] >= g0;  // the insert row held a visible value
          if (a_after > 0 && ins_row_visible) {
            // was e the element right behind a when the op was applied? (elements inserted later may stand between them now)
            uint32_t pa = d.pos_of[a], pe = d.pos_of[e];
            bool neighbours = pe > pa;
            if (neighbours && pe - pa - 1 > GAP_WALK_MAX) { neighbours = false; err |= refuse(d, DR_GAP_WALK); }
            for (uint32_t q = pa + 1; neighbours && q < pe; q++) neighbours = b.order[q] > g0;
            if (neighbours) {
              if (ev == EV_REMOVE) lag = 1;
              else if (d.first_del[e] > g) err |= refuse(d, DR_LAGGING_UPDATE);  // the insert row's value stays: its update edit alone would lag
            }
          }
        } else err |= refuse(d, DR_ELEM_ROWS);
      }
      if (ev != EV_NONE) atomicAdd(&d.icnt[d.pos_of[e]], 1u);
    }
  }
  d.ev_kind[t] = ev;
  d.ev_before[t] = (before ? 1u : 0u) | lag << 1;
  d.ev_nafter[t] = after;
  // an update (or an element that comes back) writes one edit record per visible value: the records beyond one per item (rare: conflicts)
  if (after > 1 && (ev == EV_UPDATE || ev == EV_INSERT)) atomicAdd(&d.counts->rec_extra, after - 1);
  if (err) atomicOr(&d.counts->flags, err);
}

// items: tk = time << 2 | kind (0 insert: +1 visible element, 1 remove: -1, 2 update: 0), element, group = the items of its list object
__device__ __forceinline__ void put_item(const MergeBufs& b, DeltaBufs& d, uint32_t e, uint32_t tk) {
  uint32_t p = d.pos_of[e];
  uint32_t oi = obj_index_of(b, b.obj_row[e]);
  uint32_t fp = b.obj_first_pos[oi];
  uint32_t lo = d.item_ex[fp], hi = d.item_ex[fp + b.obj_n[oi]];
  uint32_t at = d.item_ex[p] + atomicAdd(&d.icur[p], 1u);
  d.tk[0][at] = tk; d.elem[0][at] = e; d.acc[0][at] = 0; d.lo[0][at] = lo; d.hi[0][at] = hi;
}

__global__ __launch_bounds__(BLOCK) void kd_items(MergeBufs b, DeltaBufs d) {
  uint32_t t = gtid();
  if (t >= d.n_new) return;
  uint32_t g = d.T0 + t;
  uint8_t kind = b.kind[g];
  if (kind == K_LIST_INS || kind == K_LIST_INS_VIS) put_item(b, d, g, t << 2);
  uint32_t ev = d.ev_kind[t];
  if (ev != EV_NONE) put_item(b, d, b.ref_row[g], t << 2 | (ev == EV_INSERT ? 0u : ev == EV_REMOVE ? 1u : 2u));
}

// the items of one element in time order: an item counts the earlier items IN FRONT of it, and those of its own element are
// then all counted (kd_edit_index takes them out again)
__global__ __launch_bounds__(BLOCK) void kd_items_sort(DeltaBufs d) {
  uint32_t p = gtid();
  if (p >= d.n_list) return;
  uint32_t base = d.item_ex[p], c = d.item_ex[p + 1] - base;
  for (uint32_t i = 1; i < c; i++) {
    uint32_t tk = d.tk[0][base + i];
    uint32_t j = i;
    for (; j > 0 && d.tk[0][base + j - 1] > tk; j--) d.tk[0][base + j] = d.tk[0][base + j - 1];
    d.tk[0][base + j] = tk;
  }
}

__global__ __launch_bounds__(BLOCK) void kd_bit_flags(DeltaBufs d, int src, uint32_t m, uint32_t bit) {
  uint32_t i = gtid();
  if (i > m) return;
  uint32_t z = 0, w = 0;
  if (i < m) {
    uint32_t tk = d.tk[src][i];
    z = (((tk >> 2) >> bit) & 1u) ? 0u : 1u;
    w = z ? ((tk & 3u) == 1u ? 0xffffffffu : (tk & 3u) == 0u ? 1u : 0u) : 0u;  // +1 insert, -1 remove, 0 update (item_weight)
  }
  d.zf[i] = z;
  d.zw[i] = w;
}

// one stable partition of every group by the bit: zeros first. An item whose bit is set has every zero of its group in front of
// it (in position order) earlier in time: it adds their weights.
__device__ __forceinline__ uint32_t item_weight(uint32_t tk) { return (tk & 3u) == 1u ? 0xffffffffu : (tk & 3u) == 0u ? 1u : 0u; }  // +1 insert, -1 remove, 0 update

// (it also leaves the flags of the NEXT level at the item's new place: one launch less per level)
__global__ __launch_bounds__(BLOCK) void kd_partition(DeltaBufs d, int src, uint32_t m, uint32_t bit) {
  uint32_t i = gtid();
  if (i >= m) return;
  int dst = src ^ 1;
  uint32_t tk = d.tk[src][i], lo = d.lo[src][i], hi = d.hi[src][i], acc = d.acc[src][i];
  uint32_t zl = d.zf_ex[lo], zi = d.zf_ex[i], nz = d.zf_ex[hi] - zl;
  uint32_t to, nlo, nhi;
  if (((tk >> 2) >> bit) & 1u) {
    acc += d.zw_ex[i] - d.zw_ex[lo];
    to = lo + nz + ((i - lo) - (zi - zl));
    nlo = lo + nz; nhi = hi;
  } else {
    to = lo + (zi - zl);
    nlo = lo; nhi = lo + nz;
  }
  d.tk[dst][to] = tk; d.elem[dst][to] = d.elem[src][i]; d.acc[dst][to] = acc; d.lo[dst][to] = nlo; d.hi[dst][to] = nhi;
  if (bit > 0) {
    uint32_t z = (((tk >> 2) >> (bit - 1)) & 1u) ? 0u : 1u;
    d.zf[to] = z;
    d.zw[to] = z ? item_weight(tk) : 0u;
  }
}

// All levels in one workgroup with the items in LDS: a batch of a few changes has a few hundred items and would otherwise spend two
// launches per level of a dozen levels on them.
constexpr uint32_t PART_LDS_MAX = 1024;  // items (4 per thread)
__device__ __forceinline__ void partition_lds_block(const DeltaBufs& d, uint32_t m, uint32_t bits) {
  __shared__ uint32_t s_tk[2][PART_LDS_MAX], s_el[2][PART_LDS_MAX], s_acc[2][PART_LDS_MAX], s_lh[2][PART_LDS_MAX];  // lo | hi << 16
  __shared__ uint32_t s_zf[PART_LDS_MAX + 1], s_zw[PART_LDS_MAX + 1], s_scan[BLOCK / WAVE];
  const uint32_t t = threadIdx.x;
  for (uint32_t i = t; i < m; i += BLOCK) {
    s_tk[0][i] = d.tk[0][i]; s_el[0][i] = d.elem[0][i]; s_acc[0][i] = 0; s_lh[0][i] = d.lo[0][i] | d.hi[0][i] << 16;
  }
  __syncthreads();
  int cur = 0;
  for (int bit = (int)bits - 1; bit >= 0; bit--) {
    // exclusive prefixes of the zero flags and of their weights over all items (4 consecutive items per thread)
    uint32_t z[4], w[4], zs = 0, ws = 0;
    for (uint32_t k = 0; k < 4; k++) {
      uint32_t i = 4 * t + k;
      z[k] = w[k] = 0;
      if (i < m) {
        uint32_t tk = s_tk[cur][i];
        z[k] = (((tk >> 2) >> bit) & 1u) ? 0u : 1u;
        w[k] = z[k] ? item_weight(tk) : 0u;
      }
      zs += z[k]; ws += w[k];
    }
    uint32_t zt, wt;
    uint32_t zb = block_exclusive_scan_u32(zs, s_scan, &zt), wb = block_exclusive_scan_u32(ws, s_scan, &wt);
    for (uint32_t k = 0; k < 4; k++) {
      uint32_t i = 4 * t + k;
      if (i <= m) { s_zf[i] = zb; s_zw[i] = wb; }
      zb += z[k]; wb += w[k];
    }
    if (t == 0) { s_zf[m] = zt; s_zw[m] = wt; }   // (m == PART_LDS_MAX: no thread's four items reach index m)
    __syncthreads();
    for (uint32_t k = 0; k < 4; k++) {
      uint32_t i = 4 * t + k;
      if (i >= m) continue;
      uint32_t tk = s_tk[cur][i], lh = s_lh[cur][i], lo = lh & 0xffffu, hi = lh >> 16, acc = s_acc[cur][i];
      uint32_t zl = s_zf[lo], zi = s_zf[i], nz = s_zf[hi] - zl;
      uint32_t to, nlo, nhi;
      if (((tk >> 2) >> bit) & 1u) {
        acc += s_zw[i] - s_zw[lo];
        to = lo + nz + ((i - lo) - (zi - zl));
        nlo = lo + nz; nhi = hi;
      } else {
        to = lo + (zi - zl);
        nlo = lo; nhi = lo + nz;
      }
      s_tk[cur ^ 1][to] = tk; s_el[cur ^ 1][to] = s_el[cur][i]; s_acc[cur ^ 1][to] = acc; s_lh[cur ^ 1][to] = nlo | nhi << 16;
    }
    __syncthreads();
    cur ^= 1;
  }
  const int out = (int)(bits & 1u);  // (where the level-by-level version leaves them)
  for (uint32_t i = t; i < m; i += BLOCK) {
    d.tk[out][i] = s_tk[cur][i]; d.elem[out][i] = s_el[cur][i]; d.acc[out][i] = s_acc[cur][i];
    d.lo[out][i] = s_lh[cur][i] & 0xffffu; d.hi[out][i] = s_lh[cur][i] >> 16;
  }
}
__global__ __launch_bounds__(BLOCK) void kd_partition_lds(DeltaBufs d, uint32_t m, uint32_t bits) { partition_lds_block(d, m, bits); }

// The same for up to PART_BIG_MAX items -- the list edits of a few dozen changes -- in ONE workgroup of 1024 threads with 158 KB of the
// CU's 160 KB of LDS: the level-by-level version spends three dependent launches (~15-24 us) on each of 13-15 levels there. Every
// thread owns up to eleven consecutive items; ONE copy of the items lives in LDS (a level reads its items into registers, computes
// where they go from the packed prefixes, and writes them back after a barrier), the count so far as 16 bits (|acc| <= items < 2^15),
// the element not at all (it follows from the item's row: put_item). (A single workgroup working on HBM instead of LDS was measured
// slower than the launches: profiles/r06_apply_partition_wg.txt.)
constexpr uint32_t PART_BIG_THREADS = 1024, PART_BIG_PER = 11, PART_BIG_MAX = PART_BIG_THREADS * PART_BIG_PER;
__global__ __launch_bounds__(PART_BIG_THREADS) void kd_partition_lds_big(MergeBufs b, DeltaBufs d, uint32_t m, uint32_t bits) {
  __shared__ uint32_t s_tk[PART_BIG_MAX], s_lh[PART_BIG_MAX], s_pre[PART_BIG_MAX + 1];   // lo | hi << 16; zeros in front | weight of them << 16
  __shared__ uint16_t s_acc[PART_BIG_MAX];
  __shared__ uint32_t s_z[PART_BIG_THREADS / WAVE], s_w[PART_BIG_THREADS / WAVE];
  const uint32_t t = threadIdx.x, lane = t & (WAVE - 1), wv = t / WAVE;
  for (uint32_t i = t; i < m; i += PART_BIG_THREADS) { s_tk[i] = d.tk[0][i]; s_lh[i] = d.lo[0][i] | d.hi[0][i] << 16; s_acc[i] = 0; }
  const uint32_t per = (m + PART_BIG_THREADS - 1) / PART_BIG_THREADS;   // <= PART_BIG_PER
  const uint32_t b0 = t * per < m ? t * per : m, cnt = (b0 + per < m ? b0 + per : m) - b0;
  __syncthreads();
  for (int bit = (int)bits - 1; bit >= 0; bit--) {
    uint32_t tk[PART_BIG_PER], lh[PART_BIG_PER], acc[PART_BIG_PER], to[PART_BIG_PER];
    uint32_t zs = 0, ws = 0;
#pragma unroll
    for (uint32_t k = 0; k < PART_BIG_PER; k++)
      if (k < cnt) {
        tk[k] = s_tk[b0 + k]; lh[k] = s_lh[b0 + k]; acc[k] = s_acc[b0 + k];
        if (!(((tk[k] >> 2) >> bit) & 1u)) { zs++; ws += item_weight(tk[k]); }
      }
    // exclusive prefixes of the zero flags and of their weights over all items (both fit 16 bits: at most PART_BIG_MAX items, weights +-1)
    uint32_t zi = wave_incl_scan_u32(zs, lane), wi = wave_incl_scan_u32(ws, lane);
    if (lane == WAVE - 1) { s_z[wv] = zi; s_w[wv] = wi; }
    __syncthreads();
    uint32_t zb = zi - zs, wb = wi - ws, zt = 0, wt = 0;
    for (uint32_t k = 0; k < PART_BIG_THREADS / WAVE; k++) {
      uint32_t x = s_z[k], y = s_w[k];
      zb += k < wv ? x : 0; wb += k < wv ? y : 0;
      zt += x; wt += y;
    }
    const uint32_t zb0 = zb, wb0 = wb;
#pragma unroll
    for (uint32_t k = 0; k < PART_BIG_PER; k++)
      if (k < cnt) {
        s_pre[b0 + k] = (zb & 0xffffu) | wb << 16;
        if (!(((tk[k] >> 2) >> bit) & 1u)) { zb++; wb += item_weight(tk[k]); }
      }
    if (t == 0) s_pre[m] = (zt & 0xffffu) | wt << 16;
    __syncthreads();
    // the stable partition of every group by the bit (kd_partition), destinations computed from the prefixes at the group's bounds
    zb = zb0; wb = wb0;
#pragma unroll
    for (uint32_t k = 0; k < PART_BIG_PER; k++)
      if (k < cnt) {
        const uint32_t i = b0 + k, lo = lh[k] & 0xffffu, hi = lh[k] >> 16;
        const uint32_t pl = s_pre[lo], ph = s_pre[hi];
        const uint32_t zl = pl & 0xffffu, nz = ((ph & 0xffffu) - zl) & 0xffffu;
        if (((tk[k] >> 2) >> bit) & 1u) {
          acc[k] += (wb - (pl >> 16)) & 0xffffu;   // (mod 2^16: the count is kept as 16 bits)
          to[k] = lo + nz + ((i - lo) - ((zb - zl) & 0xffffu));
          lh[k] = (lo + nz) | hi << 16;
        } else {
          to[k] = lo + ((zb - zl) & 0xffffu);
          lh[k] = lo | (lo + nz) << 16;
          zb++; wb += item_weight(tk[k]);
        }
      }
    __syncthreads();   // every thread has its items and its prefixes in registers: the single copy may be overwritten
#pragma unroll
    for (uint32_t k = 0; k < PART_BIG_PER; k++)
      if (k < cnt) { s_tk[to[k]] = tk[k]; s_lh[to[k]] = lh[k]; s_acc[to[k]] = (uint16_t)acc[k]; }
    __syncthreads();
  }
  const int out = (int)(bits & 1u);  // (where the level-by-level version leaves them)
  for (uint32_t i = t; i < m; i += PART_BIG_THREADS) {
    const uint32_t tk = s_tk[i], g = d.T0 + (tk >> 2);
    const uint8_t kind = b.kind[g];
    d.tk[out][i] = tk;
    d.elem[out][i] = ((tk & 3u) == 0u && (kind == K_LIST_INS || kind == K_LIST_INS_VIS)) ? g : b.ref_row[g];   // (put_item: a new element's own insert row, or the element a row deletes / assigns to)
    d.acc[out][i] = (uint32_t)(int32_t)(int16_t)s_acc[i];
    d.lo[out][i] = s_lh[i] & 0xffffu; d.hi[out][i] = s_lh[i] >> 16;
  }
}

// The same counts TILE BY TILE over the whole device (round 6, second half): for 1 k - 64 k items. What an item needs -- the weights of
// the earlier items in front of it, and how many items of its list come before it in (time, position) order -- is a two-dimensional
// dominance count, and it splits over pairs of 256-item tiles: every tile sorts its items by time once (kd_dom_tiles, which also does
// the pairs inside the tile), and a (tile J, tile I) workgroup answers for the 256 items of I with ONE binary search each into J's sorted
// times (kd_dom_cross: prefix weights at the found place for J in front of I, the bare count for J behind). 40 changes of 250 ops:
// 40 x 40 workgroups of ~2 us side by side instead of 14 dependent levels in one workgroup (94 us -> ~10). A tile that holds the border
// of two lists is compared item by item. kd_dom_scatter then moves every item to its place in (list, time) order and leaves the index
// of its edit there (kd_edit_index's work: one launch less).
constexpr uint32_t DOM_TILE = BLOCK;
constexpr uint32_t DOM_ITEMS_MAX = DOM_TILE * 256;
constexpr uint32_t DOM_SPLIT = 4;   // threads per item in kd_dom_tiles: each takes a quarter of the tile's items (256 dependent steps of one thread were 17 us)
__global__ __launch_bounds__(DOM_TILE * DOM_SPLIT) void kd_dom_tiles(DeltaBufs d, uint32_t m) {
  __shared__ uint32_t s_tk[DOM_TILE], s_lo[DOM_TILE], s_srt_t[DOM_TILE], s_srt_w[DOM_TILE], s_scan[DOM_TILE / WAVE];
  __shared__ uint32_t s_acc[DOM_TILE], s_rank_list[DOM_TILE], s_rank_tile[DOM_TILE];
  const uint32_t t = threadIdx.x % DOM_TILE, part = threadIdx.x / DOM_TILE, i0 = blockIdx.x * DOM_TILE, i = i0 + t, cnt = m - i0 < DOM_TILE ? m - i0 : DOM_TILE;
  uint32_t tk = NONE32, lo = NONE32;
  if (t < cnt) { tk = d.tk[0][i]; lo = d.lo[0][i]; }
  if (part == 0) { s_tk[t] = tk; s_lo[t] = lo; s_acc[t] = 0; s_rank_list[t] = 0; s_rank_tile[t] = 0; }
  __syncthreads();
  if (t < cnt) {
    uint32_t acc = 0, rank_list = 0, rank_tile = 0;
    const uint32_t ti = tk >> 2, per = DOM_TILE / DOM_SPLIT, j1 = (part + 1) * per < cnt ? (part + 1) * per : cnt;
    for (uint32_t j = part * per; j < j1; j++) {
      const uint32_t tkj = s_tk[j], tj = tkj >> 2;
      const bool less = tj < ti, before = less || (tj == ti && j < t), same = s_lo[j] == lo;
      rank_tile += before ? 1u : 0u;
      rank_list += before && same ? 1u : 0u;
      if (less && same && j < t) acc += item_weight(tkj);
    }
    if (acc) atomicAdd(&s_acc[t], acc);
    if (rank_list) atomicAdd(&s_rank_list[t], rank_list);
    if (rank_tile) atomicAdd(&s_rank_tile[t], rank_tile);
  }
  __syncthreads();
  if (part == 0 && t < cnt) { s_srt_t[s_rank_tile[t]] = tk >> 2; s_srt_w[s_rank_tile[t]] = item_weight(tk); }
  __syncthreads();
  // the weights up to and with each item in time order (the first DOM_TILE threads: a wavefront scan each, then the wavefronts in front)
  const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const uint32_t w = part == 0 && t < cnt ? s_srt_w[t] : 0u;
  uint32_t incl = 0;
  if (part == 0) {   // (whole wavefronts: DOM_TILE is a multiple of WAVE)
    incl = wave_incl_scan_u32(w, lane);
    if (lane == WAVE - 1) s_scan[wv] = incl;
  }
  __syncthreads();
  if (part == 0 && t < cnt) {
    for (uint32_t k = 0; k < wv; k++) incl += s_scan[k];
    d.zf[i] = s_srt_t[t]; d.zw[i] = incl; d.acc[0][i] = s_acc[t]; d.lo[1][i] = s_rank_list[t];   // (zf | zw: the tile's times ascending, the weights up to and with each)
  }
}
__global__ __launch_bounds__(BLOCK) void kd_dom_cross(DeltaBufs d, uint32_t m) {
  const uint32_t J = blockIdx.x, I = blockIdx.y;
  if (J == I) return;
  const uint32_t i0 = I * DOM_TILE, j0 = J * DOM_TILE, icnt = m - i0 < DOM_TILE ? m - i0 : DOM_TILE, jcnt = m - j0 < DOM_TILE ? m - j0 : DOM_TILE;
  if (j0 + jcnt <= d.lo[0][i0] || j0 >= d.hi[0][i0 + icnt - 1]) return;   // (the lists of tile I's items: from its first item's to its last item's)
  __shared__ uint32_t s_t[DOM_TILE], s_w[DOM_TILE], s_tk[DOM_TILE], s_lo[DOM_TILE];
  const uint32_t t = threadIdx.x;
  const bool whole = d.lo[0][j0] == d.lo[0][j0 + jcnt - 1];   // tile J lies inside one list
  if (t < jcnt) {
    if (whole) { s_t[t] = d.zf[j0 + t]; s_w[t] = d.zw[j0 + t]; }
    else { s_tk[t] = d.tk[0][j0 + t]; s_lo[t] = d.lo[0][j0 + t]; }
  }
  __syncthreads();
  if (t >= icnt) return;
  const uint32_t i = i0 + t, lo = d.lo[0][i];
  if (j0 + jcnt <= lo || j0 >= d.hi[0][i]) return;
  const uint32_t ti = d.tk[0][i] >> 2;
  const bool front = J < I;
  uint32_t acc = 0, rank = 0;
  if (whole) {
    uint32_t a = 0, n = jcnt;   // items of J earlier than ti
    while (n) {
      const uint32_t h = n >> 1;
      if (s_t[a + h] < ti) { a += h + 1; n -= h + 1; } else n = h;
    }
    rank = a;
    if (front) {
      acc = a ? s_w[a - 1] : 0u;
      while (rank < jcnt && s_t[rank] == ti) rank++;   // (items of the same row in front of it come first)
    }
  } else {
    for (uint32_t j = 0; j < jcnt; j++) {
      if (s_lo[j] != lo) continue;
      const uint32_t tkj = s_tk[j], tj = tkj >> 2;
      if (tj < ti) { rank++; if (front) acc += item_weight(tkj); }
      else if (front && tj == ti) rank++;
    }
  }
  if (acc) atomicAdd(&d.acc[0][i], acc);
  if (rank) atomicAdd(&d.lo[1][i], rank);
}

// items are now in (object, time) order: index of each edit
__device__ __forceinline__ uint32_t edit_index_of(const MergeBufs& b, const DeltaBufs& d, uint32_t tk, uint32_t e, uint32_t acc) {
  uint32_t t = tk >> 2, g = d.T0 + t;
  uint32_t oi = obj_index_of(b, b.obj_row[e]);
  uint32_t p = d.pos_of[e];
  uint32_t idx = d.v0_ex[p] - d.v0_ex[b.obj_first_pos[oi]] + acc;
  // the earlier items of its own element sit at the same position and are not "in front": together they moved the element from
  // its visibility before the batch to its visibility just before this item
  if (g != e) idx -= (d.ev_before[t] & 1u) - d.v0[p] + (d.ev_before[t] >> 1);  // (bit 1: the reference's index lag, kd_events)
  return idx;
}
__device__ __forceinline__ void edit_index_item(const MergeBufs& b, const DeltaBufs& d, int src, uint32_t i) {
  d.e_index[i] = edit_index_of(b, d, d.tk[src][i], d.elem[src][i], d.acc[src][i]);
}
__global__ __launch_bounds__(BLOCK) void kd_edit_index(MergeBufs b, DeltaBufs d, int src, uint32_t m) {
  uint32_t i = gtid();
  if (i < m) edit_index_item(b, d, src, i);
}
__global__ __launch_bounds__(BLOCK) void kd_dom_scatter(MergeBufs b, DeltaBufs d, uint32_t m) {
  uint32_t i = gtid();
  if (i >= m) return;
  const uint32_t to = d.lo[0][i] + d.lo[1][i], tk = d.tk[0][i], e = d.elem[0][i], acc = d.acc[0][i];
  d.tk[1][to] = tk; d.elem[1][to] = e; d.acc[1][to] = acc;
  d.e_index[to] = edit_index_of(b, d, tk, e, acc);
}

// ---- the edits array (new.js:747-869), item by item -------------------------------------------------------------------------
struct ItemView {
  uint32_t k, e, g, t, oi, idx;
  bool own_insert;  // the insert row of a new element (elemId == opId: the only items appendEdit can chain into a multi-insert)
};
__device__ __forceinline__ ItemView item_view(const MergeBufs& b, const DeltaBufs& d, int src, uint32_t i) {
  ItemView v;
  uint32_t tk = d.tk[src][i];
  v.k = tk & 3u; v.t = tk >> 2; v.e = d.elem[src][i]; v.g = d.T0 + v.t;
  v.oi = obj_index_of(b, b.obj_row[v.e]);
  v.idx = d.e_index[i];
  v.own_insert = v.k == 0 && v.g == v.e;
  return v;
}
// appendUpdate(firstUpdate) of a later call pops what the edits array ends with at the same index (new.js:800-818): an item is
// taken away again when the next item of its object is an update at its index and it is itself an insert or an update
__device__ __forceinline__ bool item_popped(const MergeBufs& b, const DeltaBufs& d, int src, uint32_t i, uint32_t m, const ItemView& v) {
  if (v.k == 1 || i + 1 >= m) return false;
  ItemView nx = item_view(b, d, src, i + 1);
  return nx.oi == v.oi && nx.k == 2 && nx.idx == v.idx;
}
// does own-insert item i continue the multi-insert of item i - 1 (appendEdit, new.js:754-772)? bit 0: yes, bit 1: as a new record
__device__ __forceinline__ uint32_t continues_multi_insert(const MergeBufs& b, const DeltaBufs& d, int src, uint32_t i, const ItemView& v) {
  if (!v.own_insert || i == 0) return 0;
  const OpCols& o = b.ops;
  ItemView pv = item_view(b, d, src, i - 1);
  if (!pv.own_insert || pv.oi != v.oi) return 0;
  uint32_t e = v.e, pe = pv.e;
  bool child = (o.action[e] & 1) == 0, pchild = (o.action[pe] & 1) == 0;
  if (child || pchild || o.id_actor[e] != o.id_actor[pe] || o.id_ctr[e] != o.id_ctr[pe] + 1 || value_class(o.val_tl[e]) != value_class(o.val_tl[pe]) ||
      v.idx != pv.idx + 1)
    return 0;
  uint32_t tl = o.val_tl[e], ptl = o.val_tl[pe];
  return (tl != ptl || o.val_off[e] != o.val_off[pe] + (ptl >> 4)) ? 3u : 1u;
}

constexpr uint32_t POP_WALK_MAX = 1u << 16;  // update items at one index in a row the LAST of them walks back over (more: refused)

__device__ __forceinline__ void edit_runs_item(const MergeBufs& b, const DeltaBufs& d, int src, uint32_t m, uint32_t i) {   // i <= m
  if (i == m) { d.e_head[i] = 0; d.e_val[i] = 0; return; }
  const OpCols& o = b.ops;
  ItemView v = item_view(b, d, src, i);
  uint32_t f = 0, recs = 0, vals = 0;
  uint32_t prev_oi = i > 0 ? item_view(b, d, src, i - 1).oi : NONE32;
  uint32_t next_oi = i + 1 < m ? item_view(b, d, src, i + 1).oi : NONE32;
  const bool popped = item_popped(b, d, src, i, m, v);
  if (v.k == 1) {
    f = AM355_EDIT_REMOVE;
    bool joins = false;
    if (i > 0) {
      ItemView pv = item_view(b, d, src, i - 1);
      joins = pv.oi == v.oi && pv.k == 1 && pv.idx == v.idx;  // (a popped item is followed by an update, never by a remove)
    }
    if (joins) f |= 2u;
    recs = joins ? 0u : 1u;
    vals = 1;
  } else if (v.own_insert) {
    if ((o.action[v.e] & 1) == 0) f |= AM355_EDIT_CHILD;
    uint32_t c = continues_multi_insert(b, d, src, i, v);
    if (c & 1u) f |= 2u;
    if (c & 2u) f |= 0x400u;
    if (popped) f |= 0x800u;
    else {
      recs = (!(c & 1u) || (c & 2u)) ? 1u : 0u;
      vals = 1;
      // the head of a two-value multi-insert whose second value a later update popped stays a multi-insert (new.js:812-814)
      if (!(c & 1u) && i + 1 < m) {
        ItemView nx = item_view(b, d, src, i + 1);
        if (nx.oi == v.oi && (continues_multi_insert(b, d, src, i + 1, nx) & 1u) && item_popped(b, d, src, i + 1, m, nx)) f |= AM355_EDIT_MULTI;
      }
    }
  } else {
    // an update, or an element that comes back (insert with opId != elemId): one record per visible value
    if (popped) f |= 0x800u;
    else {
      recs = vals = d.ev_nafter[v.t];
      bool as_insert = v.k == 0;
      if (v.k == 2) {
        // what did the first appendUpdate of this call pop? every update item in front at this index, then at most one insert
        uint32_t j = i, steps = 0;
        while (j > 0 && steps <= POP_WALK_MAX) {
          ItemView pv = item_view(b, d, src, j - 1);
          if (pv.oi != v.oi || pv.idx != v.idx || pv.k == 1) break;
          if (pv.k == 0) { as_insert = true; break; }
          j--; steps++;
        }
        if (steps > POP_WALK_MAX) atomicOr(&d.counts->flags, refuse(d, DR_POP_WALK));
      }
      if (as_insert) f |= 0x1000u;
    }
  }
  if (v.oi != prev_oi) f |= 0x100u;
  if (v.oi != next_oi) f |= 0x200u;
  d.e_flags[i] = f;
  d.e_head[i] = recs;
  d.e_val[i] = vals;
}
__global__ __launch_bounds__(BLOCK) void kd_edit_runs(MergeBufs b, DeltaBufs d, int src, uint32_t m) {
  uint32_t i = gtid();
  if (i <= m) edit_runs_item(b, d, src, m, i);
}

__device__ __forceinline__ void edit_pack_item(const MergeBufs& b, const DeltaBufs& d, int src, uint32_t m, uint32_t i) {   // i <= m
  if (i == m) {
    uint32_t n = d.e_head_ex[m];
    d.edit[n] = am355_ir_edit{0, 0, 0, 0, 0, 0, d.e_val_ex[m], 0, 0, 0};
    d.counts->n_erecs = n;
    return;
  }
  const OpCols& o = b.ops;
  ItemView v = item_view(b, d, src, i);
  uint32_t f = d.e_flags[i], k = d.e_head_ex[i], first = d.e_val_ex[i], recs = d.e_head[i];
  uint32_t e = v.e;
  if (recs) {
    if (f & AM355_EDIT_REMOVE) d.edit[k] = am355_ir_edit{AM355_EDIT_REMOVE, v.idx, 0, 0, 0, 0, first, 0, 0, 0};
    else if (v.own_insert) {
      uint32_t rf = f & (AM355_EDIT_CHILD | AM355_EDIT_MULTI);
      if ((f & 2u) && (f & 0x400u)) rf |= AM355_EDIT_CONT;
      d.edit[k] = am355_ir_edit{rf, v.idx, o.id_ctr[e], o.id_actor[e], o.id_ctr[e], o.id_actor[e], first, o.val_tl[e], (f & AM355_EDIT_CHILD) ? b.obj_index[e] : o.val_off[e], 0};
    } else {
      // the element's visible values after row g, ascending by op id
      const uint32_t nu = d.upd_n[e], base = d.upd_off[e];
      unsigned long long last = 0;
      const bool quirk = elem_has_inc(b, d, e);
      for (uint32_t j = 0; quirk && j < recs; j++) {
        // (an element with increments: a counter stands where its last increment stands and shows its total, quirk_value_at)
        unsigned long long best = ~0ull;
        uint32_t best_r = NONE32;
        long long best_total = 0;
        bool best_is_total = false;
        for (uint32_t q = 0; q <= nu; q++) {
          const uint32_t r = elem_row(d, e, q);
          unsigned long long key = 0;
          long long total = 0;
          bool is_total = false;
          if (!quirk_value_at(b, d, e, r, v.g + 1, &key, &total, &is_total)) continue;
          if (key > last && key < best) { best = key; best_r = r; best_total = total; best_is_total = is_total; }
        }
        if (best_r == NONE32) { atomicOr(&d.counts->flags, refuse(d, DR_INTERNAL)); break; }
        last = best;
        uint32_t rf = (j == 0 && (f & 0x1000u)) ? 0u : (uint32_t)AM355_EDIT_UPDATE;
        const bool child = (o.action[best_r] & 1u) == 0;
        if (child) rf |= AM355_EDIT_CHILD;
        uint32_t val = child ? b.obj_index[best_r] : o.val_off[best_r], val_hi = 0;
        if (best_is_total) { rf |= AM355_EDIT_COUNTER; val = (uint32_t)(unsigned long long)best_total; val_hi = (uint32_t)((unsigned long long)best_total >> 32); }
        d.edit[k + j] = am355_ir_edit{rf, v.idx, o.id_ctr[best_r], o.id_actor[best_r], o.id_ctr[e], o.id_actor[e], first + j, o.val_tl[best_r], val, val_hi};
      }
      for (uint32_t j = 0; !quirk && j < recs; j++) {
        unsigned long long best = ~0ull;
        uint32_t best_r = NONE32;
        for (uint32_t q = 0; q <= nu; q++) {
          uint32_t r = q == 0 ? e : d.upd_rows[base + q - 1];
          if (r > v.g || !alive_at_T0(b, d, r) || d.first_del[r] <= v.g) continue;
          unsigned long long id = pack_id(o.id_ctr[r], o.id_actor[r]);
          if (id > last && id < best) { best = id; best_r = r; }
        }
        if (best_r == NONE32) { atomicOr(&d.counts->flags, refuse(d, DR_INTERNAL)); break; }
        last = best;
        uint32_t rf = (j == 0 && (f & 0x1000u)) ? 0u : (uint32_t)AM355_EDIT_UPDATE;
        const bool child = (o.action[best_r] & 1u) == 0;
        if (child) rf |= AM355_EDIT_CHILD;
        d.edit[k + j] = am355_ir_edit{rf, v.idx, o.id_ctr[best_r], o.id_actor[best_r], o.id_ctr[e], o.id_actor[e], first + j, o.val_tl[best_r],
                                      child ? b.obj_index[best_r] : o.val_off[best_r], 0};
      }
    }
  }
  if (f & 0x100u) d.link[v.oi].edit_begin = k;
  if (f & 0x200u) d.link[v.oi].edit_end = d.e_head_ex[i + 1];
}
__global__ __launch_bounds__(BLOCK) void kd_edit_pack(MergeBufs b, DeltaBufs d, int src, uint32_t m) {
  uint32_t i = gtid();
  if (i <= m) edit_pack_item(b, d, src, m, i);
}

// The whole second half of the stage for a batch of a change or two, in ONE workgroup: the partition levels in LDS, the index of every
// edit, the runs, their two prefix sums and the records -- six launches and, in front of them, the host's wait for the item count
// (which sizes their grids) otherwise: ~45 us of a call that takes 0.3 ms, most of it the host's launch rate. The item count is read
// HERE; what this workgroup does not handle -- more than PART_LDS_MAX items, map records to order, more edit records than the table holds
// -- is left alone and reported through DeltaCounts.deferred: the host then runs the second half as it does for larger batches.
// With d.host_edit set it also hands the result over: the object links and the edit records go to pinned host memory from here and the
// counters are signalled behind them (HostSignals.delta_mid) -- in every case, also when there is nothing to hand over.
__device__ __forceinline__ void edit_small_block(const MergeBufs& b, const DeltaBufs& d);
__global__ __launch_bounds__(BLOCK) void kd_edit_small(MergeBufs b, DeltaBufs d) {
  edit_small_block(b, d);
  if (!d.host_edit) return;
  __syncthreads();
  const DeltaCounts c = *d.counts;
  if (!c.flags && !c.deferred) {
    const uint32_t* src = (const uint32_t*)d.link;
    uint32_t* dst = (uint32_t*)d.host_link;
    for (uint32_t w = threadIdx.x; w < d.n_obj * (uint32_t)(sizeof(ObjLink) / 4); w += BLOCK) dst[w] = src[w];
    src = (const uint32_t*)d.edit;
    dst = (uint32_t*)d.host_edit;
    for (uint32_t w = threadIdx.x; w < (c.n_erecs + 1) * (uint32_t)(sizeof(am355_ir_edit) / 4); w += BLOCK) dst[w] = src[w];
    __threadfence_system();
  }
  __syncthreads();
  if (threadIdx.x == 0) signal_host(d.sig->delta_mid, (const uint32_t*)d.counts, (uint32_t)(sizeof(DeltaCounts) / 4), &d.sig->delta_mid_seq, d.sig_seq);
}
__device__ __forceinline__ void edit_small_block(const MergeBufs& b, const DeltaBufs& d) {
  __shared__ uint32_t s_scan[BLOCK / WAVE];
  const DeltaCounts c = *d.counts;
  if (c.flags) return;   // (refused by the first half: nothing to build)
  const uint32_t m = c.n_items;
  if (m > PART_LDS_MAX || c.n_kept + c.n_place != 0 || (uint64_t)m + c.rec_extra + 2 > d.edit_cap) {
    if (threadIdx.x == 0) d.counts->deferred = 1;
    return;
  }
  const uint32_t t = threadIdx.x;
  int src = 0;
  if (m) {
    partition_lds_block(d, m, d.bits_new);
    src = (int)(d.bits_new & 1u);
    __syncthreads();
    for (uint32_t i = t; i < m; i += BLOCK) edit_index_item(b, d, src, i);
    __syncthreads();
  }
  for (uint32_t i = t; i <= m; i += BLOCK) edit_runs_item(b, d, src, m, i);
  __syncthreads();
  {
    // e_head_ex / e_val_ex over the m + 1 entries: a thread takes `per` consecutive ones
    const uint32_t n = m + 1, per = (n + BLOCK - 1) / BLOCK, i0 = t * per < n ? t * per : n, i1 = i0 + per < n ? i0 + per : n;
    uint32_t hs = 0, vs = 0;
    for (uint32_t i = i0; i < i1; i++) { hs += d.e_head[i]; vs += d.e_val[i]; }
    uint32_t tot;
    uint32_t hb = block_exclusive_scan_u32(hs, s_scan, &tot), vb = block_exclusive_scan_u32(vs, s_scan, &tot);
    for (uint32_t i = i0; i < i1; i++) {
      const uint32_t h = d.e_head[i], v = d.e_val[i];
      d.e_head_ex[i] = hb; d.e_val_ex[i] = vb;
      hb += h; vb += v;
    }
  }
  __syncthreads();
  for (uint32_t i = t; i <= m; i += BLOCK) edit_pack_item(b, d, src, m, i);
}

// ---------------------------------------------------------------------------------------------------------
// maps
// ---------------------------------------------------------------------------------------------------------
// first key < second key in UTF-16 code unit order (new.js:1126 compares JS strings)
__device__ __forceinline__ bool key_less_utf16(const MergeBufs& b, uint32_t ra, uint32_t rb) {
  const uint8_t *p = b.arena + b.ops.key_off[ra], *q = b.arena + b.ops.key_off[rb];
  uint32_t la = b.ops.key_len[ra], lb = b.ops.key_len[rb], n = la < lb ? la : lb;
  for (uint32_t k = 0; k < n; k++) {
    uint32_t x = utf16_order_byte(p[k]), y = utf16_order_byte(q[k]);
    if (x != y) return x < y;
  }
  return la < lb;
}

// per touched key: id of the batch's last op on it, and whether the merge call that handled that op went on to another key
__global__ __launch_bounds__(BLOCK) void kd_slots(MergeBufs b, DeltaBufs d) {
  uint32_t s = gtid();
  if (s > d.key_mask) return;
  if (!d.slot_rep[s]) return;
  const OpCols& o = b.ops;
  uint32_t j = d.slot_last[s];
  unsigned long long L = pack_id(o.id_ctr[j], o.id_actor[j]);
  bool ambiguous = false;
  if (b.kind[j] == K_DEL) {
    // a deletion leaves the merge loop's work list as soon as the document op of its last pred has been taken (new.js:1207-1216):
    // from there on -- not from the deletion's own id -- the remaining document ops of the key are at the mercy of the next op
    L = 0;
    for (uint32_t k = 0; k < o.pred_num[j]; k++) {
      unsigned long long pid = pack_id(o.pred_ctr[o.pred_first[j] + k], o.pred_actor[o.pred_first[j] + k]);
      L = pid > L ? pid : L;
    }
    // ... unless an earlier op on the same key travels in the same work list (new.js:1111-1113): not modelled
    if (j > d.T0) {
      uint32_t pj = j - 1;
      uint8_t kp = b.kind[pj];
      ambiguous = (kp == K_MAP || kp == K_DEL) && o.key_len[pj] != NONE32 && o.id_actor[pj] == o.id_actor[j] && same_obj(b, pj, j) && same_key(b, pj, j);
      // (a deletion OF that earlier op starts a merge call of its own, new.js:1114-1121: nothing travels with it)
      for (uint32_t k = 0; k < o.pred_num[j]; k++)
        if (o.pred_ctr[o.pred_first[j] + k] == o.id_ctr[pj] && o.pred_actor[o.pred_first[j] + k] == o.id_actor[pj]) ambiguous = false;
    }
  }
  d.slot_L[s] = L;
  uint32_t cont = 0, nx = j + 1;
  if (nx < b.n_ops) {
    bool new_pass = false;
    for (uint32_t k = 0; k < d.n_pass; k++) new_pass = new_pass || d.pass_rows[k] == nx;
    uint8_t kn = b.kind[nx];
    if (!new_pass && (kn == K_MAP || kn == K_DEL) && o.key_len[nx] != NONE32 && !o.insert[nx] && o.id_actor[nx] == o.id_actor[j] && same_obj(b, nx, j) &&
        key_less_utf16(b, j, nx))
      cont = 1;
  }
  if (cont && ambiguous) atomicOr(&d.counts->flags, refuse(d, DR_AMBIGUOUS_DEL));
  d.slot_cont[s] = cont;
}

// per map record of the whole-document patch: does the batch touch its key, does the patch list it
__global__ __launch_bounds__(BLOCK) void kd_map_records(MergeBufs b, PatchIR ir, DeltaBufs d) {
  uint32_t i = gtid();
  if (i > d.n_map) return;
  uint32_t keep = 0;
  if (i < d.n_map) {
    const am355_ir_map rec = ir.map[i];
    uint32_t g = row_of(b, rec.id_actor, rec.id_ctr);
    uint32_t s = g == NONE32 ? NONE32 : key_slot(b, d, g, false);
    d.rec_slot[i] = s;
    if (s != NONE32) {
      unsigned long long trig = (rec.flags & AM355_MAP_COUNTER) ? b.last_inc[g] : pack_id(rec.id_ctr, rec.id_actor);
      bool drop = d.slot_cont[s] && trig > d.slot_L[s];
      if (rec.flags & AM355_MAP_CHILD) d.slot_child[s] = 1;
      if (drop) d.slot_drop[s] = 1;
      else { keep = 1; atomicAdd(&d.slot_cnt[s], 1u); }
    }
  }
  d.keep[i] = keep;
}

__global__ __launch_bounds__(BLOCK) void kd_placeholders(DeltaBufs d) {
  uint32_t s = gtid();
  if (s > d.key_mask + 1) return;
  uint32_t pl = 0;
  if (s <= d.key_mask && d.slot_rep[s]) {
    pl = d.slot_cnt[s] == 0 ? 1u : 0u;
    // values were skipped on a key that holds a child object: objectMeta.children of the reference now lacks them, which later
    // patches show (new.js:916-931) -- from here on the host asks delta_key_history what a property lists instead of taking its
    // visible values
    if (d.slot_child[s] && d.slot_drop[s]) d.counts->hazard = 1;
  }
  d.place[s] = pl;
}

// sort keys: (object | first row that touched the key); records of one key keep their op id order (stable sort)
__global__ __launch_bounds__(BLOCK) void kd_map_pairs(MergeBufs b, DeltaBufs d) {
  uint32_t i = gtid();
  const uint32_t n_kept = d.counts->n_kept;
  if (i < d.n_map && d.keep[i]) {
    uint32_t s = d.rec_slot[i], rep = d.slot_rep[s] - 1;
    uint32_t pos = d.keep_ex[i];
    d.pair_key[0][pos] = (uint64_t)obj_index_of(b, b.obj_row[rep]) << d.bits_new | (d.slot_first[s] - d.T0);
    d.pair_val[0][pos] = i;
  }
  if (i <= d.key_mask && d.place[i]) {
    uint32_t rep = d.slot_rep[i] - 1;
    uint32_t pos = n_kept + d.place_ex[i];
    d.pair_key[0][pos] = (uint64_t)obj_index_of(b, b.obj_row[rep]) << d.bits_new | (d.slot_first[i] - d.T0);
    d.pair_val[0][pos] = d.n_map + i;
  }
}

__global__ __launch_bounds__(BLOCK) void kd_map_out(MergeBufs b, PatchIR ir, DeltaBufs d, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n) {
  uint32_t i = gtid();
  if (i >= n) return;
  uint32_t v = vals[i];
  if (v < d.n_map) d.map[i] = ir.map[v];
  else {
    uint32_t rep = d.slot_rep[v - d.n_map] - 1;
    d.map[i] = am355_ir_map{0, 0, b.ops.key_off[rep], b.ops.key_len[rep], 0, 0, AM355_MAP_EMPTY, 0, 0};
  }
  uint32_t oi = (uint32_t)(keys[i] >> d.bits_new);
  uint32_t prev = i > 0 ? (uint32_t)(keys[i - 1] >> d.bits_new) : NONE32, next = i + 1 < n ? (uint32_t)(keys[i + 1] >> d.bits_new) : NONE32;
  if (oi != prev) d.link[oi].map_begin = i;
  if (oi != next) d.link[oi].map_end = i + 1;
}

const char* delta_reason_text(uint32_t reason) {
  switch (reason) {
    case DR_FOREIGN_ROW: return "rows of objects another shard owns";
    case DR_KEY_TABLE: return "touched-key table full";
    case DR_ELEM_ROWS: return "a touched list element holds more value rows than the stage walks";
    case DR_ELEM_NOT_PLAIN: return "an assigned list element holds an op that is neither a value nor a child object";
    case DR_SAME_ELEM_CALL: return "two ops on one list element in one merge call";
    case DR_GAP_WALK: return "too many later insertions between two elements of one merge call";
    case DR_LAGGING_UPDATE: return "the first update edit of a conflict would sit at the reference's lagging list index";
    case DR_POP_WALK: return "too many update edits at one list index in a row";
    case DR_AMBIGUOUS_DEL: return "a deletion whose place in the merge loop's work list is ambiguous";
    case DR_CHILD_HAZARD: return "values skipped on a key that holds a child object";
    case DR_EDIT_TABLE: return "more edit records than the edit table holds";
    case DR_INTERNAL: return "internal: visible values of an element not found";
    default: return "outside the served subset";
  }
}

static int dbits_for(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b)) b++;
  return b;
}

void delta_run(MergeBufs& b, PatchIR& ir, DeltaBufs& d, DeltaCounts* hc, hipStream_t st, bool check_only, DeltaGrowEdit grow_edit, void* grow_user,
               DeltaBeforeEnd before_end) {
  // the counters of a half: through the pinned words a one-thread launch behind it signals (no copy dispatch, no interrupt wake-up: ~30 us
  // of a 0.15 ms stage), or -- no signal block, or no signal -- by a copy after a drain
  auto read_counts = [&](volatile uint32_t* seq_word, uint32_t* words) {
    if (d.sig) {
      launch_signal_words((const uint32_t*)d.counts, (uint32_t)(sizeof(DeltaCounts) / 4), nullptr, 0, words, seq_word, d.sig_seq, st);
      if (wait_host_signal(seq_word, d.sig_seq, st)) { *hc = *(const DeltaCounts*)words; return; }
    }
    (void)hipMemcpyAsync(hc, d.counts, sizeof(DeltaCounts), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
  };
  const uint32_t N = b.n_ops, cap = d.key_mask + 1;
  static const bool debug = getenv("AM355_DELTA_DEBUG") != nullptr;  // (diagnostic: drain the stream after every step and say which)
  auto step = [&](const char* what) {
    if (canary_on()) {  // AM355_CANARY=1: which step of the stage wrote past an array?
      char msg[320];
      if (!canary_check(msg, sizeof msg)) fprintf(stderr, "delta_run: after '%s': %s\n", what, msg);
    }
    if (!debug) return;
    hipError_t e = hipStreamSynchronize(st);
    fprintf(stderr, "delta_run: %-18s %s (N %u T0 %u new %u obj %u map %u list %u cap %u)\n", what, hipGetErrorString(e), N, d.T0, d.n_new, d.n_obj, d.n_map, d.n_list, cap);
  };
  {
    // every fill of the stage in ONE launch (seven memsets in a row cost a launch gap each: ~35 us of a 0.2 ms stage)
    FillRanges f;
    f.add(d.counts, sizeof(DeltaCounts), 0);
    f.add(&d.counts->reason, 4, 0xffffffffu);   // (behind the range that clears it: small ranges are written in order by one thread)
    f.add(d.first_del, (size_t)((uint8_t*)(d.first_kill + N + 1) - (uint8_t*)d.first_del), 0xffffffffu);   // first_del | first_kill
    f.add(d.new_succ, (size_t)((uint8_t*)(d.upd_cur + N + 2) - (uint8_t*)d.new_succ), 0);                  // new_succ | upd_n | upd_cur
    f.add(d.icur, 4 * ((size_t)d.n_list + 2), 0);
    f.add(d.slot_rep, (size_t)((uint8_t*)(d.slot_drop + cap + 1) - (uint8_t*)d.slot_rep), 0);  // slot_rep | slot_last | slot_cnt | slot_child | slot_drop
    f.add(d.slot_first, 4 * ((size_t)cap + 1), 0xffffffffu);
    launch_fill_ranges(f, st);
  }
  AM355_LAUNCH_INDEPENDENT(kd_objects, dgrid(d.n_obj), dim3(BLOCK), st, b, ir, d);
  step("objects");
  if (N) AM355_LAUNCH_INDEPENDENT(kd_rows, dgrid(N), dim3(BLOCK), st, b, d);
  if (d.n_new) hipLaunchKernelGGL(kd_touch, dgrid(d.n_new), dim3(BLOCK), 0, st, b, d);
  step("rows");
  if (debug) {
    std::vector<uint32_t> ord(d.n_list + 1);
    (void)hipMemcpy(ord.data(), b.order, 4 * (size_t)d.n_list, hipMemcpyDeviceToHost);
    uint32_t bad = 0;
    for (uint32_t k = 0; k < d.n_list; k++) if (ord[k] >= N) { bad++; fprintf(stderr, "delta_run:   order[%u] = 0x%x\n", k, ord[k]); }
    {
      std::vector<uint32_t> fp(d.n_obj + 2), on(d.n_obj + 2);
      (void)hipMemcpy(fp.data(), b.obj_first_pos, 4 * (size_t)(d.n_obj + 1), hipMemcpyDeviceToHost);
      (void)hipMemcpy(on.data(), b.obj_n, 4 * (size_t)(d.n_obj + 1), hipMemcpyDeviceToHost);
      for (uint32_t k = 0; k <= d.n_obj; k++) fprintf(stderr, "delta_run:   object %u first_pos %u n %u\n", k, fp[k], on[k]);
    }
    fprintf(stderr, "delta_run: order[0..%u): %u entries out of range; first %u %u %u\n", d.n_list, bad, d.n_list ? ord[0] : 0, d.n_list > 1 ? ord[1] : 0, d.n_list > 2 ? ord[2] : 0);
  }
  // ---- lists: the assignment rows of every element; items in position order ----
  exclusive_scan_u32(d.upd_n, d.upd_off, N + 1, nullptr, d.scan_ws, st);
  if (N) AM355_LAUNCH_INDEPENDENT(kd_upd_scatter, dgrid(N), dim3(BLOCK), st, b, d);
  AM355_LAUNCH_INDEPENDENT(kd_positions, dgrid(d.n_list + 1), dim3(BLOCK), st, b, d);
  step("positions");
  if (d.n_new) AM355_LAUNCH_INDEPENDENT(kd_events, dgrid(d.n_new), dim3(BLOCK), st, b, d);
  step("events");
  // (the totals the host sizes the second half from -- items, kept map records, placeholders -- are written by the scans themselves)
  exclusive_scan2_u32(d.v0, d.v0_ex, nullptr, d.icnt, d.item_ex, &d.counts->n_items, d.n_list + 1, d.scan_ws, st);
  step("scan positions");
  if (d.n_new && d.n_list) AM355_LAUNCH_INDEPENDENT(kd_items, dgrid(d.n_new), dim3(BLOCK), st, b, d);
  if (d.n_list) AM355_LAUNCH_INDEPENDENT(kd_items_sort, dgrid(d.n_list), dim3(BLOCK), st, d);
  step("items");
  // ---- maps: touched keys, kept records, placeholders (a batch without map rows touches no key: n_kept = n_place = 0 as filled) ----
  const bool all_kernels = getenv("AM355_DELTA_ALL_KERNELS") != nullptr;   // (tests, A/B: read per call)
  if (!d.list_only || all_kernels) {
    AM355_LAUNCH_INDEPENDENT(kd_slots, dgrid(cap), dim3(BLOCK), st, b, d);
    step("slots");
    AM355_LAUNCH_INDEPENDENT(kd_map_records, dgrid(d.n_map + 1), dim3(BLOCK), st, b, ir, d);
    step("map records");
    AM355_LAUNCH_INDEPENDENT(kd_placeholders, dgrid(cap + 1), dim3(BLOCK), st, d);
    step("placeholders");
    exclusive_scan_u32(d.keep, d.keep_ex, d.n_map + 1, &d.counts->n_kept, d.scan_ws, st);
    exclusive_scan_u32(d.place, d.place_ex, cap + 1, &d.counts->n_place, d.scan_ws, st);
    step("scans");
  }
  // ---- a small batch: the second half in one workgroup, behind the first without a word from the host in between (kd_edit_small) ----
  const bool no_small = getenv("AM355_DELTA_NO_SMALL") != nullptr;   // (tests, A/B: read per call)
  // (for batches the caller knows to be plain list edits: a batch with map rows has records to order, which this workgroup leaves to the
  //  host -- an attempt that ends so costs a round trip more than it could have saved)
  if (d.sig && d.list_only && !check_only && !no_small && d.n_new <= PART_LDS_MAX && getenv("AM355_DELTA_NO_LDS") == nullptr) {
    hipLaunchKernelGGL(kd_edit_small, dim3(1), dim3(BLOCK), 0, st, b, d);
    step("edit small");
    if (d.host_edit) {
      // (the kernel hands the tables over and signals itself)
      if (wait_host_signal(&d.sig->delta_mid_seq, d.sig_seq, st)) *hc = *(const DeltaCounts*)d.sig->delta_mid;
      else { (void)hipMemcpyAsync(hc, d.counts, sizeof(DeltaCounts), hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); d.host_edit = nullptr; }
    } else {
      if (before_end) {
        DeltaCounts none{};   // (no map records in what kd_edit_small serves)
        before_end(grow_user, &none, d.edit_cap);
      }
      read_counts(&d.sig->delta_mid_seq, d.sig->delta_mid);   // (the slot of the first half: when the work was left to the host, that is what this is)
    }
    if (hc->flags || !hc->deferred) return;
    d.host_edit = nullptr;
    hc->deferred = 0;   // (not served there: the counters are the first half's, the tables untouched -- on as for a larger batch)
  } else {
    d.host_edit = nullptr;   // (only kd_edit_small hands tables over)
    read_counts(d.sig ? &d.sig->delta_mid_seq : nullptr, d.sig ? d.sig->delta_mid : nullptr);
  }
  if (hc->flags || check_only) return;

  // ---- list edits: dominance counts by binary partitions on the time bits, most significant first ----
  const uint32_t m = hc->n_items;
  // edit records <= one per item + the extra values of conflicted elements + the sentinel: the table must hold them BEFORE kd_edit_pack
  // writes (the block has room for one record per new row only)
  const size_t rec_bound = (size_t)m + hc->rec_extra + 2;
  if (rec_bound > d.edit_cap) {
    am355_ir_edit* big = grow_edit ? grow_edit(grow_user, rec_bound) : nullptr;
    if (!big) { hc->flags |= AM355_F_UNSUPPORTED; hc->reason = DR_EDIT_TABLE; return; }
    d.edit = big;
    d.edit_cap = (uint32_t)std::min<size_t>(rec_bound, 0xffffffffu);
  }
  int cur = 0;
  if (m) {
    const bool no_lds = getenv("AM355_DELTA_NO_LDS") != nullptr;  // (tests: the level-by-level version on small batches too)
    // (Round 6 tried all levels in ONE 1024-thread workgroup over HBM for up to 32 k items, profiles/r06_apply_partition_wg.txt: 35 us per
    // level against ~15 for the three launches -- one CU's memory pipeline is no match for 256, and the device-side cost of a dependent
    // kernel boundary is only ~1.5-2 us, MI355X_MICROARCH.md "boundary". Taken out again.)
    const bool no_big = getenv("AM355_DELTA_NO_BIG_LDS") != nullptr;   // (tests, A/B: the level-by-level version for 1 k - 11 k items)
    const bool no_tiles = getenv("AM355_DELTA_NO_TILES") != nullptr;   // (tests, A/B: the versions below for 1 k - 64 k items)
    bool indexed = false;
    if (m <= PART_LDS_MAX && !no_lds) {
      hipLaunchKernelGGL(kd_partition_lds, dim3(1), dim3(BLOCK), 0, st, d, m, d.bits_new);
      cur = (int)(d.bits_new & 1u);
    } else if (m <= DOM_ITEMS_MAX && !no_lds && !no_tiles) {
      const uint32_t nt = (m + DOM_TILE - 1) / DOM_TILE;
      hipLaunchKernelGGL(kd_dom_tiles, dim3(nt), dim3(DOM_TILE * DOM_SPLIT), 0, st, d, m);
      hipLaunchKernelGGL(kd_dom_cross, dim3(nt, nt), dim3(BLOCK), 0, st, d, m);
      AM355_LAUNCH_INDEPENDENT(kd_dom_scatter, dgrid(m), dim3(BLOCK), st, b, d, m);
      cur = 1;
      indexed = true;
    } else if (m <= PART_BIG_MAX && !no_lds && !no_big && d.bits_new) {
      hipLaunchKernelGGL(kd_partition_lds_big, dim3(1), dim3(PART_BIG_THREADS), 0, st, b, d, m, d.bits_new);
      cur = (int)(d.bits_new & 1u);
    } else {
      if (d.bits_new) AM355_LAUNCH_INDEPENDENT(kd_bit_flags, dgrid(m + 1), dim3(BLOCK), st, d, cur, m, d.bits_new - 1);
      for (int bit = (int)d.bits_new - 1; bit >= 0; bit--) {
        exclusive_scan2_u32(d.zf, d.zf_ex, nullptr, d.zw, d.zw_ex, nullptr, m + 1, d.scan_ws, st);
        AM355_LAUNCH_INDEPENDENT(kd_partition, dgrid(m), dim3(BLOCK), st, d, cur, m, (uint32_t)bit);
        cur ^= 1;
      }
    }
    if (!indexed) AM355_LAUNCH_INDEPENDENT(kd_edit_index, dgrid(m), dim3(BLOCK), st, b, d, cur, m);
  }
  step("edit index");
  AM355_LAUNCH_INDEPENDENT(kd_edit_runs, dgrid(m + 1), dim3(BLOCK), st, b, d, cur, m);
  exclusive_scan2_u32(d.e_head, d.e_head_ex, nullptr, d.e_val, d.e_val_ex, nullptr, m + 1, d.scan_ws, st);
  step("edit runs");
  AM355_LAUNCH_INDEPENDENT(kd_edit_pack, dgrid(m + 1), dim3(BLOCK), st, b, d, cur, m);
  step("edit pack");

  // ---- map records in patch order ----
  const uint32_t nm = hc->n_kept + hc->n_place;
  if (nm) {
    AM355_LAUNCH_INDEPENDENT(kd_map_pairs, dgrid(std::max(d.n_map, cap)), dim3(BLOCK), st, b, d);
    step("map pairs");
    int res = radix_sort_pairs(d.pair_key[0], d.pair_val[0], d.pair_key[1], d.pair_val[1], nm, 0, (int)d.bits_new + dbits_for(d.n_obj), d.sort_ws, st);
    AM355_LAUNCH_INDEPENDENT(kd_map_out, dgrid(nm), dim3(BLOCK), st, b, ir, d, (const uint64_t*)d.pair_key[res], (const uint32_t*)d.pair_val[res], nm);
    step("map out");
  }
  if (before_end) {
    const DeltaCounts mid = *hc;
    before_end(grow_user, &mid, rec_bound);
  }
  read_counts(d.sig ? &d.sig->delta_end_seq : nullptr, d.sig ? d.sig->delta_end : nullptr);
  hc->n_items = m;
}

// ---------------------------------------------------------------------------------------------------------
// objectMeta.children of a property, from the history of the rows on it (see am355_delta.h)
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t KH_ROWS_MAX = 96;

// the property that holds object t: rows on the same map key of the same object, or on the same list element
__global__ __launch_bounds__(BLOCK) void kh_collect(MergeBufs b, PatchIR ir, const uint32_t* __restrict__ targets, uint32_t n, uint32_t* __restrict__ lists,
                                                    uint32_t* __restrict__ counts) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  const OpCols& o = b.ops;
  uint8_t kind = b.kind[g];
  if (kind == K_NONE || kind == K_FOREIGN) return;
  const bool map_op = kind == K_MAP || (kind == K_DEL && o.key_len[g] != NONE32);
  for (uint32_t t = 0; t < n; t++) {
    uint32_t m = ir.obj[targets[t]].make_row;
    bool hit;
    if (o.key_len[m] != NONE32) hit = map_op && b.obj_row[g] == b.obj_row[m] && same_key(b, g, m);
    else {
      uint32_t el = o.insert[m] ? m : b.ref_row[m];
      hit = el != NONE32 && (g == el || (!map_op && !o.insert[g] && b.ref_row[g] == el));
    }
    if (hit) {
      uint32_t k = atomicAdd(&counts[t], 1u);
      if (k < KH_ROWS_MAX) lists[t * KH_ROWS_MAX + k] = g;
    }
  }
}

__global__ __launch_bounds__(WAVE) void kh_simulate(MergeBufs b, PatchIR ir, DeltaBufs d, const uint32_t* __restrict__ targets, uint32_t n,
                                                    const uint32_t* __restrict__ lists, const uint32_t* __restrict__ counts, uint32_t* __restrict__ state,
                                                    uint32_t* __restrict__ values) {
  uint32_t t = gtid();
  if (t >= n) return;
  const OpCols& o = b.ops;
  const uint32_t c = counts[t];
  state[2 * t] = KH_UNKNOWN;
  state[2 * t + 1] = 0;
  if (c > KH_ROWS_MAX) { state[2 * t + 1] = 1; return; }  // (state[2t + 1] of an unknown answer: why -- diagnostics only)
  const bool map_prop = o.key_len[ir.obj[targets[t]].make_row] != NONE32;
  uint32_t rows[KH_ROWS_MAX], death[KH_ROWS_MAX], by_id[KH_ROWS_MAX];
  for (uint32_t i = 0; i < c; i++) {  // ascending row number = time
    uint32_t r = lists[t * KH_ROWS_MAX + i], j = i;
    for (; j > 0 && rows[j - 1] > r; j--) rows[j] = rows[j - 1];
    rows[j] = r;
  }
  for (uint32_t i = 0; i < c; i++) {  // ascending op id
    unsigned long long id = pack_id(o.id_ctr[rows[i]], o.id_actor[rows[i]]);
    uint32_t j = i;
    for (; j > 0 && pack_id(o.id_ctr[rows[by_id[j - 1]]], o.id_actor[rows[by_id[j - 1]]]) > id; j--) by_id[j] = by_id[j - 1];
    by_id[j] = i;
    death[i] = NONE32;
  }
  for (uint32_t j = 0; j < c; j++) {  // first successor of every row
    uint32_t g = rows[j];
    for (uint32_t k = 0; k < o.pred_num[g]; k++) {
      uint32_t pc = o.pred_ctr[o.pred_first[g] + k], pa = o.pred_actor[o.pred_first[g] + k];
      for (uint32_t i = 0; i < j; i++)
        if (o.id_ctr[rows[i]] == pc && o.id_actor[rows[i]] == pa && death[i] > g) death[i] = g;
    }
  }
  bool unknown = false;
  uint32_t listed[(KH_ROWS_MAX + 31) / 32] = {};  // children[key]: bit q = the row by_id[q] is listed
  auto none_listed = [&]() { uint32_t any = 0; for (uint32_t w = 0; w < (KH_ROWS_MAX + 31) / 32; w++) any |= listed[w]; return any == 0; };
  if (d.T_doc) {
    // The lineage began with Backend.load: objectMeta came from ONE pass over the document's rows in id order, every row judged by
    // the successors it had in the document (documentPatch, new.js:1604-1635 -> updatePatchProperty :893-931). A make row first puts
    // itself into children[key] (:894-897) -- overwritten or not --, which makes the list non-empty for the refresh that follows.
    uint32_t seen[(KH_ROWS_MAX + 31) / 32] = {};
    bool has_child = false;
    for (uint32_t q = 0; q < c; q++) {
      const uint32_t i = by_id[q], r = rows[i];
      if (r >= d.T_doc || b.kind[r] == K_DEL) continue;  // (a deletion is no row of a document)
      const uint32_t a = o.action[r];
      const bool is_make = (a & 1u) == 0;
      if (death[i] >= d.T_doc) {
        if (is_make) has_child = true;
        if (a == 1 || is_make) seen[q >> 5] |= 1u << (q & 31);
      }
      if (is_make || has_child || !none_listed())
        for (uint32_t w = 0; w < (KH_ROWS_MAX + 31) / 32; w++) listed[w] = seen[w];
    }
  }
  // (an op stream ends with the scheduling pass, and with the call of applyChanges)
  auto stream_begins_at = [&](uint32_t row) {
    uint32_t lo = 0, hi = d.n_breaks;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.breaks[mid] < row) lo = mid + 1; else hi = mid; }
    return lo < d.n_breaks && d.breaks[lo] == row;
  };
  uint32_t call_first = 0;  // rows[call_first .. j]: the ops of the merge call row j belongs to
  for (uint32_t j = 0; j < c && !unknown; j++) {
    const uint32_t g = rows[j];
    if (g < d.T_doc) { call_first = j + 1; continue; }  // (no call visited this row: it came with the document)
    const bool is_del = b.kind[g] == K_DEL;
    if (!map_prop && o.insert[g] && j > 0) { unknown = true; state[2 * t + 1] = 2; break; }
    // Ops of one actor on the property that follow each other in the op stream share ONE merge call -- one visit, with all of them
    // in place -- unless the next one overwrites an op the call already holds (new.js:1092-1117 "Collect several updates to the same
    // key / list element": e.g. two increments of one counter in a change). A deletion among them is not modelled.
    if (j + 1 < c && rows[j + 1] == g + 1 && o.id_actor[g + 1] == o.id_actor[g]) {
      bool overwrites = false;
      for (uint32_t q2 = call_first; q2 <= j; q2++)
        for (uint32_t k = 0; k < o.pred_num[g + 1]; k++)
          overwrites = overwrites || (o.pred_ctr[o.pred_first[g + 1] + k] == o.id_ctr[rows[q2]] && o.pred_actor[o.pred_first[g + 1] + k] == o.id_actor[rows[q2]]);
      if (is_del) { unknown = true; state[2 * t + 1] = 3; break; }
      if (!overwrites) {
        // (where the op streams of EARLIER calls began is not known when the staged changes were replayed in one go)
        if (!d.breaks_exact && g + 1 < d.T0) { unknown = true; state[2 * t + 1] = 4; break; }
        if (!stream_begins_at(g + 1)) {
          if (b.kind[g + 1] == K_DEL) { unknown = true; state[2 * t + 1] = 3; break; }
          continue;  // visited together with the next op
        }
      }
    }
    const uint32_t call_row0 = rows[call_first];  // the call's ops are the rows call_row0 .. g (consecutive, all on this property)
    call_first = j + 1;
    // a call that goes on to a greater key of the object leaves the rows above a threshold unvisited (see kd_slots)
    bool cont = false;
    unsigned long long thr = pack_id(o.id_ctr[g], o.id_actor[g]);
    if (map_prop && g + 1 < b.n_ops) {
      uint32_t nx = g + 1;
      const bool new_pass = stream_begins_at(nx);
      uint8_t kn = b.kind[nx];
      cont = !new_pass && (kn == K_MAP || kn == K_DEL) && o.key_len[nx] != NONE32 && !o.insert[nx] && o.id_actor[nx] == o.id_actor[g] && same_obj(b, nx, g) &&
             key_less_utf16(b, g, nx);
      if (cont && is_del) {
        thr = 0;
        for (uint32_t k = 0; k < o.pred_num[g]; k++) {
          unsigned long long pid = pack_id(o.pred_ctr[o.pred_first[g] + k], o.pred_actor[o.pred_first[g] + k]);
          thr = pid > thr ? pid : thr;
        }
      }
    }
    uint32_t seen[(KH_ROWS_MAX + 31) / 32] = {};  // visible values among the rows visited so far
    bool has_child = false;
    for (uint32_t q = 0; q < c; q++) {
      const uint32_t i = by_id[q], r = rows[i];
      if (r > g || b.kind[r] == K_DEL) continue;  // not there yet / a deletion leaves no row
      if (cont && !(r >= call_row0 && r <= g) && pack_id(o.id_ctr[r], o.id_actor[r]) > thr) {
        // (when the staged changes were replayed in one go, whether an EARLIER call went on to the next row is not known -- a call of
        // applyChanges or a scheduling pass may have ended between them --: it only matters when a row it would have skipped is visible)
        if (!d.breaks_exact && g + 1 < d.T0 && death[i] > g) { unknown = true; state[2 * t + 1] = 4; }
        continue;
      }
      if (r == g || death[i] > g) {
        uint32_t a = o.action[r];
        if ((a & 1u) == 0) has_child = true;
        if (a == 1 || (a & 1u) == 0) seen[q >> 5] |= 1u << (q & 31);
      }
      if (has_child || !none_listed())
        for (uint32_t w = 0; w < (KH_ROWS_MAX + 31) / 32; w++) listed[w] = seen[w];
    }
  }
  if (unknown) return;
  uint32_t nl = 0;
  for (uint32_t q = 0; q < c; q++)
    if (listed[q >> 5] >> (q & 31) & 1u) {
      if (nl == KH_VALUES_MAX) { state[2 * t + 1] = 5; return; }  // (KH_UNKNOWN)
      uint32_t r = rows[by_id[q]];
      values[(size_t)t * 2 * KH_VALUES_MAX + 2 * nl] = o.id_ctr[r];
      values[(size_t)t * 2 * KH_VALUES_MAX + 2 * nl + 1] = o.id_actor[r];
      nl++;
    }
  state[2 * t] = nl ? (uint32_t)KH_LIVE : (uint32_t)KH_DEAD;
  state[2 * t + 1] = nl;
}

int delta_key_history(MergeBufs& b, PatchIR& ir, DeltaBufs& d, const uint32_t* objects, uint32_t n, KeyHistory* out, hipStream_t st) {
  if (!n) return 0;
  uint32_t* dev = nullptr;
  const size_t words = (size_t)n * (KH_ROWS_MAX + 4 + 2 * KH_VALUES_MAX);
  if (hipMalloc((void**)&dev, 4 * words) != hipSuccess) return -1;
  uint32_t *targets = dev, *counts = dev + n, *state = dev + 2 * (size_t)n, *values = dev + 4 * (size_t)n, *lists = values + (size_t)n * 2 * KH_VALUES_MAX;
  (void)hipMemsetAsync(dev, 0, 4 * words, st);
  (void)hipMemcpyAsync(targets, objects, 4 * (size_t)n, hipMemcpyHostToDevice, st);
  (void)hipStreamSynchronize(st);  // (pageable source)
  if (b.n_ops) AM355_LAUNCH_INDEPENDENT(kh_collect, dgrid(b.n_ops), dim3(BLOCK), st, b, ir, (const uint32_t*)targets, n, lists, counts);
  AM355_LAUNCH_INDEPENDENT(kh_simulate, dim3((n + WAVE - 1) / WAVE), dim3(WAVE), st, b, ir, d, (const uint32_t*)targets, n, (const uint32_t*)lists, (const uint32_t*)counts, state,
                           values);
  std::vector<uint32_t> h((size_t)n * (2 + 2 * KH_VALUES_MAX));
  (void)hipMemcpyAsync(h.data(), state, 4 * h.size(), hipMemcpyDeviceToHost, st);  // (state | values: contiguous)
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(dev);
  if (e != hipSuccess) return -1;
  for (uint32_t i = 0; i < n; i++) {
    if (h[2 * (size_t)i] == KH_UNKNOWN && getenv("AM355_TRACE")) fprintf(stderr, "key history of object %u: unknown (reason %u)\n", objects[i], h[2 * (size_t)i + 1]);
    out[i].state = (uint8_t)h[2 * (size_t)i];
    out[i].n = h[2 * (size_t)i + 1];
    for (uint32_t k = 0; k < out[i].n && k < KH_VALUES_MAX; k++) {
      out[i].ctr[k] = h[2 * (size_t)n + (size_t)i * 2 * KH_VALUES_MAX + 2 * k];
      out[i].actor[k] = h[2 * (size_t)n + (size_t)i * 2 * KH_VALUES_MAX + 2 * k + 1];
    }
  }
  return 0;
}

}  // namespace am355
