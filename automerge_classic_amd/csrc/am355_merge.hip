// Stage 2 of the replay engine: per-object op-set merge, RGA list ordering, multi-value register resolution and
// whole-document patch IR, as data-parallel kernels over the fixed-width op rows produced by stage 1.
//
// Reference semantics reproduced (automerge-classic, paths relative to the reference tree; derived spec in
// SURVEY.md Appendix B):
//   op placement      backend/new.js:50-192 seekWithinBlock (object order, key order, RGA insertion rule :144-163)
//   pred -> succ      backend/new.js:1173-1188, 1205-1217 (del = succ entries only), 1252-1258 (pred must match)
//   visibility        backend/new.js:904 (overwritten iff succNum > 0), 1622-1626 (list index bookkeeping)
//   counters          backend/new.js:937-967
//   list edits        backend/new.js:983-1033 (whole-document branches), 747-782 appendEdit (multi-insert runs)
//   map props         backend/new.js:1035-1039
//
// The reference merges one op-run at a time into RLE-compressed 600-op blocks (new.js:1304-1380); nothing of
// that machinery exists here. Instead: op ids resolve to rows by arithmetic (each change owns a contiguous
// counter range), succ counts are atomics, the RGA order of every list is the pre-order of the insertion tree
// (parent = reference element, siblings by descending op id) obtained by one radix sort + Euler-tour list
// ranking, and visibility / list indexes / edit positions are prefix sums over that order.
#include "am355_merge.h"
#include "am355_prims.h"
#include "am355_rows.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace am355 {

// ---------------------------------------------------------------------------------------------------------
// k_resolve: one lane per op row.  Object / element / pred resolution and validation, succ counting.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_resolve(MergeBufs b) {
  wave_priority_high();
  uint32_t g = b.first_row + gtid();   // (first_row > 0: the rows in front were resolved by an earlier replay and kept, see MergeBufs)
  const bool in_range = g < b.n_ops;
  if (b.seed_list_inc && gtid() == 0) atomicAdd(&b.counts->n_list_inc, b.seed_list_inc);
  const OpCols& o = b.ops;
  uint32_t err = 0;
  uint32_t a = in_range ? o.action[g] : 1;
  bool ins = in_range && o.insert[g] != 0;
  unsigned long long my_id = in_range ? pack_id(o.id_ctr[g], o.id_actor[g]) : 0;

  // object: nearly every row of a wavefront names the same object (a change edits one Text, a whole document may be one
  // Text), so one lane resolves it for the wave and only rows naming another object search the span table themselves
  uint32_t oa = in_range ? o.obj_actor[g] : NONE32, oc = in_range ? o.obj_ctr[g] : 0;
  bool has_obj = oa != NONE32;
  uint32_t orow = NONE32;
  {
    unsigned long long m = __ballot(has_obj);
    uint32_t lane = threadIdx.x & (WAVE - 1);
    uint32_t leader = m ? (uint32_t)__ffsll(m) - 1 : 0;
    uint32_t la = __shfl(oa, (int)leader), lc = __shfl(oc, (int)leader);
    uint32_t lrow = (m && lane == leader) ? row_of(b, oa, oc) : NONE32;
    lrow = __shfl(lrow, (int)leader);
    if (has_obj) orow = (oa == la && oc == lc) ? lrow : row_of(b, oa, oc);
  }
  if (!in_range) return;
  if (b.shard_world > 1 && shard_owner(oa, oc, b.shard_world) != b.shard_rank) {
    b.obj_row[g] = NONE32;
    b.ref_row[g] = NONE32;
    b.kind[g] = K_FOREIGN;
    return;
  }
  bool list_obj = false;
  if (has_obj) {
    if (orow == NONE32 || (o.action[orow] & 1)) { err |= F_UNKNOWN_OBJECT; orow = NONE32; }
    else {
      uint32_t oact = o.action[orow];
      list_obj = (oact == 2 || oact == 4);
      if (pack_id(o.id_ctr[orow], o.id_actor[orow]) >= my_id) err |= F_UNSUPPORTED;
    }
  }
  bool has_str = o.key_len[g] != NONE32, has_elem = o.key_ctr[g] != NONE32;
  uint8_t kind = K_NONE;
  uint32_t ref = NONE32;
  const uint32_t ka = o.key_actor[g], kc = o.key_ctr[g];
  if (has_str == has_elem) {
    err |= has_str ? F_UNSUPPORTED : F_BAD_ROW;
  } else if (has_str) {
    if (list_obj || ins) err |= F_UNSUPPORTED;  // the reference would crash building the patch
    kind = a == 3 ? K_DEL : K_MAP;
  } else {
    if (!list_obj) err |= F_UNSUPPORTED;
    if (kc != 0) {
      // typing: the reference element of an insert is nearly always the op right before it (same change, previous row)
      if (g > 0 && o.id_ctr[g - 1] == kc && o.id_actor[g - 1] == ka) ref = g - 1;
      else ref = row_of(b, ka, kc);
      // Every field read of the reference row is a random 64-byte line (the rows of a deleted element are anywhere): only
      // `insert` and the object are looked at. Its id IS (kc, ka) -- that is how the row was found -- and an insert row with a
      // string key is flagged by its own lane (F_UNSUPPORTED above), so neither needs loading here.
      if (ref == NONE32 || !o.insert[ref] || !same_obj(b, ref, g)) { err |= F_BAD_ELEM; ref = NONE32; }
      else if (pack_id(kc, ka) >= my_id) err |= F_UNSUPPORTED;
    } else if (!ins) {
      err |= F_UNSUPPORTED;  // non-insert on _head
    }
    if (ins) {
      kind = K_LIST_INS;
      if (o.pred_num[g]) err |= F_BAD_PRED;  // an insert never finds its preds (new.js:1252-1258)
    } else {
      kind = a == 3 ? K_DEL : K_LIST_UPD;
    }
  }
  if (a == 3 && o.pred_num[g] == 0) err |= F_UNSUPPORTED;

  uint32_t np = o.pred_num[g], pf = o.pred_first[g];
  uint32_t counter_row = NONE32;
  unsigned long long counter_id = 0;
  for (uint32_t j = 0; j < np && !(err & (F_BAD_PRED | F_BAD_ROW)); j++) {
    uint32_t pa = o.pred_actor[pf + j], pc = o.pred_ctr[pf + j];
    for (uint32_t k = 0; k < j; k++)
      if (o.pred_actor[pf + k] == pa && o.pred_ctr[pf + k] == pc) err |= F_BAD_PRED;  // second copy never matches
    // deleting / overwriting a list element names the element's own insert op as pred: already resolved as `ref`
    const bool pred_is_ref = ref != NONE32 && pa == ka && pc == kc;
    uint32_t pr = pred_is_ref ? ref : row_of(b, pa, pc);
    // (pred == the element's own insert row: its object was compared above, and an insert row whose action is `del` is flagged by its
    // own lane -- BAD_PRED or UNSUPPORTED -- so its action needs no load)
    if (pr == NONE32 || (!pred_is_ref && (o.action[pr] == 3 || !same_obj(b, pr, g)))) { err |= F_BAD_PRED; continue; }
    bool same_slot;
    if (has_str) same_slot = same_key(b, pr, g);
    else if (pred_is_ref) same_slot = !ins;
    else if (o.key_len[pr] != NONE32 || ref == NONE32 || ins) same_slot = false;
    else if (pr == ref) same_slot = true;
    else if (!o.insert[pr] && o.key_actor[pr] == ka && o.key_ctr[pr] == kc) same_slot = true;  // an earlier update of the same element
    else same_slot = elem_of(b, pr) == ref;
    if (!same_slot) { err |= F_BAD_PRED; continue; }
    if (pack_id(pc, pa) >= my_id) err |= F_UNSUPPORTED;
    atomicAdd(&b.succ_cnt[pr], 1u);
    if (a == 5 && o.action[pr] == 1 && (o.val_tl[pr] & 15) == 8) {
      // inc: it feeds the counter `set` among its preds; with several (conflicting counters) the one with the greatest
      // op id takes it -- counterStates[succOp] is overwritten by each later counter row (new.js:944-950)
      unsigned long long pid = pack_id(pc, pa);
      if (counter_row == NONE32 || pid > counter_id) { counter_row = pr; counter_id = pid; }
    }
  }
  if (a == 5 && !(err & (F_BAD_PRED | F_BAD_ROW))) {
    long long v;
    if (counter_row == NONE32) err |= F_BAD_COUNTER;  // increment operation for unknown counter (new.js:954-956)
    else if (!int_value(b, g, v)) err |= F_UNSUPPORTED;
    else {
      atomicAdd(&b.inc_cnt[counter_row], 1u);
      atomicAdd(&b.inc_sum[counter_row], (unsigned long long)v);
      atomicMax(&b.last_inc[counter_row], my_id);
      if (!has_str) atomicAdd(&b.counts->n_list_inc, 1u);  // (rare: tells k_emit that lists may hold counters completed by increments)
    }
  }
  b.obj_row[g] = orow;
  b.ref_row[g] = ref;
  b.kind[g] = kind;
  if (err) atomicOr(&b.counts->flags, err);
}

// ---------------------------------------------------------------------------------------------------------
// k_emit: one lane per op row, after all succ counts are final.  Classifies visible values.
// ---------------------------------------------------------------------------------------------------------
// Wave-aggregated append: one atomic per wavefront instead of one per lane. Must be reached by every lane.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool want) {
  unsigned long long m = __ballot(want);
  uint32_t lane = threadIdx.x & (WAVE - 1);
  uint32_t leader = m ? (uint32_t)__ffsll(m) - 1 : 0;
  uint32_t base = 0;
  if (m && lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = __shfl(base, (int)leader);
  return base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
}

// Workgroup-aggregated append: ONE atomic per workgroup of four wavefronts (same-word device atomics serialise at ~12 ns each: a map
// batch of 80 k rows, every row an emission, spent 15 of k_emit's 20 us in 1252 of them). Must be reached by every thread; the order of
// the appended items among workgroups is arbitrary, as with wave_append. s: BLOCK / WAVE + 2 words of LDS.
__device__ __forceinline__ uint32_t block_append(uint32_t* counter, bool want, uint32_t* s) {
  const unsigned long long m = __ballot(want);
  const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  __syncthreads();  // (s may still be read from a previous use)
  if (lane == 0) s[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t k = 0; k < BLOCK / WAVE; k++) { uint32_t c = s[k]; s[k] = total; total += c; }
    s[BLOCK / WAVE] = total ? atomicAdd(counter, total) : 0u;
    s[BLOCK / WAVE + 1] = total;  // (how many the workgroup appended: read by the caller before its next use of s)
  }
  __syncthreads();
  return s[BLOCK / WAVE] + s[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1));
}

__global__ __launch_bounds__(BLOCK) void k_emit(MergeBufs b) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  __shared__ uint32_t s_app[BLOCK / WAVE + 2];
  uint32_t g = gtid();
  const OpCols& o = b.ops;
  bool in_range = g < b.n_ops;
  uint8_t kind = in_range ? b.kind[g] : (uint8_t)K_NONE;
  // (a kept row of an earlier replay may carry that replay's verdict "its own value is visible": judged again below)
  const bool was_vis = kind == K_LIST_INS_VIS;
  if (was_vis) kind = K_LIST_INS;
  uint32_t a = in_range ? o.action[g] : 1;
  bool live = kind != K_NONE && kind != K_DEL;
  bool is_make = live && (a & 1) == 0;  // (a foreign make row too: the object table is the same on every rank)
  if (kind == K_FOREIGN) live = false;
  bool vis = live && b.succ_cnt[g] == 0;
  // (one word for the whole kernel: without increments on list elements -- every ordinary document -- the invisible `set` rows of lists
  // cost no second look; k_resolve has completed)
  const bool any_list_inc = b.counts->n_list_inc != 0;
  bool want_map = false, want_ins = false, want_upd = false;
  unsigned long long trig = 0;
  uint32_t el = NONE32;
  if (live && kind == K_MAP) {
    uint32_t tl = o.val_tl[g];
    if (a == 1) {
      if (vis) { want_map = true; trig = pack_id(o.id_ctr[g], o.id_actor[g]); }
      else if ((tl & 15) == 8 && b.inc_cnt[g] == b.succ_cnt[g]) { want_map = true; trig = b.last_inc[g]; }  // every succ is an inc
    } else if ((a & 1) == 0 && vis) {
      want_map = true;
      trig = pack_id(o.id_ctr[g], o.id_actor[g]);
    }
  } else if (live) {
    bool valued = (a == 1) || (a & 1) == 0;
    want_ins = kind == K_LIST_INS;
    bool quirk = false;
    if (kind == K_LIST_INS && was_vis != (vis && valued)) b.kind[g] = (vis && valued) ? K_LIST_INS_VIS : K_LIST_INS;
    if (vis && valued) {
      if (kind == K_LIST_INS) {}
      else {
        el = b.ref_row[g];
        if (el != NONE32) {
          atomicAdd(&b.val_cnt[el], 1u);
          want_upd = true;
        }
      }
    } else if (vis) {
      // a visible row without a value (an increment, a link): it takes part in the element's edits by the reference's `remove` rule
      // (new.js:1026-1033) -- k_quirk_rows / k_list_edits
      quirk = true;
    } else if (any_list_inc && a == 1 && (o.val_tl[g] & 15) == 8 && b.inc_cnt[g] == b.succ_cnt[g]) {
      // a counter whose successors are all increments (new.js:937-965): one value of its element, listed where the LAST increment
      // stands among the element's rows -- it travels with the update rows, keyed by that increment's id (k_upd_keys), also when the
      // counter is the element's own insert row
      el = kind == K_LIST_INS ? g : b.ref_row[g];
      if (el != NONE32) {
        atomicAdd(&b.val_cnt[el], 1u);
        want_upd = true;
      }
      quirk = true;
    }
    if (quirk) atomicAdd(&b.counts->n_quirk, 1u);
  }
  uint32_t slot = block_append(&b.counts->n_map_emit, want_map, s_app);
  const uint32_t block_maps = s_app[BLOCK / WAVE + 1];  // map emissions of this workgroup (the same word for every thread)
  if (want_map) {
    b.em_row[slot] = g;
    b.em_trig[slot] = trig;
    // (monotone maximum: the plain read only filters; same-word device atomics cost ~12 ns each, serialised)
    if (o.key_len[g] > *(volatile uint32_t*)&b.counts->max_key_len) atomicMax(&b.counts->max_key_len, o.key_len[g]);
  }
  // what the keys of the emissions have in common (MapKeyStats): OR over the workgroup in LDS, ten atomics per workgroup that emits
  if (block_maps) {
    uint32_t w[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (want_map) {
      const uint32_t len = o.key_len[g];
      const uint8_t* kp = b.arena + o.key_off[g];
      w[0] = len; w[1] = ~len;
      // the first sixteen key bytes in two 8-byte loads (the arena has slack behind its last byte), bytes behind the key's end zeroed
      unsigned long long lo, hi;
      __builtin_memcpy(&lo, kp, 8);
      __builtin_memcpy(&hi, kp + 8, 8);
      if (len < 8) { lo &= len ? (~0ull >> (64 - 8 * len)) : 0ull; hi = 0; }
      else if (len < 16) hi &= len > 8 ? (~0ull >> (64 - 8 * (len - 8))) : 0ull;
      uint32_t q[4] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
      for (uint32_t k = 0; k < 4; k++) {
        uint32_t x = __builtin_bswap32(q[k]);  // first byte highest, as map_key_of packs them
        // (the UTF-16 order remap only touches lead bytes EE..F4: rare, byte by byte then)
        if (((x >> 24) >= 0xee) || (((x >> 16) & 0xff) >= 0xee) || (((x >> 8) & 0xff) >= 0xee) || ((x & 0xff) >= 0xee))
          x = utf16_order_byte(x >> 24) << 24 | utf16_order_byte((x >> 16) & 0xff) << 16 | utf16_order_byte((x >> 8) & 0xff) << 8 | utf16_order_byte(x & 0xff);
        w[2 + k] = x;
        w[6 + k] = ~x;
      }
    }
    __shared__ uint32_t s_or[10];
    if (threadIdx.x < 10) s_or[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t k = 0; k < 10; k++) {
      uint32_t x = w[k];
      for (int d = WAVE / 2; d >= 1; d >>= 1) x |= __shfl_xor(x, d);
      if ((threadIdx.x & (WAVE - 1)) == 0 && x) atomicOr(&s_or[k], x);
    }
    __syncthreads();
    if (threadIdx.x < 10 && s_or[threadIdx.x]) {
      uint32_t* dst = (uint32_t*)b.counts + MAP_KEY_STATS_WORD + threadIdx.x;
      if ((*(volatile uint32_t*)dst | s_or[threadIdx.x]) != *(volatile uint32_t*)dst) atomicOr(dst, s_or[threadIdx.x]);
    }
  }
  slot = block_append(&b.counts->n_list_upd, want_upd, s_app);
  if (want_upd) b.upd_row[slot] = g;
  // list insert rows are most of a text document, make rows must keep row order: both are compacted by prefix sums, whose
  // workgroup sums this kernel publishes (k_compact_rows rebuilds the positions: no scan launch in between)
  // (the totals -- Counts.n_list_ins, n_objects -- are written by the consumer's last thread: one atomic per workgroup on one
  // word would serialise 4 k workgroups behind each other)
  carry_publish(b.cs_ins, want_ins ? 1u : 0u, s_red);
  carry_publish(b.cs_make, is_make ? 1u : 0u, s_red);
}

// documents (doc_patch): insert rows to dense positions from a scanned flag array
__global__ __launch_bounds__(BLOCK) void k_ins_scatter(MergeBufs b, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos) {
  uint32_t g = gtid();
  if (g < b.n_ops && flag[g]) b.ins_row[pos[g]] = g;
}

__device__ __forceinline__ void object_table_entry(const MergeBufs& b, const PatchIR& ir, uint32_t g, bool is_make, uint32_t idx) {
  if (is_make) {
    ir.obj[idx] = am355_ir_object{b.ops.id_ctr[g], b.ops.id_actor[g], b.ops.action[g], 0, 0, 0, 0, g};
    b.obj_index[g] = idx;
  } else {
    b.obj_index[g] = NONE32;
  }
  if (g == 0) ir.obj[0] = am355_ir_object{0, 0, 0, 0, 0, 0, 0, NONE32};
}

// consumer of k_emit's carried scans: list insert rows -> dense list (ins_row), make rows -> object table (index 0 = _root)
__global__ __launch_bounds__(BLOCK) void k_compact_rows(MergeBufs b, PatchIR ir) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  uint32_t g = gtid();
  bool in_range = g < b.n_ops;
  uint8_t kind = in_range ? b.kind[g] : (uint8_t)K_NONE;
  bool live = kind != K_NONE && kind != K_DEL;
  bool is_make = live && (b.ops.action[g] & 1) == 0;
  bool want_ins = kind == K_LIST_INS || kind == K_LIST_INS_VIS;
  uint32_t pos = carry_prefix(b.cs_ins, want_ins ? 1u : 0u, s_red);
  uint32_t idx = carry_prefix(b.cs_make, is_make ? 1u : 0u, s_red) + 1;  // 0 is _root
  if (want_ins) b.ins_row[pos] = g;
  if (in_range) object_table_entry(b, ir, g, is_make, idx);
  if (g + 1 == b.n_ops) {
    b.counts->n_list_ins = pos + (want_ins ? 1u : 0u);
    b.counts->n_objects = idx - 1 + (is_make ? 1u : 0u);
    // (everything else in Counts up to here was written by k_resolve / k_emit, which have completed)
    if (b.sig) signal_host(b.sig->counts, (const uint32_t*)b.counts, MAP_KEY_STATS_WORD + sizeof(MapKeyStats) / 4, &b.sig->counts_seq, b.sig_seq);
  }
}

// ---------------------------------------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_object_table(MergeBufs b, const uint32_t* is_make_ex /* may be b.obj_index itself */, PatchIR ir) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  uint8_t kind = b.kind[g];
  bool is_make = kind != K_DEL && kind != K_NONE && (b.ops.action[g] & 1) == 0;
  object_table_entry(b, ir, g, is_make, is_make_ex[g] + 1);  // 0 is _root
}


// ---------------------------------------------------------------------------------------------------------
// map emissions: ordered by (object, key in UTF-16 code unit order, trigger op id) with stable LSD passes
// ---------------------------------------------------------------------------------------------------------
enum MapKeyMode { MK_TRIGGER, MK_LEN, MK_CHUNK, MK_OBJECT };

__device__ __forceinline__ uint64_t map_key_of(const MergeBufs& b, uint32_t e, int mode, uint32_t chunk, const uint32_t* __restrict__ obj_rank) {
  uint32_t g = b.em_row[e];
  uint64_t k = 0;
  if (mode == MK_TRIGGER) {
    unsigned long long t = b.em_trig[e];
    k = (uint64_t)(t >> 32) << b.bits_actor | (uint32_t)t;
  } else if (mode == MK_LEN) {
    k = b.ops.key_len[g];
  } else if (mode == MK_CHUNK) {
    const uint8_t* p = b.arena + b.ops.key_off[g];
    uint32_t len = b.ops.key_len[g], start = chunk * 8;
    for (uint32_t j = 0; j < 8; j++) {
      uint32_t x = start + j < len ? utf16_order_byte(p[start + j]) : 0;
      k = k << 8 | x;
    }
  } else {
    k = obj_index_of(b, b.obj_row[g]);
    if (obj_rank) k = obj_rank[k];  // save(): objects in ascending id order instead of creation order
  }
  return k;
}
__global__ __launch_bounds__(BLOCK) void k_map_keys(MergeBufs b, const uint32_t* __restrict__ perm, uint64_t* __restrict__ keys, uint32_t n,
                                                    int mode, uint32_t chunk, const uint32_t* __restrict__ obj_rank) {
  uint32_t i = gtid();
  if (i >= n) return;
  keys[i] = map_key_of(b, perm[i], mode, chunk, obj_rank);
}
// the same + the histogram of the first digit the sort behind it looks at (radix_sort_pairs, first_hist_done): one launch less per field
__global__ __launch_bounds__(BLOCK) void k_map_keys_hist(MergeBufs b, const uint32_t* __restrict__ perm, uint64_t* __restrict__ keys, uint32_t n,
                                                         int mode, uint32_t chunk, const uint32_t* __restrict__ obj_rank, int shift,
                                                         uint32_t* __restrict__ table, uint32_t n_tiles) {
  __shared__ uint32_t hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * SORT_TILE_ELEMS;
  for (uint32_t j = 0; j < SORT_TILE_ELEMS / BLOCK; j++) {
    const uint32_t i = base + j * BLOCK + threadIdx.x;
    if (i < n) {
      const uint64_t k = map_key_of(b, perm[i], mode, chunk, obj_rank);
      keys[i] = k;
      atomicAdd(&hist[(uint32_t)(k >> shift) & 0xff], 1u);
    }
  }
  __syncthreads();
  table[threadIdx.x * n_tiles + blockIdx.x] = hist[threadIdx.x];
}

// The same order for a handful of emissions (a root map with a few keys next to a large Text is the common document):
// the multi-pass radix sort would spend ~16 passes of 5 launches on them. One workgroup ranks every emission against all
// others instead. Order: (object, key bytes in UTF-16 order -- a prefix sorts first --, trigger op id).
constexpr uint32_t MAP_SORT_SMALL = 512;  // n comparisons per lane: beyond a few hundred the radix passes win
__device__ __forceinline__ bool map_emission_less(const MergeBufs& b, uint32_t ea, uint32_t eb, const uint32_t* __restrict__ obj_rank) {
  uint32_t ga = b.em_row[ea], gb = b.em_row[eb];
  uint32_t oa = obj_index_of(b, b.obj_row[ga]), ob = obj_index_of(b, b.obj_row[gb]);
  if (obj_rank) { oa = obj_rank[oa]; ob = obj_rank[ob]; }
  if (oa != ob) return oa < ob;
  const uint8_t *p = b.arena + b.ops.key_off[ga], *q = b.arena + b.ops.key_off[gb];
  uint32_t la = b.ops.key_len[ga], lb = b.ops.key_len[gb], n = la < lb ? la : lb;
  for (uint32_t k = 0; k < n; k++) {
    uint32_t x = utf16_order_byte(p[k]), y = utf16_order_byte(q[k]);
    if (x != y) return x < y;
  }
  if (la != lb) return la < lb;
  unsigned long long ta = b.em_trig[ea], tb = b.em_trig[eb];
  if (ta != tb) return ta < tb;
  return ea < eb;
}
// What a comparison of two emissions looks at, staged once per emission (LDS): object, key length, the key's first sixteen bytes as two
// big-endian words in UTF-16 order (zero behind the key's end), trigger id. A rank by comparison with every other emission then reads
// LDS: from the rows, every one of a thread's n comparisons was a chain of dependent loads (64 emissions: 137 us in one workgroup).
struct MapStaged {
  uint32_t oi, len;
  unsigned long long k0, k1, trig;
};
__device__ __forceinline__ MapStaged map_stage(const MergeBufs& b, uint32_t e, const uint32_t* __restrict__ obj_rank) {
  const uint32_t g = b.em_row[e];
  MapStaged m;
  m.oi = obj_index_of(b, b.obj_row[g]);
  if (obj_rank) m.oi = obj_rank[m.oi];
  m.len = b.ops.key_len[g];
  const uint8_t* p = b.arena + b.ops.key_off[g];
  m.k0 = m.k1 = 0;
  for (uint32_t j = 0; j < 16; j++) {
    const unsigned long long x = j < m.len ? utf16_order_byte(p[j]) : 0u;
    if (j < 8) m.k0 = m.k0 << 8 | x; else m.k1 = m.k1 << 8 | x;
  }
  m.trig = b.em_trig[e];
  return m;
}
// a < b in (object, key bytes in UTF-16 order -- a prefix sorts first --, trigger id, emission) order; ea / eb: their emission indexes
__device__ __forceinline__ bool map_staged_less(const MergeBufs& b, const MapStaged& x, uint32_t ea, const MapStaged& y, uint32_t eb, const uint32_t* __restrict__ obj_rank) {
  if (x.oi != y.oi) return x.oi < y.oi;
  // (differing prefixes decide: where they first differ either both keys have a byte, or the shorter one is padded with zero and sorts first)
  if (x.k0 != y.k0) return x.k0 < y.k0;
  if (x.k1 != y.k1) return x.k1 < y.k1;
  if (x.len > 16 || y.len > 16) return map_emission_less(b, ea, eb, obj_rank);   // (the bytes behind the sixteenth: from the rows)
  if (x.len != y.len) return x.len < y.len;   // (equal prefixes, both short: the longer one ends in zero bytes)
  if (x.trig != y.trig) return x.trig < y.trig;
  return ea < eb;
}
__global__ __launch_bounds__(BLOCK) void k_map_sort_small(MergeBufs b, uint32_t n, const uint32_t* __restrict__ obj_rank, uint32_t* __restrict__ perm) {
  __shared__ MapStaged s_m[MAP_SORT_SMALL];   // (every workgroup stages all n <= MAP_SORT_SMALL emissions: two per thread)
  for (uint32_t j = threadIdx.x; j < n && j < MAP_SORT_SMALL; j += BLOCK) s_m[j] = map_stage(b, j, obj_rank);
  __syncthreads();
  uint32_t i = gtid();
  if (i >= n) return;
  const MapStaged mine = s_m[i];
  uint32_t rank = 0;
  for (uint32_t j = 0; j < n; j++) rank += map_staged_less(b, s_m[j], j, mine, i, obj_rank) ? 1u : 0u;
  perm[rank] = i;
}

// The values of one key in op id order WITHOUT radix passes over the trigger ids (three passes of two launches for 22-bit ids): the key
// passes leave the emissions of one (object, key) next to each other in emission order; every emission ranks itself among them --
// a handful: the conflicting values of a register -- and takes its place. More than MAP_GROUP_MAX values on one key (thousands of
// actors assigning one key concurrently): Counts.map_group_big, and the host orders again with the trigger passes.
constexpr uint32_t MAP_GROUP_MAX = 256;
// (first8: the sort keys of the last pass when that pass was over the keys' first eight bytes -- in sorted order, one coalesced load:
// neighbours whose first eight bytes differ are told apart without a look at their rows; null: not available)
__global__ __launch_bounds__(BLOCK) void k_map_group_rank(MergeBufs b, const uint32_t* __restrict__ perm_in, uint32_t* __restrict__ perm_out, uint32_t n,
                                                          const uint64_t* __restrict__ first8) {
  uint32_t i = gtid();
  if (i >= n) return;
  const uint32_t e = perm_in[i], g = b.em_row[e], orow = b.obj_row[g];
  const uint64_t my8 = first8 ? first8[i] : 0;
  auto same = [&](uint32_t j) {
    if (first8 && first8[j] != my8) return false;
    const uint32_t g2 = b.em_row[perm_in[j]];
    return b.obj_row[g2] == orow && same_key(b, g2, g);
  };
  uint32_t s = i, t = i, steps = 0;
  while (s > 0 && steps <= MAP_GROUP_MAX && same(s - 1)) { s--; steps++; }
  while (t + 1 < n && steps <= MAP_GROUP_MAX && same(t + 1)) { t++; steps++; }
  // (the host orders the emissions again when it reads the flag; until then k_map_finish runs over THIS buffer: a valid entry, not
  // whatever the scratch held -- ADVICE r5)
  if (steps > MAP_GROUP_MAX) { b.counts->map_group_big = 1; perm_out[i] = e; return; }
  const unsigned long long mine = b.em_trig[e];
  uint32_t rank = 0;
  for (uint32_t j = s; j <= t; j++) {
    if (j == i) continue;
    const uint32_t e2 = perm_in[j];
    const unsigned long long other = b.em_trig[e2];
    rank += (other < mine || (other == mine && e2 < e)) ? 1u : 0u;
  }
  perm_out[s + rank] = e;
}

__global__ __launch_bounds__(BLOCK) void k_iota(uint32_t* __restrict__ v, uint32_t n) {
  uint32_t i = gtid();
  if (i < n) v[i] = i;
}

// one map emission: its record at output position i (e = emission at i, e_prev / e_next = its neighbours in output order, NONE32 at the ends)
__device__ __forceinline__ void map_finish_one(const MergeBufs& b, const PatchIR& ir, uint32_t i, uint32_t e, uint32_t e_prev, uint32_t e_next) {
  uint32_t g = b.em_row[e];
  const OpCols& o = b.ops;
  uint32_t a = o.action[g];
  uint32_t flags = 0;
  long long counter = 0;
  if (a == 1 && b.succ_cnt[g] != 0) {
    flags |= AM355_MAP_COUNTER;
    long long base = 0;
    if (!int_value(b, g, base)) atomicOr(&b.counts->flags, (uint32_t)F_BAD_LEB);
    counter = base + (long long)b.inc_sum[g];
  } else if ((a & 1) == 0) {
    flags |= AM355_MAP_CHILD;
  }
  ir.map[i] = am355_ir_map{o.id_ctr[g], o.id_actor[g], o.key_off[g], o.key_len[g], o.val_tl[g], (flags & AM355_MAP_CHILD) ? b.obj_index[g] : o.val_off[g],
                           flags, 0, counter};
  uint32_t oi = obj_index_of(b, b.obj_row[g]);
  uint32_t prev = e_prev != NONE32 ? obj_index_of(b, b.obj_row[b.em_row[e_prev]]) : NONE32;
  uint32_t next = e_next != NONE32 ? obj_index_of(b, b.obj_row[b.em_row[e_next]]) : NONE32;
  if (oi != prev) ir.obj[oi].map_begin = i;
  if (oi != next) ir.obj[oi].map_end = i + 1;
}

__global__ __launch_bounds__(BLOCK) void k_map_finish(MergeBufs b, const uint32_t* __restrict__ perm, uint32_t n, PatchIR ir) {
  uint32_t i = gtid();
  if (i >= n) return;
  map_finish_one(b, ir, i, perm[i], i > 0 ? perm[i - 1] : NONE32, i + 1 < n ? perm[i + 1] : NONE32);
}

// up to BLOCK emissions (a text document's root map: one): ranks by comparison and the records, one workgroup, one launch
__device__ __forceinline__ void map_small_finish(const MergeBufs& b, uint32_t n, const PatchIR& ir) {
  __shared__ uint32_t s_perm[BLOCK];
  __shared__ MapStaged s_m[BLOCK];
  uint32_t i = threadIdx.x;
  if (i < n) s_m[i] = map_stage(b, i, nullptr);
  __syncthreads();
  if (i < n) {
    const MapStaged mine = s_m[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) rank += map_staged_less(b, s_m[j], j, mine, i, nullptr) ? 1u : 0u;
    s_perm[rank] = i;
  }
  __syncthreads();
  if (i < n) map_finish_one(b, ir, i, s_perm[i], i > 0 ? s_perm[i - 1] : NONE32, i + 1 < n ? s_perm[i + 1] : NONE32);
}
__global__ __launch_bounds__(BLOCK) void k_map_small_finish(MergeBufs b, uint32_t n, PatchIR ir) {
  wave_priority_high();
  map_small_finish(b, n, ir);
}

static int bits_for(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b)) b++;
  return b;
}

// ---------------------------------------------------------------------------------------------------------
// RGA order of all list/text objects
// ---------------------------------------------------------------------------------------------------------
struct ListKeyBits {
  int b_row, b_ctr, b_actor;
};

__global__ __launch_bounds__(BLOCK) void k_list_keys(MergeBufs b, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t n, ListKeyBits kb) {
  uint32_t i = gtid();
  if (i >= n) return;
  uint32_t g = b.ins_row[i];
  uint32_t ref = b.ref_row[g];
  uint64_t is_head = ref == NONE32 ? 1 : 0;
  uint64_t parent = is_head ? b.obj_row[g] : ref;  // head children are grouped under their object's make row
  uint64_t mask_c = (1ull << kb.b_ctr) - 1, mask_a = (1ull << kb.b_actor) - 1;
  // siblings in DESCENDING op id order: complement the id bits
  uint64_t k = is_head << (kb.b_row + kb.b_ctr + kb.b_actor) | parent << (kb.b_ctr + kb.b_actor) | (~(uint64_t)b.ops.id_ctr[g] & mask_c) << kb.b_actor |
               (~(uint64_t)b.ops.id_actor[g] & mask_a);
  keys[i] = k;
  vals[i] = g;
}

// after the sort (radix path, taken when some element has more children than the in-place ordering handles): sibling links
// and first children
__global__ __launch_bounds__(BLOCK) void k_list_link(MergeBufs b, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                     ListKeyBits kb) {
  uint32_t j = gtid();
  if (j >= n) return;
  int sh = kb.b_ctr + kb.b_actor;
  uint64_t pk = keys[j] >> sh;
  uint32_t v = vals[j];
  bool is_head = (pk >> kb.b_row) != 0;
  bool first = j == 0 || (keys[j - 1] >> sh) != pk;
  bool last = j + 1 == n || (keys[j + 1] >> sh) != pk;
  if (first) {
    uint32_t parent = (uint32_t)(pk & ((1ull << kb.b_row) - 1));
    b.first_child[is_head ? b.n_ops + parent : parent] = v;
  }
  b.next_sib[v] = last ? NONE32 : vals[j + 1];
}

// ---- sibling ordering in place (the common case) -------------------------------------------------------------
// The RGA order of a list is the pre-order of its insertion forest with siblings in DESCENDING op id. Nearly every element
// has at most one child (typing), so no sort is run: every insert row pushes itself onto an unordered list of its parent
// (one atomic exchange), then walks that list once to find its next sibling (the greatest id below its own) and whether it
// is the first child. O(k) per node for a parent with k children; beyond SEG_SORT_MAX children the walk is abandoned
// (Counts.pad) and the host redoes the ordering with one radix sort of (parent | ~id) keys.
// Parent slot space: element parents are rows [0, N), "head of list object o" is N + make row of o.
constexpr uint32_t SEG_SORT_MAX = 256;

__device__ __forceinline__ uint32_t parent_slot(const MergeBufs& b, uint32_t g) {
  uint32_t ref = b.ref_row[g];
  return ref == NONE32 ? b.n_ops + b.obj_row[g] : ref;
}

// Children of a list HEAD are the one long sibling list of ordinary documents (everybody's first insert into an empty text,
// every insert at position 0): walking it costs every one of its k children k dependent loads. They are therefore ALSO
// collected in an array (up to HEAD_CHILD_MAX of them in the whole batch) which one workgroup orders from LDS
// (k_head_children); beyond that the linked lists serve as for any other parent.
constexpr uint32_t HEAD_CHILD_MAX = 2048;
constexpr uint32_t HEAD_CHILD_THREADS = 1024;

__global__ __launch_bounds__(BLOCK) void k_child_push(MergeBufs b) {
  wave_priority_high();
  uint32_t i = gtid();
  bool in_range = i < b.counts->n_list_ins;
  uint32_t v = in_range ? b.ins_row[i] : 0;
  bool head_child = in_range && b.ref_row[v] == NONE32;
  if (in_range) b.child_next[v] = atomicExch(&b.child_head[parent_slot(b, v)], v);
  uint32_t slot = wave_append(&b.counts->n_head_children, head_child);
  if (head_child && slot < HEAD_CHILD_MAX) b.head_child[slot] = v;
}

// one workgroup: sibling links of all head children (siblings in DESCENDING op id, per list object)
__global__ __launch_bounds__(HEAD_CHILD_THREADS) void k_head_children(MergeBufs b) {
  wave_priority_high();
  __shared__ unsigned long long s_id[HEAD_CHILD_MAX];
  __shared__ uint32_t s_obj[HEAD_CHILD_MAX], s_row[HEAD_CHILD_MAX];
  const uint32_t n = b.counts->n_head_children;
  if (n == 0 || n > HEAD_CHILD_MAX) return;
  for (uint32_t k = threadIdx.x; k < n; k += HEAD_CHILD_THREADS) {
    uint32_t v = b.head_child[k];
    s_row[k] = v;
    s_obj[k] = b.obj_row[v];
    s_id[k] = pack_id(b.ops.id_ctr[v], b.ops.id_actor[v]);
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n; k += HEAD_CHILD_THREADS) {
    const uint32_t obj = s_obj[k];
    const unsigned long long kv = s_id[k];
    unsigned long long best_id = 0;
    uint32_t greater = 0, best = NONE32;
    for (uint32_t j = 0; j < n; j++) {
      if (s_obj[j] != obj || j == k) continue;
      unsigned long long ku = s_id[j];
      if (ku > kv) greater++;
      else if (best == NONE32 || ku > best_id) { best = s_row[j]; best_id = ku; }
    }
    uint32_t v = s_row[k];
    b.next_sib[v] = best;
    if (greater == 0) b.first_child[b.n_ops + obj] = v;
  }
}

// Typing runs. Most elements of a text have exactly one child, inserted right after them (the next character typed):
// then enter(v) -> enter(child) and leave(child) -> leave(v) are forced links of the Euler tour, and a maximal run of such
// elements enters the tour as ONE pair of entries whose enter edge weighs the run's length. Runs are consecutive in the
// insert list (ins_row), so run k is [heads[k], heads[k+1]) there. The tour shrinks from 2 x elements to 2 x runs entries
// (x100 for typical typing) and the list ranking with it.
// FROM_LINKS: first_child / next_sib were produced by the radix path; only the run flags are computed here.
template <bool FROM_LINKS>
__global__ __launch_bounds__(BLOCK) void k_child_order(MergeBufs b, uint32_t* __restrict__ is_head) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  const uint32_t ni = b.counts->n_list_ins;
  if (blockIdx.x * BLOCK >= ni) return;  // (whole workgroup)
  uint32_t i = gtid();
  uint32_t head = 0;
  if (i < ni) {
    const OpCols& o = b.ops;
    uint32_t v = b.ins_row[i], ps = parent_slot(b, v);
    bool first, none_after;
    if (FROM_LINKS) {
      first = b.first_child[ps] == v;
      none_after = b.next_sib[v] == NONE32;
    } else if (ps >= b.n_ops && b.counts->n_head_children <= HEAD_CHILD_MAX) {
      first = none_after = false;  // a head child (linked by k_head_children) always starts a run: its parent is no element
    } else {
      unsigned long long kv = pack_id(o.id_ctr[v], o.id_actor[v]), best_id = 0;
      uint32_t greater = 0, best = NONE32, steps = 0;
      bool abandoned = false;
      for (uint32_t u = b.child_head[ps]; u != NONE32; u = b.child_next[u]) {
        if (++steps > SEG_SORT_MAX) { abandoned = true; break; }
        if (u == v) continue;
        unsigned long long ku = pack_id(o.id_ctr[u], o.id_actor[u]);
        if (ku > kv) greater++;
        else if (best == NONE32 || ku > best_id) { best = u; best_id = ku; }
      }
      if (abandoned) {  // consistent placeholder links; the host reruns the ordering through the radix path
        b.counts->pad = 1;
        b.next_sib[v] = NONE32;
        first = false;
        none_after = false;
      } else {
        b.next_sib[v] = best;
        if (greater == 0) b.first_child[ps] = v;
        first = greater == 0;
        none_after = best == NONE32;
      }
    }
    // v continues the run of its predecessor in the insert list iff it is that element's only child
    head = (i > 0 && ps == b.ins_row[i - 1] && first && none_after) ? 0u : 1u;
    is_head[i] = head;
  }
  carry_publish(b.cs_runs, head, s_red);  // (Counts.n_runs: written by k_run_heads)
}

// heads[k] = first insert-list index of run k (heads[H] = n); head_ex[i] = runs that start before i; row_run[] = run of a
// row, kept for run heads and tails only (the rows other runs refer to: a first child and a next sibling are always heads,
// a parent is always a tail)
__global__ __launch_bounds__(BLOCK) void k_run_heads(MergeBufs b, const uint32_t* __restrict__ is_head, uint32_t* __restrict__ head_ex,
                                                     uint32_t* __restrict__ heads, uint32_t* __restrict__ row_run) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  const uint32_t n = b.counts->n_list_ins;
  if (blockIdx.x * BLOCK >= n) return;
  uint32_t i = gtid();
  uint32_t flag = i < n ? is_head[i] : 0;
  uint32_t k = carry_prefix(b.cs_runs, flag, s_red);
  if (i >= n) return;
  head_ex[i] = k;
  if (flag) {
    heads[k] = i;
    row_run[b.ins_row[i]] = k;
    if (i > 0) row_run[b.ins_row[i - 1]] = k - 1;
  }
  if (i + 1 == n) {
    uint32_t H = k + flag;
    heads[H] = n;
    row_run[b.ins_row[i]] = H - 1;
    b.counts->n_runs = H;
    if (b.sig) signal_host(b.sig->runs, (const uint32_t*)b.counts, 16, &b.sig->runs_seq, b.sig_seq);  // (pad: k_child_order, completed)
  }
}

// Euler tour of the insertion forest over typing runs: enter(run k) = 2k, leave(run k) = 2k+1; END = 2 x runs. Every list
// object is its own tree; all of them are ranked in one pass (each tour ends at END) and the objects are laid out one after
// another afterwards (k_obj_n + prefix sum). One 64-bit word per entry (successor in the low half, weight-to-end in the
// high half): a pointer-jumping round is one 8-byte read, one dependent 8-byte read and one 8-byte write per entry.
__device__ __forceinline__ unsigned long long euler_pack(uint32_t succ, uint32_t dist) { return (unsigned long long)dist << 32 | succ; }

__global__ __launch_bounds__(BLOCK) void k_euler_init_runs(MergeBufs b, const uint32_t* __restrict__ heads, const uint32_t* __restrict__ row_run,
                                                           unsigned long long* __restrict__ el) {
  wave_priority_high();
  uint32_t k = gtid();
  uint32_t H = b.counts->n_runs, END = 2 * H;
  if (k == 0) el[END] = euler_pack(END, 0);
  if (k >= H) return;
  uint32_t i0 = heads[k], i1 = heads[k + 1];
  uint32_t h = b.ins_row[i0], t = b.ins_row[i1 - 1];
  uint32_t fc = b.first_child[t], ns = b.next_sib[h], ref = b.ref_row[h];
  el[2 * (size_t)k] = euler_pack(fc != NONE32 ? 2 * row_run[fc] : 2 * k + 1, i1 - i0);
  el[2 * (size_t)k + 1] = euler_pack(ns != NONE32 ? 2 * row_run[ns] : (ref != NONE32 ? 2 * row_run[ref] + 1 : END), 0);
}

// List ranking (Wyllie pointer jumping: dist'[x] = dist[x] + dist[succ[x]], succ'[x] = succ[succ[x]]) of a tour that fits
// the LDS of ONE compute unit -- 16 K entries = 8 K typing runs, which covers the headline 1 M-op text (8082 runs): one
// launch of one 1024-thread workgroup, log2(entries) rounds in place (every lane keeps its 16 entries in registers across
// the barrier), instead of log2(entries) launches over an array that small. Longer tours: k_euler_jump rounds in HBM.
//
// The rounds are bound by LDS bandwidth (per round and entry two random 8-byte reads and an 8-byte write: 36 us for 14 rounds over
// 16 k entries). A tour that is ONE path -- a document with one list object, the ordinary text document -- is ranked in half the
// bytes: the rounds jump over 32-bit entries (successor << 16 | hops to the end, both < 2^14), which gives every entry its POSITION in
// the tour; the run lengths are then scattered to their positions, one prefix sum over the positions gives the weight from every
// position to the end, and every entry picks up its own. Several list objects are several paths ending at the same END (equal hop
// counts on different paths): those tours keep the 64-bit rounds.
constexpr uint32_t EULER_LDS_ENTRIES = 16384;  // x 8 B = 128 KiB of the CU's 160 KiB
constexpr uint32_t EULER_LDS_THREADS = 1024;
__global__ __launch_bounds__(EULER_LDS_THREADS) void k_euler_rank_lds(Counts* __restrict__ counts, unsigned long long* __restrict__ el) {
  wave_priority_high();
  __shared__ unsigned long long L[EULER_LDS_ENTRIES];
  __shared__ uint32_t s_wave[EULER_LDS_THREADS / WAVE];
  const uint32_t H = counts->n_runs, END = 2 * H, E = END + 1;
  if (H == 0 || E > EULER_LDS_ENTRIES) return;
  const uint32_t t = threadIdx.x, lane = t & (WAVE - 1), wv = t / WAVE;
  constexpr uint32_t PER = EULER_LDS_ENTRIES / EULER_LDS_THREADS;
  int rounds = 1;
  while ((E >> rounds) != 0) rounds++;
  unsigned long long e[PER];
  uint32_t ends = 0;
#pragma unroll
  for (uint32_t j = 0; j < PER; j++) {
    const uint32_t x = t + j * EULER_LDS_THREADS;
    e[j] = x < E ? el[x] : 0ull;
    ends += (x < END && (uint32_t)e[j] == END) ? 1u : 0u;
  }
  ends = wave_sum_u32(ends);
  if (lane == 0) s_wave[wv] = ends;
  __syncthreads();
  uint32_t paths = 0;
  for (uint32_t k = 0; k < EULER_LDS_THREADS / WAVE; k++) paths += s_wave[k];
  __syncthreads();
  if (paths == 1) {
    uint32_t* P = (uint32_t*)L;                 // successor << 16 | hops to the end
    uint32_t* Wt = P + EULER_LDS_ENTRIES;       // run length by tour position (0 = the entry in front of END), then its prefix sums
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
      const uint32_t x = t + j * EULER_LDS_THREADS;
      if (x < E) P[x] = (uint32_t)e[j] << 16 | (x < END ? 1u : 0u);
      Wt[x] = 0;
    }
    __syncthreads();
      // (every round: the sixteen dependent reads of a thread are requested together -- END is a fixed point with zero hops, so they need
    // no branch; with a branch around each the compiler waited for every read in turn: 2.3 us a round. Now 1.6 us: 16 k random 4-byte LDS
    // reads at ~5 cycles per wavefront for their bank conflicts, which is what is left of this kernel: rounds 22 of its 32 us)
    uint32_t pv[PER];
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
      const uint32_t x = t + j * EULER_LDS_THREADS;
      pv[j] = x < END ? (uint32_t)e[j] << 16 | 1u : END << 16;  // (what P[x] holds; a thread keeps its own entries in registers between the rounds)
    }
    for (int r = 0; r < rounds; r++) {
      uint32_t qv[PER];
#pragma unroll
      for (uint32_t j = 0; j < PER; j++) qv[j] = P[pv[j] >> 16];
#pragma unroll
      for (uint32_t j = 0; j < PER; j++) pv[j] = (qv[j] & 0xffff0000u) | ((pv[j] + qv[j]) & 0xffffu);
      __syncthreads();
#pragma unroll
      for (uint32_t j = 0; j < PER; j++) {
        const uint32_t x = t + j * EULER_LDS_THREADS;
        if (x < END) P[x] = pv[j];
      }
      __syncthreads();
    }
    // run lengths to their tour positions
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
      const uint32_t x = t + j * EULER_LDS_THREADS;
      pv[j] = (pv[j] & 0xffffu) - 1;  // hops - 1 (an entry that never reaches END -- not in a well-formed tour -- lands out of range)
      if (x < END && pv[j] < END) Wt[pv[j]] = (uint32_t)(e[j] >> 32);
    }
    __syncthreads();
    // inclusive prefix sums over the positions: PER consecutive positions per thread, one scan over the threads' totals
    uint32_t loc[PER], run = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) { run += Wt[t * PER + k]; loc[k] = run; }
    const uint32_t incl = wave_incl_scan_u32(run, lane);
    if (lane == WAVE - 1) s_wave[wv] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    for (uint32_t k = 0; k < wv; k++) base += s_wave[k];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) Wt[t * PER + k] = base + loc[k];
    __syncthreads();
  #pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
      const uint32_t x = t + j * EULER_LDS_THREADS;
      if (x < END) el[x] = euler_pack(END, pv[j] < END ? Wt[pv[j]] : 0u);
    }
    if (t == 0) counts->euler_done = 1;
      return;
  }
#pragma unroll
  for (uint32_t j = 0; j < PER; j++) {
    const uint32_t x = t + j * EULER_LDS_THREADS;
    if (x < E) L[x] = e[j];
  }
  __syncthreads();
  for (int r = 0; r < rounds; r++) {
    unsigned long long nv[PER], fv[PER];
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
      const uint32_t x = t + j * EULER_LDS_THREADS;
      nv[j] = L[x < END ? x : END];
    }
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) fv[j] = L[(uint32_t)nv[j]];  // (L[END] = (END, 0): a fixed point, no branch)
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) nv[j] = euler_pack((uint32_t)fv[j], (uint32_t)(nv[j] >> 32) + (uint32_t)(fv[j] >> 32));
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
      const uint32_t x = t + j * EULER_LDS_THREADS;
      if (x < END) L[x] = nv[j];
    }
    __syncthreads();
  }
  for (uint32_t x = t; x < END; x += EULER_LDS_THREADS) el[x] = L[x];
  if (t == 0) counts->euler_done = 1;
}

// (Round 3 tried ranking by splitters instead -- one entry in eight walks to the next splitter, the splitters are ranked by pointer
// jumping, the stretches are walked again: a third of the LDS traffic, and SLOWER on the headline tour of 16 k entries, 40.7 us against
// 36.2 us (profiles/r03_ab_euler_rank.txt): the walks of a wavefront last as long as its longest one (62 steps where the average is 8)
// and sixteen wavefronts of such lane-divergent loops are bound by instruction issue, not by LDS bandwidth.)
// one pointer-jumping round over a tour in HBM (tours beyond the LDS kernel)
__global__ __launch_bounds__(BLOCK) void k_euler_jump(uint32_t n_runs, const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out) {
  uint32_t x = gtid();
  uint32_t END = 2 * n_runs;
  if (x > END) return;  // (x == END is the end marker itself)
  unsigned long long e = in[x];
  uint32_t s = (uint32_t)e;
  if (s != END) {
    unsigned long long t = in[s];
    e = euler_pack((uint32_t)t, (uint32_t)(e >> 32) + (uint32_t)(t >> 32));
  }
  out[x] = e;
}

// elements per list object = weight of the tour from its first head child to the end of its tree; entry n_obj is 0 (prefix sum)
__device__ __forceinline__ uint32_t obj_elems(const MergeBufs& b, const PatchIR& ir, uint32_t n_obj, const uint32_t* __restrict__ row_run,
                                              const unsigned long long* __restrict__ el, uint32_t oi) {
  uint32_t c = 0;
  if (oi > 0 && oi < n_obj) {
    uint32_t fc = b.first_child[b.n_ops + ir.obj[oi].make_row];
    if (fc != NONE32) c = (uint32_t)(el[2 * (size_t)row_run[fc]] >> 32);
  }
  return c;
}
__global__ __launch_bounds__(BLOCK) void k_obj_n(MergeBufs b, PatchIR ir, uint32_t n_obj, const uint32_t* __restrict__ row_run,
                                                 const unsigned long long* __restrict__ el) {
  uint32_t oi = gtid();
  if (oi > n_obj) return;
  b.obj_n[oi] = obj_elems(b, ir, n_obj, row_run, el, oi);
}

// element i of the compacted insert list -> its place in document order (n_o / first: element count and first position of its object)
__device__ __forceinline__ void list_order_one(const MergeBufs& b, uint32_t n, uint32_t i, uint32_t k, uint32_t head_i, uint32_t d, uint32_t n_o, uint32_t first) {
  uint32_t within = i - head_i;
  uint32_t pos = first + n_o - d + within;
  if (d == 0 || d > n_o || within >= d || pos >= n) { atomicOr(&b.counts->flags, (uint32_t)F_BAD_ELEM); return; }
  b.order[pos] = b.ins_row[i];
}
__global__ __launch_bounds__(BLOCK) void k_list_order(MergeBufs b, uint32_t n, const uint32_t* __restrict__ is_head, const uint32_t* __restrict__ head_ex,
                                                      const uint32_t* __restrict__ heads, const unsigned long long* __restrict__ el) {
  uint32_t i = gtid();
  if (i >= n) return;
  uint32_t k = head_ex[i] + is_head[i] - 1;  // i = 0 is always a head
  uint32_t d = (uint32_t)(el[2 * (size_t)k] >> 32);  // elements from the run's first one to the end of its object's tour, inclusive
  uint32_t oi = obj_index_of(b, b.obj_row[b.ins_row[i]]);
  list_order_one(b, n, i, k, heads[k], d, b.obj_n[oi], b.obj_first_pos[oi]);
}

// k_obj_n + the prefix sum over the objects + k_list_order in one launch, for documents of up to OBJ_LDS_MAX objects: every workgroup
// works the (few) object counts and their prefix sum out for itself in LDS -- three dependent loads per object -- instead of two more
// launches in front of it; workgroup 0 leaves obj_n / obj_first_pos in HBM for the kernels behind.
constexpr uint32_t OBJ_LDS_MAX = 1023;
// n_map_small != 0: one EXTRA workgroup (the last) orders and writes that many map emissions (map_small_finish): the root map of a text
// document is one emission, not worth a launch of its own in a chain of launches.
__global__ __launch_bounds__(BLOCK) void k_list_order_objs(MergeBufs b, PatchIR ir, uint32_t n_obj, uint32_t n, const uint32_t* __restrict__ is_head,
                                                           const uint32_t* __restrict__ head_ex, const uint32_t* __restrict__ heads,
                                                           const uint32_t* __restrict__ row_run, const unsigned long long* __restrict__ el, uint32_t n_map_small) {
  wave_priority_high();
  __shared__ uint32_t s_n[OBJ_LDS_MAX + 1], s_first[OBJ_LDS_MAX + 1], s_red[BLOCK / WAVE];
  if (n_map_small && blockIdx.x + 1 == gridDim.x) {
    map_small_finish(b, n_map_small, ir);
    return;
  }
  // (the element's own loads first: they are in flight while the object table is worked out)
  uint32_t i = gtid();
  uint32_t k = 0, d = 0, oi = 0, head_i = 0;
  if (i < n) {
    k = head_ex[i] + is_head[i] - 1;
    d = (uint32_t)(el[2 * (size_t)k] >> 32);
    head_i = heads[k];
    oi = obj_index_of(b, b.obj_row[b.ins_row[i]]);
  }
  for (uint32_t o = threadIdx.x; o <= n_obj; o += BLOCK) s_n[o] = obj_elems(b, ir, n_obj, row_run, el, o);
  __syncthreads();
  uint32_t carry = 0;
  for (uint32_t base = 0; base <= n_obj; base += BLOCK) {
    uint32_t o = base + threadIdx.x, total;
    uint32_t ex = block_exclusive_scan_u32(o <= n_obj ? s_n[o] : 0u, s_red, &total);
    if (o <= n_obj) s_first[o] = carry + ex;
    carry += total;
  }
  __syncthreads();
  if (blockIdx.x == 0)
    for (uint32_t o = threadIdx.x; o <= n_obj; o += BLOCK) { b.obj_n[o] = s_n[o]; b.obj_first_pos[o] = s_first[o]; }
  if (i < n) list_order_one(b, n, i, k, head_i, d, s_n[oi], s_first[oi]);
}

// per list position: visibility and edit counts, scanned by k_list_scan through the carried sums published here
// Lists that hold counters or visible rows without a value (Counts.n_quirk != 0; new.js:937-965, 1010-1018, 1026-1033). Per element the
// reference's state machine comes down to: the VALUES of the element in row order -- visible `set` / make rows, and every counter
// whose successors are all increments at the place of its last increment --; when a visible row without a value (an increment that
// does not complete its counter, a link) stands in front of the first value the element is first reported as `remove` and its
// values then all become `update` edits, and with no value at all the `remove` edit stays. vl_min[element row] = the smallest id
// (ctr << 32 | actor) among its visible rows without a value (all ones: none). An increment that completes its counter counts as
// such a row too: its id is the counter's place among the values, so it is never in FRONT of the first value.
__global__ __launch_bounds__(BLOCK) void k_quirk_rows(MergeBufs b, unsigned long long* __restrict__ vl_min) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  const uint8_t kind = b.kind[g];
  if (kind != K_LIST_INS && kind != K_LIST_UPD) return;  // (K_LIST_INS_VIS: valued)
  const uint32_t a = b.ops.action[g];
  if (a == 1 || (a & 1) == 0 || b.succ_cnt[g] != 0) return;
  const uint32_t el = kind == K_LIST_INS ? g : b.ref_row[g];
  if (el != NONE32) atomicMin(&vl_min[el], pack_id(b.ops.id_ctr[g], b.ops.id_actor[g]));
}

__global__ __launch_bounds__(BLOCK) void k_list_counts(MergeBufs b, uint32_t n, const unsigned long long* __restrict__ vl_min) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  uint32_t p = gtid();
  uint32_t c = 0;
  if (p < n) {
    uint32_t v = b.order[p];
    if (v == NONE32) atomicOr(&b.counts->flags, (uint32_t)F_BAD_ELEM);
    else {
      c = b.val_cnt[v] + (b.kind[v] == K_LIST_INS_VIS ? 1u : 0u);
      if (vl_min && c == 0 && vl_min[v] != ~0ull) c = 1;  // the `remove` edit of an element that shows rows but no value
    }
    b.list_vis[p] = c ? 1 : 0;
    b.list_cnt[p] = c;
  }
  carry_publish(b.cs_vis, c ? 1u : 0u, s_red);
  carry_publish(b.cs_cnt, c, s_red);
}
__global__ __launch_bounds__(BLOCK) void k_list_scan(MergeBufs b, uint32_t n, uint32_t* __restrict__ vis_ex, uint32_t* __restrict__ cnt_ex) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  uint32_t p = gtid();
  uint32_t vis = p < n ? b.list_vis[p] : 0, c = p < n ? b.list_cnt[p] : 0;
  uint32_t ve = carry_prefix(b.cs_vis, vis, s_red);
  uint32_t ce = carry_prefix(b.cs_cnt, c, s_red);
  if (p < n) { vis_ex[p] = ve; cnt_ex[p] = ce; }
  if (p + 1 == n) b.counts->n_edits = ce + c;
}

// The same two steps without the per-position arrays (list_vis / list_cnt written and read again, vis_ex / cnt_ex written and read again:
// 32 bytes per element of a text document): k_list_counts_pub only publishes the workgroup sums, k_list_scan_edits rebuilds a position's
// count, takes its two prefixes from the carried sums and writes the element's edits right away. An edit's index is then the number of
// visible elements in front of it over ALL lists (the lists are chained); the first position of every object leaves that number at
// its object's first element in vis_base[object], which k_edit_pack subtracts.
__device__ __forceinline__ uint32_t list_pos_count(const MergeBufs& b, uint32_t v, const unsigned long long* __restrict__ vl_min) {
  uint32_t c = b.val_cnt[v] + (b.kind[v] == K_LIST_INS_VIS ? 1u : 0u);
  if (vl_min && c == 0 && vl_min[v] != ~0ull) c = 1;  // the `remove` edit of an element that shows rows but no value
  return c;
}
__global__ __launch_bounds__(BLOCK) void k_list_counts_pub(MergeBufs b, uint32_t n, const unsigned long long* __restrict__ vl_min) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  uint32_t p = gtid();
  uint32_t c = 0;
  if (p < n) {
    uint32_t v = b.order[p];
    if (v == NONE32) atomicOr(&b.counts->flags, (uint32_t)F_BAD_ELEM);
    else c = list_pos_count(b, v, vl_min);
  }
  carry_publish(b.cs_vis, c ? 1u : 0u, s_red);
  carry_publish(b.cs_cnt, c, s_red);
}

__global__ __launch_bounds__(BLOCK) void k_upd_keys(MergeBufs b, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t n) {
  uint32_t i = gtid();
  if (i >= n) return;
  uint32_t g = b.upd_row[i];
  uint32_t el = b.ref_row[g], ctr = b.ops.id_ctr[g], actor = b.ops.id_actor[g];
  if (b.succ_cnt[g] != 0) {  // a counter completed by increments (k_emit): placed by its last increment; may be the element's insert row
    const unsigned long long t = b.last_inc[g];
    ctr = (uint32_t)(t >> 32);
    actor = (uint32_t)t;
    if (b.kind[g] == K_LIST_INS) el = g;
  }
  keys[i] = (uint64_t)el << (b.bits_ctr + b.bits_actor) | (uint64_t)ctr << b.bits_actor | actor;
  vals[i] = g;
}

// one element's edits (insert first, then updates in ascending op id) from entry e on; index: the element's list index
__device__ __forceinline__ void list_edits_of(const MergeBufs& b, uint32_t v, uint32_t index, uint32_t e, const uint64_t* __restrict__ upd_keys,
                                              const uint32_t* __restrict__ upd_vals, uint32_t n_upd, const PatchIR& ir, const unsigned long long* __restrict__ vl_min) {
  const bool own = b.kind[v] == K_LIST_INS_VIS;  // the insert's own value is visible
  uint32_t c = b.val_cnt[v] + (own ? 1u : 0u);
  const unsigned long long vl = vl_min ? vl_min[v] : ~0ull;  // first visible row without a value (k_quirk_rows)
  if (!c && vl == ~0ull) return;
  uint32_t k = 0;
  if (!c) {  // rows, but no value: the `remove` edit stays (new.js:1026-1033)
    ir.e_row[e] = v; ir.e_elem[e] = v; ir.e_index[e] = index; ir.e_flags[e] = 8u;
    return;
  }
  uint32_t a = b.ops.action[v];
  // (the insert row has the smallest id of its element: a row without a value is never in front of it)
  if (own) {
    ir.e_row[e] = v; ir.e_elem[e] = v; ir.e_index[e] = index; ir.e_flags[e] = ((a & 1) == 0 ? 4u : 0u);
    k = 1;
  }
  if (k < c) {
    int sh = b.bits_ctr + b.bits_actor;
    uint32_t lo = 0, hi = n_upd;
    while (lo < hi) {  // first update record of element v
      uint32_t mid = (lo + hi) >> 1;
      if ((upd_keys[mid] >> sh) < v) lo = mid + 1; else hi = mid;
    }
    // a row without a value in front of the first value: the element was reported as `remove` first, every value is an `update` then
    // (new.js:1010-1018; appendUpdate finds nothing of this element to take away: its index is its own)
    bool removed_first = false;
    if (!own && vl != ~0ull && lo < n_upd && (upd_keys[lo] >> sh) == v) {
      const uint64_t first_id = upd_keys[lo] & ((1ull << sh) - 1);
      removed_first = ((uint64_t)(vl >> 32) << b.bits_actor | (uint32_t)vl) < first_id;
    }
    bool shown = own || vl != ~0ull;  // some row of the element has no successor (the reference's elemVisible, new.js:1626)
    for (; k < c && lo < n_upd && (upd_keys[lo] >> sh) == v; k++, lo++) {
      uint32_t u = upd_vals[lo];
      const bool counter = b.succ_cnt[u] != 0;  // (k_emit: the only rows with successors that are listed)
      shown = shown || !counter;
      ir.e_row[e + k] = u; ir.e_elem[e + k] = v; ir.e_index[e + k] = index;
      ir.e_flags[e + k] = ((k || removed_first) ? 1u : 0u) | ((b.ops.action[u] & 1) == 0 ? 4u : 0u) | (counter ? 32u : 0u);
    }
    if (k != c) atomicOr(&b.counts->flags, (uint32_t)F_BAD_ELEM);
    // a counter whose increments have all been deleted: the reference lists its total but does not count the element as visible -- the
    // next element is then reported at the same index and the two interfere (appendUpdate, new.js:803-815): left to the JS path
    if (!shown) atomicOr(&b.counts->flags, (uint32_t)F_UNSUPPORTED);
  }
}

// one lane per list position
__global__ __launch_bounds__(BLOCK) void k_list_edits(MergeBufs b, uint32_t n, const uint32_t* __restrict__ vis_ex, const uint32_t* __restrict__ cnt_ex,
                                                      const uint64_t* __restrict__ upd_keys, const uint32_t* __restrict__ upd_vals, uint32_t n_upd,
                                                      PatchIR ir, const unsigned long long* __restrict__ vl_min) {
  wave_priority_high();
  uint32_t p = gtid();
  if (p >= n) return;
  uint32_t v = b.order[p];
  if (v == NONE32) return;
  uint32_t oi = obj_index_of(b, b.obj_row[v]);
  list_edits_of(b, v, vis_ex[p] - vis_ex[b.obj_first_pos[oi]], cnt_ex[p], upd_keys, upd_vals, n_upd, ir, vl_min);
}

// k_list_scan + k_list_edits in one pass over the positions (see k_list_counts_pub)
__global__ __launch_bounds__(BLOCK) void k_list_scan_edits(MergeBufs b, uint32_t n, const uint64_t* __restrict__ upd_keys, const uint32_t* __restrict__ upd_vals,
                                                           uint32_t n_upd, PatchIR ir, const unsigned long long* __restrict__ vl_min, uint32_t* __restrict__ vis_base) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  uint32_t p = gtid();
  uint32_t v = p < n ? b.order[p] : NONE32;
  uint32_t c = v != NONE32 ? list_pos_count(b, v, vl_min) : 0u;
  uint32_t ve = carry_prefix(b.cs_vis, c ? 1u : 0u, s_red);
  uint32_t ce = carry_prefix(b.cs_cnt, c, s_red);
  if (p + 1 == n) b.counts->n_edits = ce + c;
  if (v == NONE32) return;
  uint32_t oi = obj_index_of(b, b.obj_row[v]);
  if (p == b.obj_first_pos[oi]) vis_base[oi] = ve;
  list_edits_of(b, v, ve, ce, upd_keys, upd_vals, n_upd, ir, vl_min);
}

// multi-insert run detection (new.js:754-773), first / last edit of every list object; publishes the number of edit RECORDS
// (an edit that does not continue a multi-insert run starts one) for k_edit_pack's carried scan
__global__ __launch_bounds__(BLOCK) void k_edit_runs(MergeBufs b, PatchIR ir) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  uint32_t n = b.counts->n_edits;
  if (blockIdx.x * BLOCK >= n && blockIdx.x) return;  // (whole workgroup; workgroup 0 always publishes)
  uint32_t e = gtid();
  uint32_t head = 0;
  if (e < n) {
    const OpCols& o = b.ops;
    uint32_t r = ir.e_row[e], el = ir.e_elem[e], f = ir.e_flags[e] & 0x2du;  // update | child | remove | counter total
    uint32_t oi = obj_index_of(b, b.obj_row[el]);
    uint32_t prev_oi = NONE32, next_oi = NONE32;
    if (e > 0) {
      uint32_t pr = ir.e_row[e - 1], pel = ir.e_elem[e - 1], pf = ir.e_flags[e - 1];
      prev_oi = obj_index_of(b, b.obj_row[pel]);
      bool simple = !(f & 0xdu) && r == el, psimple = !(pf & 0xdu) && pr == pel;
      if (simple && psimple && prev_oi == oi && o.id_actor[r] == o.id_actor[pr] && o.id_ctr[r] == o.id_ctr[pr] + 1 &&
          value_class(o.val_tl[r]) == value_class(o.val_tl[pr]) && ir.e_index[e] == ir.e_index[e - 1] + 1) {
        f |= 2u;
        // a record holds values with one type/length word, back to back in the arena (consecutive ops of a change: consecutive
        // bytes of its valRaw column); anything else -- a counter's total is not in the arena at all -- starts a new record of
        // the same multi-insert
        uint32_t tl = o.val_tl[r], ptl = o.val_tl[pr];
        if (tl != ptl || o.val_off[r] != o.val_off[pr] + (ptl >> 4) || ((f | pf) & 32u)) f |= 0x400u;
      }
    }
    if (e + 1 < n) next_oi = obj_index_of(b, b.obj_row[ir.e_elem[e + 1]]);
    if (oi != prev_oi) f |= 0x100u;
    if (oi != next_oi) f |= 0x200u;
    ir.e_flags[e] = f;
    head = (!(f & 2u) || (f & 0x400u)) ? 1u : 0u;
  }
  carry_publish(b.cs_erec, head, s_red);
}

// consumer of k_edit_runs' carried scan (same grid): one edit record per head (start of an edit, or of a new uniform stretch of a
// multi-insert), the edit ranges of the list objects, the sentinel record and Counts.n_erecs
// (vis_base: the replay's e_index counts the visible elements of ALL lists in front -- k_list_scan_edits --, less the object's own base;
// null: e_index is the list index already)
__global__ __launch_bounds__(BLOCK) void k_edit_pack(MergeBufs b, PatchIR ir, const uint32_t* __restrict__ vis_base) {
  wave_priority_high();
  __shared__ uint32_t s_red[BLOCK / WAVE];
  uint32_t n = b.counts->n_edits;
  if (blockIdx.x * BLOCK >= n && blockIdx.x) return;
  uint32_t e = gtid();
  uint32_t f = e < n ? ir.e_flags[e] : 2u;
  uint32_t head = (!(f & 2u) || (f & 0x400u)) ? 1u : 0u;
  uint32_t k = carry_prefix(b.cs_erec, head, s_red);
  if (e == 0 && n == 0) {
    ir.edit[0] = am355_ir_edit{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    b.counts->n_erecs = 0;
    if (b.sig) signal_host(b.sig->final_counts, (const uint32_t*)b.counts, 16, &b.sig->final_seq, b.sig_seq);
  }
  if (e >= n) return;
  const OpCols& o = b.ops;
  uint32_t r = ir.e_row[e], el = ir.e_elem[e];
  if (head) {
    uint32_t val = (f & 4u) ? b.obj_index[r] : o.val_off[r], val_hi = 0;
    if (f & 32u) {  // the total of a counter (new.js:944, 958): its `set` value + the increments k_resolve summed up
      long long base = 0;
      if (!int_value(b, r, base)) atomicOr(&b.counts->flags, (uint32_t)F_BAD_LEB);
      const unsigned long long total = (unsigned long long)(base + (long long)b.inc_sum[r]);
      val = (uint32_t)total;
      val_hi = (uint32_t)(total >> 32);
    }
    const uint32_t index = ir.e_index[e] - (vis_base ? vis_base[obj_index_of(b, b.obj_row[el])] : 0u);
    ir.edit[k] = am355_ir_edit{f & 0x2fu, index, o.id_ctr[r], o.id_actor[r], o.id_ctr[el], o.id_actor[el], e, o.val_tl[r], val, val_hi};
  }
  if (f & 0x300u) {
    uint32_t oi = obj_index_of(b, b.obj_row[el]);
    if (f & 0x100u) ir.obj[oi].edit_begin = k;         // (the first edit of an object is always a head)
    if (f & 0x200u) ir.obj[oi].edit_end = k + head;
  }
  if (e + 1 == n) {
    ir.edit[k + head] = am355_ir_edit{0, 0, 0, 0, 0, 0, n, 0, 0, 0};
    b.counts->n_erecs = k + head;
    // the counters are final (this kernel raises no flags); other workgroups may still be writing their records, which the host
    // only reads through later operations on this stream
    if (b.sig) signal_host(b.sig->final_counts, (const uint32_t*)b.counts, 16, &b.sig->final_seq, b.sig_seq);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Document load (Backend.load): whole-document patch of rows that are ALREADY in canonical order with their succ
// lists (a saved document: columnar.js:892, new.js:2047).  No sorting or list ranking is needed -- object groups,
// map keys and list elements are contiguous in row order -- so the patch is element-wise work plus prefix sums.
// Reference: new.js:1604-1635 documentPatch, 884-1040 updatePatchProperty.  Rows come from k_decode_columns in
// document mode (pred_* arrays hold the succ lists).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t doc_hash_slot(unsigned long long key, uint32_t mask) { return (uint32_t)((key * 0x9e3779b97f4a7c15ull) >> 40) & mask; }

// per row: kind, succ count, list-insert flag, make flag; make rows enter the id -> row table
__global__ __launch_bounds__(BLOCK) void k_doc_prepare(MergeBufs b, unsigned long long* __restrict__ tab_key, uint32_t* __restrict__ tab_row, uint32_t mask) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  const OpCols& o = b.ops;
  uint32_t a = o.action[g], err = 0;
  bool has_str = o.key_len[g] != NONE32, has_elem = o.key_ctr[g] != NONE32, ins = o.insert[g] != 0;
  uint8_t kind = K_NONE;
  if (has_str == has_elem) err |= has_str ? F_UNSUPPORTED : F_BAD_ROW;
  else if (has_str) { kind = K_MAP; if (ins) err |= F_UNSUPPORTED; }
  else kind = ins ? K_LIST_INS : K_LIST_UPD;
  if (a == 3) err |= F_UNSUPPORTED;  // a `del` row only exists in documents written from the empty-pred corner case
  b.kind[g] = kind;
  b.succ_cnt[g] = o.pred_num[g];
  b.scan_b[g] = kind == K_LIST_INS ? 1u : 0u;
  bool is_make = kind != K_NONE && (a & 1) == 0;
  b.obj_index[g] = is_make ? 1u : 0u;
  if (is_make) {
    unsigned long long key = pack_id(o.id_ctr[g], o.id_actor[g]);
    uint32_t i = doc_hash_slot(key, mask);
    for (uint32_t probes = 0; probes <= mask; probes++) {
      unsigned long long old = atomicCAS(&tab_key[i], 0ull, key);
      if (old == 0) { tab_row[i] = g; break; }
      if (old == key) { err |= F_DUP_OPID; break; }
      i = (i + 1) & mask;
    }
  }
  if (err) atomicOr(&b.counts->flags, err);
}

__device__ __forceinline__ uint32_t doc_find_make(const unsigned long long* __restrict__ tab_key, const uint32_t* __restrict__ tab_row, uint32_t mask,
                                                  unsigned long long key) {
  uint32_t i = doc_hash_slot(key, mask);
  for (uint32_t probes = 0; probes <= mask; probes++) {
    unsigned long long k = tab_key[i];
    if (k == 0) return NONE32;
    if (k == key) return tab_row[i];
    i = (i + 1) & mask;
  }
  return NONE32;
}

// The counter an increment ON A LIST ELEMENT feeds (document rows: the rows of an element follow its insert row `el`, ascending by
// id): the latest preceding `set` of a counter that lists the increment among its successors (counterStates[succOp] = counterState,
// later assignment wins: new.js:944-950). NONE32: none -- "increment operation for unknown counter".
// (the walk is bounded: a property with more than DOC_COUNTER_WALK rows between an increment and its counter -- a counter incremented
// a hundred thousand times -- would cost its rows a quadratic number of steps inside one launch; such documents get F_UNSUPPORTED from
// k_doc_resolve and go to the JS path)
constexpr uint32_t DOC_COUNTER_WALK = 1u << 16;
__device__ __forceinline__ uint32_t doc_list_counter_of(const MergeBufs& b, uint32_t g, uint32_t el) {
  const OpCols& o = b.ops;
  if (el == NONE32 || el >= g) return NONE32;
  const uint32_t my_a = o.id_actor[g], my_c = o.id_ctr[g];
  const uint32_t stop = g - el > DOC_COUNTER_WALK ? g - DOC_COUNTER_WALK : el;
  for (uint32_t r = g; r-- > stop;) {
    if (o.action[r] == 1 && (o.val_tl[r] & 15) == 8) {
      const uint32_t f = o.pred_first[r], n = o.pred_num[r];
      for (uint32_t k = 0; k < n; k++)
        if (o.pred_actor[f + k] == my_a && o.pred_ctr[f + k] == my_c) return r;
    }
  }
  return NONE32;
}

// per row: object (make row) lookup, element of a list update, counter bookkeeping, first row of every object group
__global__ __launch_bounds__(BLOCK) void k_doc_resolve(MergeBufs b, const unsigned long long* __restrict__ tab_key, const uint32_t* __restrict__ tab_row,
                                                       uint32_t mask, const uint32_t* __restrict__ ins_ex, uint32_t* __restrict__ obj_first) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  const OpCols& o = b.ops;
  uint32_t err = 0;
  uint8_t kind = b.kind[g];
  uint32_t orow = NONE32;
  bool list_obj = false;
  if (o.obj_actor[g] != NONE32) {
    orow = doc_find_make(tab_key, tab_row, mask, pack_id(o.obj_ctr[g], o.obj_actor[g]));
    if (orow == NONE32) err |= F_UNKNOWN_OBJECT;
    else { uint32_t oa = o.action[orow]; list_obj = oa == 2 || oa == 4; }
  }
  if (kind == K_MAP && list_obj) err |= F_UNSUPPORTED;
  if ((kind == K_LIST_INS || kind == K_LIST_UPD) && !list_obj) err |= F_UNSUPPORTED;
  b.obj_row[g] = orow;
  bool new_obj = g == 0 || o.obj_actor[g - 1] != o.obj_actor[g] || o.obj_ctr[g - 1] != o.obj_ctr[g];
  if (new_obj && !(err & F_UNKNOWN_OBJECT)) {
    uint32_t oi = obj_index_of(b, orow);
    if (atomicCAS(&obj_first[oi], NONE32, g) != NONE32) err |= F_UNSUPPORTED;  // the rows of one object must be contiguous
  }
  // canonical row order (the order save() writes: columnar.js:892, new.js:2047): objects ascending by id with _root
  // first, map keys ascending in UTF-16 order, the ops of one key / element ascending by id. Anything else is legal
  // input the reference treats row by row; it is left to the JS path.
  if (g > 0) {
    unsigned long long prev_id = pack_id(o.id_ctr[g - 1], o.id_actor[g - 1]), my_id = pack_id(o.id_ctr[g], o.id_actor[g]);
    if (new_obj) {
      bool prev_root = o.obj_actor[g - 1] == NONE32, cur_root = o.obj_actor[g] == NONE32;
      if (cur_root || (!prev_root && pack_id(o.obj_ctr[g - 1], o.obj_actor[g - 1]) >= pack_id(o.obj_ctr[g], o.obj_actor[g]))) err |= F_UNSUPPORTED;
    } else if (kind == K_MAP && b.kind[g - 1] == K_MAP) {
      if (o.key_off[g - 1] == o.key_off[g] || same_key(b, g - 1, g)) { if (prev_id >= my_id) err |= F_UNSUPPORTED; }
      else {
        const uint8_t *p = b.arena + o.key_off[g - 1], *q = b.arena + o.key_off[g];
        uint32_t lp = o.key_len[g - 1], lq = o.key_len[g], k = 0;
        while (k < lp && k < lq && p[k] == q[k]) k++;
        bool less = k == lp ? k < lq : (k < lq && utf16_order_byte(p[k]) < utf16_order_byte(q[k]));
        if (!less) err |= F_UNSUPPORTED;
      }
    } else if (kind == K_LIST_UPD && prev_id >= my_id) err |= F_UNSUPPORTED;
  }
  uint32_t ref = NONE32;
  if (kind == K_LIST_UPD) {
    uint32_t before = ins_ex[g];  // list-insert rows before g: the update belongs to the latest one
    ref = before ? b.ins_row[before - 1] : NONE32;
    if (ref == NONE32 || o.id_actor[ref] != o.key_actor[g] || o.id_ctr[ref] != o.key_ctr[g] || !same_obj(b, ref, g)) { err |= F_BAD_ELEM; ref = NONE32; }
  }
  b.ref_row[g] = ref;
  if (kind == K_MAP && o.action[g] == 5) {
    // inc: the counter it feeds is the latest preceding `set` of a counter, on the same key, that lists this op as a
    // successor (counterStates[succOp] = counterState, later assignment wins: new.js:944-950)
    uint32_t my_a = o.id_actor[g], my_c = o.id_ctr[g];
    uint32_t owner = NONE32;
    for (uint32_t r = g; r-- > 0 && owner == NONE32;) {
      if (g - r > DOC_COUNTER_WALK) { err |= F_UNSUPPORTED; break; }  // (bounded walk, see doc_list_counter_of)
      if (!same_obj(b, r, g) || !same_key(b, r, g)) break;
      if (o.action[r] == 1 && (o.val_tl[r] & 15) == 8) {
        uint32_t f = o.pred_first[r], n = o.pred_num[r];
        for (uint32_t k = 0; k < n; k++)
          if (o.pred_actor[f + k] == my_a && o.pred_ctr[f + k] == my_c) owner = r;
      }
    }
    long long v;
    if (owner == NONE32) { if (!(err & F_UNSUPPORTED)) err |= F_BAD_COUNTER; }  // increment operation for unknown counter (new.js:954-956); (not when the walk was cut)
    else if (!int_value(b, g, v)) err |= F_UNSUPPORTED;
    else {
      atomicAdd(&b.inc_cnt[owner], 1u);
      atomicAdd(&b.inc_sum[owner], (unsigned long long)v);
      atomicMax(&b.last_inc[owner], (unsigned long long)g);  // row order within a key is op-id order
    }
  } else if (kind != K_NONE && kind != K_MAP && o.action[g] == 5 && !(err & F_BAD_ELEM)) {
    // an increment on a list element (new.js:952-965)
    const uint32_t owner = doc_list_counter_of(b, g, ref);
    long long v;
    if (owner == NONE32 && ref != NONE32 && g - ref > DOC_COUNTER_WALK) err |= F_UNSUPPORTED;  // (the walk was cut short)
    else if (owner == NONE32) err |= F_BAD_COUNTER;
    else if (!int_value(b, g, v)) err |= F_UNSUPPORTED;
    else {
      atomicAdd(&b.inc_cnt[owner], 1u);
      atomicAdd(&b.inc_sum[owner], (unsigned long long)v);
      atomicMax(&b.last_inc[owner], (unsigned long long)g);
    }
  }
  if (err) atomicOr(&b.counts->flags, err);
}

// per row: which rows trigger a map emission / produce a list edit; value counts per list element
// (lists: a value-less visible row -- an increment that does not complete its counter, a link -- and a counter completed by increments
// follow the reference's `remove` rule, see k_quirk_rows. vl_first[insert row] = the first visible row without a value of the element;
// the row of the LAST increment of a completed counter is the place of the counter's value among the element's edits, and carries it.)
__global__ __launch_bounds__(BLOCK) void k_doc_emit(MergeBufs b, uint32_t* __restrict__ trig_flag, uint32_t* __restrict__ trig_src, uint32_t* __restrict__ edit_flag,
                                                    uint32_t* __restrict__ vl_first) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  const OpCols& o = b.ops;
  uint8_t kind = b.kind[g];
  uint32_t a = o.action[g];
  bool vis = b.succ_cnt[g] == 0;
  bool valued = a == 1 || (a & 1) == 0;
  edit_flag[g] = 0;
  // objectId sharding of a document (am355_set_shard before am355_load_document): every rank decodes and checks all
  // rows -- a document's columns are run-length streams that cannot be entered in the middle -- but emits the records of the objects
  // it owns only; the object table (make rows) is the same on every rank, so the fragments stitch by object index
  if (b.shard_world > 1 && shard_owner(o.obj_actor[g], o.obj_ctr[g], b.shard_world) != b.shard_rank) return;
  if (kind == K_MAP) {
    if (vis && valued) { trig_flag[g] = 1; trig_src[g] = g; }
    else if (a == 1 && (o.val_tl[g] & 15) == 8 && b.succ_cnt[g] != 0 && b.inc_cnt[g] == b.succ_cnt[g]) {
      uint32_t t = (uint32_t)b.last_inc[g];  // the row of the last increment completes the counter
      trig_flag[t] = 1;
      trig_src[t] = g;
    }
  } else if (kind == K_LIST_INS || kind == K_LIST_UPD) {
    const uint32_t el = kind == K_LIST_INS ? g : b.ref_row[g];
    if (el == NONE32) return;
    if (vis && valued) { edit_flag[g] = 1; atomicAdd(&b.val_cnt[el], 1u); }
    if (vis && !valued) atomicMin(&vl_first[el], g);
    if (a == 5) {
      const uint32_t owner = doc_list_counter_of(b, g, el);
      if (owner != NONE32 && b.inc_cnt[owner] == b.succ_cnt[owner] && (uint32_t)b.last_inc[owner] == g) {
        edit_flag[g] = 1;
        atomicAdd(&b.val_cnt[el], 1u);
        // a counter whose increments have all been deleted is listed by the reference without the element counting as visible; the
        // next element then has the same index and the two interfere (appendUpdate, new.js:803-815): left to the JS path
        bool shown = false;
        for (uint32_t r = el; r < b.n_ops && !shown && (r == el || (b.kind[r] == K_LIST_UPD && b.ref_row[r] == el)); r++) shown = b.succ_cnt[r] == 0;
        if (!shown) atomicOr(&b.counts->flags, (uint32_t)F_UNSUPPORTED);
      }
    }
  }
}

// per insert row: does the element show in the list (a value, or -- the reference's `remove` rule -- any visible row); an element that
// shows rows but no value gets its `remove` edit at the first of them
__global__ __launch_bounds__(BLOCK) void k_doc_visflag(MergeBufs b, uint32_t* __restrict__ flag, const uint32_t* __restrict__ vl_first, uint32_t* __restrict__ edit_flag) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  bool vis = false;
  if (b.kind[g] == K_LIST_INS) {
    if (b.val_cnt[g] > 0) vis = true;
    else if (vl_first[g] != NONE32) { vis = true; edit_flag[vl_first[g]] = 1; }
  }
  flag[g] = vis ? 1u : 0u;
}

__global__ __launch_bounds__(BLOCK) void k_doc_scatter(MergeBufs b, const uint32_t* __restrict__ trig_flag, const uint32_t* __restrict__ trig_src,
                                                       const uint32_t* __restrict__ em_pos, const uint32_t* __restrict__ edit_flag,
                                                       const uint32_t* __restrict__ ed_pos, const uint32_t* __restrict__ idx_ex,
                                                       const uint32_t* __restrict__ obj_first, uint32_t* __restrict__ perm, PatchIR ir,
                                                       const uint32_t* __restrict__ vl_first) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  if (trig_flag[g]) {
    uint32_t i = em_pos[g];
    b.em_row[i] = trig_src[g];
    perm[i] = i;
  }
  if (edit_flag[g]) {
    uint32_t e = ed_pos[g];
    uint32_t el = b.kind[g] == K_LIST_INS ? g : b.ref_row[g];
    uint32_t first_row = obj_first[obj_index_of(b, b.obj_row[g])];
    const uint32_t a = b.ops.action[g];
    uint32_t src = g, f;
    if (b.val_cnt[el] == 0) f = 8u;  // rows, but no value: the `remove` edit (k_doc_visflag)
    else {
      // an increment stands for the counter it completes (k_doc_emit); a visible row without a value in front of the element's
      // first value makes every value an `update` (new.js:1010-1018)
      if (a == 5) src = doc_list_counter_of(b, g, el);
      f = ((ed_pos[g] != ed_pos[el] || vl_first[el] < g) ? 1u : 0u) | ((a & 1) == 0 ? 4u : 0u) | (a == 5 ? 32u : 0u);
    }
    ir.e_row[e] = src;
    ir.e_elem[e] = el;
    ir.e_index[e] = idx_ex[el] - idx_ex[first_row];
    ir.e_flags[e] = f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t n) { return dim3((n + BLOCK - 1) / BLOCK); }

// Host side of signal_host (am355_device.h): spin on the sequence word in pinned memory. If the word does not arrive within a
// generous bound (a device fault, a kernel that was never launched) the stream is drained instead: the caller then reads whatever
// the device left, and the error surfaces through the HIP status of the next call.
bool wait_host_signal(volatile uint32_t* seq_word, uint32_t seq, hipStream_t st) {
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spins = 0; *seq_word != seq; spins++) {
    if ((spins & 0x3ff) == 0x3ff) {
      auto waited = std::chrono::steady_clock::now() - t0;
      // Long past the time the phase takes: nudge the runtime (commands it still holds back are submitted by a query; seen under
      // rocprofv3 with copies enqueued and no host wait before the kernels); a stream that ran dry, or five seconds, end the wait.
      if (waited > std::chrono::seconds(5) || (waited > std::chrono::microseconds(400) && hipStreamQuery(st) == hipSuccess)) {
        (void)hipStreamSynchronize(st);
        break;
      }
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return *seq_word == seq;  // false: the words behind the signal are NOT this replay's -- the caller reads the device's copy
}

// the counters of a phase: from the pinned words the phase's last kernel signalled, or -- no signal -- from the device after a drain
// (key_stats: the MapKeyStats behind the counters, when the caller wants them -- the signal after k_compact_rows carries them)
static void read_phase_counts(MergeBufs& b, volatile uint32_t* seq_word, const uint32_t* words, Counts* dst, hipStream_t st, MapKeyStats* key_stats = nullptr) {
  if (wait_host_signal(seq_word, b.sig_seq, st)) {
    memcpy(dst, (const void*)words, sizeof(Counts));
    if (key_stats) memcpy(key_stats, (const void*)(words + MAP_KEY_STATS_WORD), sizeof(MapKeyStats));
    return;
  }
  if (key_stats) {  // (no signal: every byte position is taken to vary)
    key_stats->or_len = key_stats->or_inv_len = ~0u;
    for (int k = 0; k < 4; k++) key_stats->or_b[k] = key_stats->or_inv_b[k] = ~0u;
  }
  (void)hipMemcpyAsync(dst, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st);
  (void)hipStreamSynchronize(st);
}

size_t merge_counts_bytes(uint32_t n_ops) {
  size_t groups = (((((size_t)n_ops + BLOCK - 1) / BLOCK + 1) >> CARRY_GROUP_SHIFT) + 2) * CARRY_GROUP_STRIDE;
  return ((((sizeof(Counts) + 255) & ~(size_t)255) + 6 * 4 * groups) + 255) & ~(size_t)255;
}

void merge_bind_counts(MergeBufs& b, void* block) {
  size_t groups = (((((size_t)b.n_ops + BLOCK - 1) / BLOCK + 1) >> CARRY_GROUP_SHIFT) + 2) * CARRY_GROUP_STRIDE;
  b.counts = (Counts*)block;
  b.counts_bytes = merge_counts_bytes(b.n_ops);
  uint32_t* g = (uint32_t*)((uint8_t*)block + ((sizeof(Counts) + 255) & ~(size_t)255));
  b.cs_ins.group_sum = g;
  b.cs_make.group_sum = g + groups;
  b.cs_runs.group_sum = g + 2 * groups;   // runs | vis | cnt are contiguous: cleared together when the list ordering is redone
  b.cs_vis.group_sum = g + 3 * groups;
  b.cs_cnt.group_sum = g + 4 * groups;
  b.cs_erec.group_sum = g + 5 * groups;
}

void merge_prepare(MergeBufs& b, hipStream_t aux, int what, const FillRanges* extra) {
  uint32_t N = b.n_ops;
  if (b.row_stride == 0 && b.first_row == 0) {
    (void)hipMemsetAsync(b.zero_base, 0, b.zero_bytes, aux);  // succ_cnt, inc_cnt, val_cnt, inc_sum, last_inc
    if (extra) launch_fill_ranges(*extra, aux);
    if (!N) return;
    (void)hipMemsetAsync(b.fill_base, 0xff, b.fill_bytes, aux);  // order, first_child, child_head
    return;
  }
  // arrays carved for a capacity (am355_apply_changes: the rows stay where they are from call to call) are filled by their rows, in
  // one launch; the accumulators of kept rows [0, first_row) keep what the earlier rows left in them
  const size_t F = b.first_row, M = (size_t)N + 1 - F;
  FillRanges f;
  if (what & MERGE_FILL_ROWS) {
    f.add(b.succ_cnt + F, 4 * M, 0);
    f.add(b.inc_cnt + F, 4 * M, 0);
    f.add(b.inc_sum + F, 8 * M, 0);
    f.add(b.last_inc + F, 8 * M, 0);
  }
  if (what & MERGE_FILL_TABLES) {
    f.add(b.val_cnt, 4 * ((size_t)N + 1), 0);   // (recounted by k_emit over all rows)
    if (N) {
      f.add(b.order, 4 * ((size_t)N + 2), 0xffffffffu);
      f.add(b.first_child, 4 * (2 * (size_t)N + 4), 0xffffffffu);
      f.add(b.child_head, 4 * (2 * (size_t)N + 3), 0xffffffffu);
    }
  }
  if (extra)   // (the caller's own clears ride in the same launch)
    for (uint32_t k = 0; k < extra->n && f.n < 8; k++) { f.p[f.n] = extra->p[k]; f.n_words[f.n] = extra->n_words[k]; f.value[f.n] = extra->value[k]; f.n++; }
  launch_fill_ranges(f, aux);
}

// map emissions: LSD over (trigger id | key length | key chunks last..first | object)
static void order_map_emissions(MergeBufs& b, PatchIR& ir, const Counts* hc, hipStream_t st, uint32_t* ride_with_list_order = nullptr,
                                const MapKeyStats* ks = nullptr, bool trigger_passes = false) {
  uint32_t ne = hc->n_map_emit;
  if (!ne) return;
  uint32_t* perm_a = b.val_a;
  uint32_t* perm_b = b.val_b;
  int cur = 0;
  bool by_rank = false;
  bool last_is_chunk0 = false;  // the last pass run sorted by bytes of the keys' first chunk: the key buffer then holds every key's first eight bytes
  auto pass = [&](int mode, uint32_t chunk, int bits, int begin_bit = 0) {
    last_is_chunk0 = mode == MK_CHUNK && chunk == 0;
    uint32_t* pin = cur ? perm_b : perm_a;
    uint64_t* kin = cur ? b.key_b : b.key_a;
    const bool fused = sort_is_fused(ne) && bits > begin_bit;
    if (fused)
      hipLaunchKernelGGL(k_map_keys_hist, dim3(sort_tiles(ne)), dim3(BLOCK), 0, st, b, (const uint32_t*)pin, kin, ne, mode, chunk, (const uint32_t*)nullptr, begin_bit,
                         sort_first_table(b.sort_ws), sort_tiles(ne));
    else
      AM355_LAUNCH_INDEPENDENT(k_map_keys, grid_for(ne), dim3(BLOCK), st, b, (const uint32_t*)pin, kin, ne, mode, chunk, (const uint32_t*)nullptr);
    int res = cur ? radix_sort_pairs(b.key_b, perm_b, b.key_a, perm_a, ne, begin_bit, bits, b.sort_ws, st, fused)
                  : radix_sort_pairs(b.key_a, perm_a, b.key_b, perm_b, ne, begin_bit, bits, b.sort_ws, st, fused);
    cur ^= res;
  };
  // (measured in round 5 and not adopted: all passes of all key fields in ONE launch of one workgroup, 8-16 wavefronts each owning a
  // contiguous stretch of the pairs -- 400 us for the map workload's 17 k emissions against 150 us for the 22 tiled launches: a
  // wavefront's 64-item steps each wait a memory round trip, and keeping a lane's 24-48 pairs in registers spilled; profiles/r05_c3_*)
  if (ne <= BLOCK) {
    if (ride_with_list_order) *ride_with_list_order = ne;  // (an extra workgroup of k_list_order_objs: merge_run)
    else hipLaunchKernelGGL(k_map_small_finish, dim3(1), dim3(BLOCK), 0, st, b, ne, ir);
    return;
  }
  if (ne <= MAP_SORT_SMALL) {
    hipLaunchKernelGGL(k_map_sort_small, grid_for(ne), dim3(BLOCK), 0, st, b, ne, (const uint32_t*)nullptr, perm_b);  // (one emission: rank 0)
    cur = 1;
  } else {
    AM355_LAUNCH_INDEPENDENT(k_iota, grid_for(ne), dim3(BLOCK), st, perm_a, ne);
    // (AM355_MAP_TRIGGER_PASSES=1, or the second attempt after Counts.map_group_big: the values of a key ordered by radix passes over
    // their trigger ids, as in rounds 1-5; else by k_map_group_rank behind the key passes)
    static const bool trigger_env = getenv("AM355_MAP_TRIGGER_PASSES") != nullptr;
    by_rank = !(trigger_passes || trigger_env);
    if (!by_rank) pass(MK_TRIGGER, 0, b.bits_ctr + b.bits_actor);
    // (round 5, second session: a byte position -- or the length -- that is the same in every key is not sorted by: the pass would be
    // the identity. k_emit leaves what the keys have in common behind the counters, MapKeyStats; keys like `k00017` lose two of
    // their six byte passes and the length pass: three of the map workload's ten passes of two launches each)
    static const bool all_passes = getenv("AM355_MAP_ALL_PASSES") != nullptr;  // (A/B and tests)
    const bool have_stats = ks && !all_passes;
    if (!have_stats || (ks->or_len & ks->or_inv_len)) pass(MK_LEN, 0, bits_for(hc->max_key_len));
    uint32_t chunks = (hc->max_key_len + 7) / 8;
    for (uint32_t c = chunks; c-- > 0;) {
      // bytes are packed first-byte-highest: positions past the longest key are zero in every key -- no pass over them
      uint32_t used = hc->max_key_len - 8 * c < 8 ? hc->max_key_len - 8 * c : 8;
      auto varies = [&](uint32_t j) {  // byte j of chunk c
        const uint32_t pos = 8 * c + j;
        if (!have_stats || pos >= 16) return true;
        const uint32_t sh = 24 - 8 * (pos & 3);
        return (((ks->or_b[pos >> 2] & ks->or_inv_b[pos >> 2]) >> sh) & 0xffu) != 0;
      };
      // least significant byte first; every maximal stretch of varying bytes is one call (bits of byte j: [64 - 8 (j + 1), 64 - 8 j))
      for (int j = (int)used - 1; j >= 0;) {
        if (!varies((uint32_t)j)) { j--; continue; }
        int j_hi = j;
        while (j - 1 >= 0 && varies((uint32_t)(j - 1))) j--;
        pass(MK_CHUNK, c, 64 - 8 * j, 64 - 8 * (j_hi + 1));
        j--;
      }
    }
    // object index <= number of make rows (0 is _root); a document whose only map is _root needs no object pass at all
    if (hc->n_objects) pass(MK_OBJECT, 0, bits_for(hc->n_objects));
  }
  if (by_rank) {
    AM355_LAUNCH_INDEPENDENT(k_map_group_rank, grid_for(ne), dim3(BLOCK), st, b, (const uint32_t*)(cur ? perm_b : perm_a), cur ? perm_a : perm_b, ne,
                             last_is_chunk0 ? (const uint64_t*)(cur ? b.key_b : b.key_a) : (const uint64_t*)nullptr);
    cur ^= 1;
  }
  AM355_LAUNCH_INDEPENDENT(k_map_finish, grid_for(ne), dim3(BLOCK), st, b, (const uint32_t*)(cur ? perm_b : perm_a), ne, ir);
}

void merge_resolve(MergeBufs& b, hipStream_t st) {
  if (b.n_ops) hipLaunchKernelGGL(k_resolve, grid_for(std::max(b.n_ops - b.first_row, 1u)), dim3(BLOCK), 0, st, b);
}

void merge_run(MergeBufs& b, PatchIR& ir, Counts* hc, hipStream_t st, hipEvent_t ev_counts, hipEvent_t ev_runs, bool resolved) {
  const uint32_t N = b.n_ops;
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "  merge_run: %-24s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  Counts* hc_runs = hc + 1;  // second read-back (pinned host memory with room for two records)
  if (!N) {
    (void)hipMemsetAsync(ir.obj, 0, sizeof(am355_ir_object), st);
    (void)hipMemsetAsync(ir.edit, 0, sizeof(am355_ir_edit), st);
    (void)hipMemcpyAsync(hc, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    return;
  }
  uint32_t* is_head = b.scan_a;   // [ni]
  uint32_t* head_ex = b.scan_b;   // [ni]
  uint32_t* heads = b.run_heads;  // [runs + 1]
  uint32_t* row_run = b.row_run;  // [N]
  // ---- per-row resolution, visibility, compaction (b.counts already holds the decode kernels' validity flags) ----
  if (!resolved) hipLaunchKernelGGL(k_resolve, grid_for(std::max(N - b.first_row, 1u)), dim3(BLOCK), 0, st, b);
  hipLaunchKernelGGL(k_emit, grid_for(N), dim3(BLOCK), 0, st, b);
  hipLaunchKernelGGL(k_compact_rows, grid_for(N), dim3(BLOCK), 0, st, b, ir);
  if (!b.sig) (void)hipMemcpyAsync(hc, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st);
  if (ev_counts) (void)hipEventRecord(ev_counts, st);  // (also the boundary between the merge and the order phase in the statistics; null: not wanted)
  // ---- lists, first half: launched for the worst case (every row an insert) with the real count read on the device, so the
  //      host does not have to wait for the counters before the device has more work ----
  hipLaunchKernelGGL(k_child_push, grid_for(N), dim3(BLOCK), 0, st, b);
  hipLaunchKernelGGL(k_head_children, dim3(1), dim3(HEAD_CHILD_THREADS), 0, st, b);
  hipLaunchKernelGGL(k_child_order<false>, grid_for(N), dim3(BLOCK), 0, st, b, is_head);
  hipLaunchKernelGGL(k_run_heads, grid_for(N), dim3(BLOCK), 0, st, b, (const uint32_t*)is_head, head_ex, heads, row_run);
  if (!b.sig) {
    (void)hipMemcpyAsync(hc_runs, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st);
    (void)hipEventRecord(ev_runs, st);
  }
  AM355_LAUNCH_INDEPENDENT(k_euler_init_runs, grid_for(N), dim3(BLOCK), st, b, (const uint32_t*)heads, (const uint32_t*)row_run, b.euler_a);
  hipLaunchKernelGGL(k_euler_rank_lds, dim3(1), dim3(EULER_LDS_THREADS), 0, st, b.counts, b.euler_a);

  lap("first half enqueued");
  MapKeyStats key_stats;
  if (b.sig) read_phase_counts(b, &b.sig->counts_seq, b.sig->counts, hc, st, &key_stats);
  else (void)hipEventSynchronize(ev_counts);
  lap("counts read");
  if (hc->flags) { (void)hipStreamSynchronize(st); return; }
  const uint32_t ni = hc->n_list_ins, nu = hc->n_list_upd, n_obj = hc->n_objects + 1;
  uint32_t map_small = 0;  // a few map emissions: they ride with k_list_order_objs when that kernel is going to run
  order_map_emissions(b, ir, hc, st, ni && n_obj <= OBJ_LDS_MAX ? &map_small : nullptr, b.sig ? &key_stats : nullptr);

  if (ni) {
    if (b.sig) read_phase_counts(b, &b.sig->runs_seq, b.sig->runs, hc_runs, st);
    else (void)hipEventSynchronize(ev_runs);
    lap("runs read");
    uint32_t H = hc_runs->n_runs;
    ListKeyBits kb{bits_for(N), (int)b.bits_ctr, (int)b.bits_actor};
    if (hc_runs->pad) {
      // some list element has hundreds of children (e.g. everyone inserting at the same spot): order the siblings with one
      // radix sort of (is_head | parent | ~id) keys and redo the run detection from those links
      int total_bits = 1 + kb.b_row + kb.b_ctr + kb.b_actor;  // <= 64 verified by the caller
      (void)hipMemsetAsync(&b.counts->pad, 0, 3 * sizeof(uint32_t), st);  // pad, n_runs, euler_done
      (void)hipMemsetAsync(b.cs_runs.group_sum, 0, (size_t)((uint8_t*)b.counts + b.counts_bytes - (uint8_t*)b.cs_runs.group_sum), st);
      (void)hipMemsetAsync(b.first_child, 0xff, sizeof(uint32_t) * (2 * (size_t)N + 1), st);
      AM355_LAUNCH_INDEPENDENT(k_list_keys, grid_for(ni), dim3(BLOCK), st, b, b.key_a, b.val_a, ni, kb);
      int res = radix_sort_pairs(b.key_a, b.val_a, b.key_b, b.val_b, ni, 0, total_bits, b.sort_ws, st);
      AM355_LAUNCH_INDEPENDENT(k_list_link, grid_for(ni), dim3(BLOCK), st, b, (const uint64_t*)(res ? b.key_b : b.key_a), (const uint32_t*)(res ? b.val_b : b.val_a), ni, kb);
      hipLaunchKernelGGL(k_child_order<true>, grid_for(ni), dim3(BLOCK), 0, st, b, is_head);
      hipLaunchKernelGGL(k_run_heads, grid_for(ni), dim3(BLOCK), 0, st, b, (const uint32_t*)is_head, head_ex, heads, row_run);
      (void)hipMemcpyAsync(hc_runs, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st);
      AM355_LAUNCH_INDEPENDENT(k_euler_init_runs, grid_for(ni), dim3(BLOCK), st, b, (const uint32_t*)heads, (const uint32_t*)row_run, b.euler_a);
      hipLaunchKernelGGL(k_euler_rank_lds, dim3(1), dim3(EULER_LDS_THREADS), 0, st, b.counts, b.euler_a);
      (void)hipStreamSynchronize(st);
      H = hc_runs->n_runs;
    }
    // list ranking: done by the LDS kernel when the tour fits, else pointer-jumping rounds over the tour in HBM
    const unsigned long long* el = b.euler_a;
    if (2 * (uint64_t)H + 1 > EULER_LDS_ENTRIES) {
      int rounds = bits_for(2ull * H + 1);
      unsigned long long *e0 = b.euler_a, *e1 = b.euler_b;
      for (int r = 0; r < rounds; r++) {
        AM355_LAUNCH_INDEPENDENT(k_euler_jump, grid_for(2 * H + 1), dim3(BLOCK), st, H, (const unsigned long long*)e0, e1);
        unsigned long long* t = e0;
        e0 = e1;
        e1 = t;
      }
      el = e0;
    }
    // objects laid out one after another: elements per object from the ranked tour, prefix sum
    if (n_obj <= OBJ_LDS_MAX) {
      hipLaunchKernelGGL(k_list_order_objs, dim3(grid_for(ni).x + (map_small ? 1u : 0u)), dim3(BLOCK), 0, st, b, ir, n_obj, ni, (const uint32_t*)is_head,
                         (const uint32_t*)head_ex, (const uint32_t*)heads, (const uint32_t*)row_run, el, map_small);
    } else {
      AM355_LAUNCH_INDEPENDENT(k_obj_n, grid_for(n_obj + 1), dim3(BLOCK), st, b, ir, n_obj, (const uint32_t*)row_run, el);
      exclusive_scan_u32(b.obj_n, b.obj_first_pos, n_obj + 1, nullptr, b.scan_ws, st);
      AM355_LAUNCH_INDEPENDENT(k_list_order, grid_for(ni), dim3(BLOCK), st, b, ni, (const uint32_t*)is_head, (const uint32_t*)head_ex, (const uint32_t*)heads, el);
    }
    // visibility / edit-count prefix sums over document order
    uint32_t* vis_ex = b.scan_a;
    uint32_t* cnt_ex = b.scan_b;
    // counters / rows without a value inside lists (rare): the first such row of every element, in whichever half of the Euler
    // scratch the ranked tour is not in (both are free once the elements have their positions)
    unsigned long long* vl_min = nullptr;
    if (hc->n_quirk) {
      vl_min = el == b.euler_a ? b.euler_b : b.euler_a;
      (void)hipMemsetAsync(vl_min, 0xff, sizeof(unsigned long long) * (size_t)N, st);
      AM355_LAUNCH_INDEPENDENT(k_quirk_rows, grid_for(N), dim3(BLOCK), st, b, vl_min);
    }
    // (AM355_LIST_UNFUSED=1: the three-kernel form of rounds 2-5 -- counts, scan, edits over per-position arrays -- for A/B runs and tests)
    static const bool unfused = getenv("AM355_LIST_UNFUSED") != nullptr;
    uint32_t* vis_base = unfused ? nullptr : b.list_vis;  // [objects] (the per-position array of the other form: free here)
    if (unfused) {
      hipLaunchKernelGGL(k_list_counts, grid_for(ni), dim3(BLOCK), 0, st, b, ni, (const unsigned long long*)vl_min);
      hipLaunchKernelGGL(k_list_scan, grid_for(ni), dim3(BLOCK), 0, st, b, ni, vis_ex, cnt_ex);
    } else {
      hipLaunchKernelGGL(k_list_counts_pub, grid_for(ni), dim3(BLOCK), 0, st, b, ni, (const unsigned long long*)vl_min);
    }
    const uint64_t* uk = b.key_a;
    const uint32_t* uv = b.val_a;
    if (nu) {
      AM355_LAUNCH_INDEPENDENT(k_upd_keys, grid_for(nu), dim3(BLOCK), st, b, b.key_a, b.val_a, nu);
      int r2 = radix_sort_pairs(b.key_a, b.val_a, b.key_b, b.val_b, nu, 0, kb.b_row + kb.b_ctr + kb.b_actor, b.sort_ws, st);
      uk = r2 ? b.key_b : b.key_a;
      uv = r2 ? b.val_b : b.val_a;
    }
    if (unfused)
      AM355_LAUNCH_INDEPENDENT(k_list_edits, grid_for(ni), dim3(BLOCK), st, b, ni, (const uint32_t*)vis_ex, (const uint32_t*)cnt_ex, uk, uv, nu, ir,
                               (const unsigned long long*)vl_min);
    else
      hipLaunchKernelGGL(k_list_scan_edits, grid_for(ni), dim3(BLOCK), 0, st, b, ni, uk, uv, nu, ir, (const unsigned long long*)vl_min, vis_base);
    hipLaunchKernelGGL(k_edit_runs, grid_for(N), dim3(BLOCK), 0, st, b, ir);
    hipLaunchKernelGGL(k_edit_pack, grid_for(N), dim3(BLOCK), 0, st, b, ir, (const uint32_t*)vis_base);
  } else {
    (void)hipMemsetAsync(ir.edit, 0, sizeof(am355_ir_edit), st);
  }
  lap("all enqueued");
  if (b.sig && ni) {
    read_phase_counts(b, &b.sig->final_seq, b.sig->final_counts, hc, st);
  } else {
    (void)hipMemcpyAsync(hc, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
  }
  if (hc->map_group_big) {
    // some key holds more values than k_map_group_rank walks (hundreds of actors assigning one key concurrently): the emissions are
    // ordered once more, the values of a key by radix passes over their trigger ids
    if (trace) fprintf(stderr, "  merge_run: a key with more than %u values: map emissions ordered again with the trigger passes\n", MAP_GROUP_MAX);
    (void)hipMemsetAsync(&b.counts->map_group_big, 0, sizeof(uint32_t), st);
    order_map_emissions(b, ir, hc, st, nullptr, b.sig ? &key_stats : nullptr, true);
    (void)hipStreamSynchronize(st);
    hc->map_group_big = 0;
  }
  lap("done");
}

// The map half of merge_run alone, for a batch the caller knows to hold plain map rows only (replay_resident: `set` / `del` on string
// keys, no object made, no increment): visibility verdicts and map emissions of all rows (k_emit), the object table with its map ranges
// (k_compact_rows, k_map_finish), the map records in patch order. No list kernel runs: the stored list order, positions and per-object
// element counts stay what they are -- a batch without list rows cannot change them --, the whole-document EDIT tables are stale
// afterwards (the caller marks them so: ensure_ir_fresh). Counts of the list side in *hc are k_emit's / k_compact_rows' recount.
void merge_run_maps(MergeBufs& b, PatchIR& ir, Counts* hc, hipStream_t st) {
  const uint32_t N = b.n_ops;
  hipLaunchKernelGGL(k_emit, grid_for(N), dim3(BLOCK), 0, st, b);
  hipLaunchKernelGGL(k_compact_rows, grid_for(N), dim3(BLOCK), 0, st, b, ir);
  MapKeyStats key_stats;
  if (b.sig) read_phase_counts(b, &b.sig->counts_seq, b.sig->counts, hc, st, &key_stats);
  else { (void)hipMemcpyAsync(hc, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); }
  if (hc->flags) { (void)hipStreamSynchronize(st); return; }
  order_map_emissions(b, ir, hc, st, nullptr, b.sig ? &key_stats : nullptr);
  // (the counters behind the map order through the pinned words: a copy + blocking wait is an interrupt wake-up of 20-30 us)
  if (b.sig) {
    launch_signal_words((const uint32_t*)b.counts, (uint32_t)(sizeof(Counts) / 4), nullptr, 0, b.sig->final_counts, &b.sig->final_seq, b.sig_seq, st);
    read_phase_counts(b, &b.sig->final_seq, b.sig->final_counts, hc, st);
  } else {
    (void)hipMemcpyAsync(hc, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
  }
  if (hc->map_group_big) {
    (void)hipMemsetAsync(&b.counts->map_group_big, 0, sizeof(uint32_t), st);
    order_map_emissions(b, ir, hc, st, nullptr, b.sig ? &key_stats : nullptr, true);
    (void)hipStreamSynchronize(st);
    hc->map_group_big = 0;
  }
}

// Whole-document patch of canonical rows (document load). Synchronises the stream twice (counts).
void doc_patch(MergeBufs& b, PatchIR& ir, Counts* hc, hipStream_t st) {
  uint32_t N = b.n_ops;
  uint32_t cap = 1;  // id -> make-row table lives in the Euler scratch (2N+4 slots): largest power of two that fits
  while ((uint64_t)cap * 2 <= 2ull * N + 4) cap *= 2;
  uint32_t mask = cap - 1;
  unsigned long long* tab_key = b.euler_a;
  uint32_t* tab_row = b.first_child;
  uint32_t* obj_first = (uint32_t*)b.key_b;     // [N+1]
  uint32_t* trig_flag = b.order;
  uint32_t* trig_src = b.upd_row;
  uint32_t* edit_flag = b.next_sib;
  uint32_t* idx_ex = (uint32_t*)b.euler_b;      // [N+1]
  uint32_t* perm = b.val_a;
  uint32_t* vl_first = b.list_vis;              // [N] per insert row: first visible row of the element without a value (k_doc_emit)
  (void)hipMemsetAsync(vl_first, 0xff, sizeof(uint32_t) * (size_t)N, st);
  (void)hipMemsetAsync(b.zero_base, 0, b.zero_bytes, st);
  (void)hipMemsetAsync(tab_key, 0, sizeof(unsigned long long) * ((size_t)mask + 1), st);
  (void)hipMemsetAsync(obj_first, 0xff, sizeof(uint32_t) * ((size_t)N + 1), st);
  (void)hipMemsetAsync(trig_flag, 0, sizeof(uint32_t) * ((size_t)N + 1), st);
  if (N) {
    AM355_LAUNCH_INDEPENDENT(k_doc_prepare, grid_for(N), dim3(BLOCK), st, b, tab_key, tab_row, mask);
    exclusive_scan_u32(b.obj_index, b.scan_a, N, &b.counts->n_objects, b.scan_ws, st);
    AM355_LAUNCH_INDEPENDENT(k_object_table, grid_for(N), dim3(BLOCK), st, b, (const uint32_t*)b.scan_a, ir);
    exclusive_scan_u32(b.scan_b, b.scan_a, N, &b.counts->n_list_ins, b.scan_ws, st);
    AM355_LAUNCH_INDEPENDENT(k_ins_scatter, grid_for(N), dim3(BLOCK), st, b, (const uint32_t*)b.scan_b, (const uint32_t*)b.scan_a);
    AM355_LAUNCH_INDEPENDENT(k_doc_resolve, grid_for(N), dim3(BLOCK), st, b, (const unsigned long long*)tab_key, (const uint32_t*)tab_row, mask,
                             (const uint32_t*)b.scan_a, obj_first);
    AM355_LAUNCH_INDEPENDENT(k_doc_emit, grid_for(N), dim3(BLOCK), st, b, trig_flag, trig_src, edit_flag, vl_first);
    exclusive_scan_u32(trig_flag, b.scan_a, N, &b.counts->n_map_emit, b.scan_ws, st);
    AM355_LAUNCH_INDEPENDENT(k_doc_visflag, grid_for(N), dim3(BLOCK), st, b, b.ins_row, (const uint32_t*)vl_first, edit_flag);  // (may add `remove` edits)
    exclusive_scan_u32(edit_flag, b.scan_b, N, &b.counts->n_edits, b.scan_ws, st);
    exclusive_scan_u32(b.ins_row, idx_ex, N, nullptr, b.scan_ws, st);
    AM355_LAUNCH_INDEPENDENT(k_doc_scatter, grid_for(N), dim3(BLOCK), st, b, (const uint32_t*)trig_flag, (const uint32_t*)trig_src,
                             (const uint32_t*)b.scan_a, (const uint32_t*)edit_flag, (const uint32_t*)b.scan_b, (const uint32_t*)idx_ex,
                             (const uint32_t*)obj_first, perm, ir, (const uint32_t*)vl_first);
  } else {
    (void)hipMemsetAsync(ir.obj, 0, sizeof(am355_ir_object), st);
    (void)hipMemsetAsync(ir.edit, 0, sizeof(am355_ir_edit), st);
  }
  (void)hipMemcpyAsync(hc, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st);
  (void)hipStreamSynchronize(st);
  if (hc->flags || !N) return;
  if (hc->n_map_emit) AM355_LAUNCH_INDEPENDENT(k_map_finish, grid_for(hc->n_map_emit), dim3(BLOCK), st, b, (const uint32_t*)perm, hc->n_map_emit, ir);
  hipLaunchKernelGGL(k_edit_runs, grid_for(hc->n_edits ? hc->n_edits : 1), dim3(BLOCK), 0, st, b, ir);
  hipLaunchKernelGGL(k_edit_pack, grid_for(hc->n_edits ? hc->n_edits : 1), dim3(BLOCK), 0, st, b, ir, (const uint32_t*)nullptr);
  (void)hipMemcpyAsync(hc, b.counts, sizeof(Counts), hipMemcpyDeviceToHost, st);
  (void)hipStreamSynchronize(st);
}


// ---------------------------------------------------------------------------------------------------------
// Backend.save after a replay (new.js:2033-2055, columnar.js:983-1004): the op rows in the order a
// saved document holds them -- objects ascending by id with _root first, map rows by (key, op id), list rows in RGA
// order with each element's updates after it -- with their succ lists; `del` ops leave no row, only succ entries
// (new.js:1205-1217). The replay already knows every ingredient: object of each row, RGA position of each element,
// succ counts. What is left is sorting (map rows by key, updates by element, objects by id) and prefix sums.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void ks_classify(MergeBufs b, uint32_t* __restrict__ map_flag, uint32_t* __restrict__ upd_flag, uint32_t* __restrict__ upd_cnt,
                                                     uint32_t* __restrict__ max_key_len) {
  uint32_t g = gtid();
  if (g > b.n_ops) return;
  uint32_t mf = 0, uf = 0;
  if (g < b.n_ops) {
    uint8_t kind = b.kind[g];
    if (kind == K_MAP) { mf = 1; atomicMax(max_key_len, b.ops.key_len[g]); }
    else if (kind == K_LIST_UPD) { uf = 1; atomicAdd(&upd_cnt[b.ref_row[g]], 1u); }
  }
  map_flag[g] = mf;
  upd_flag[g] = uf;
}

__global__ __launch_bounds__(BLOCK) void ks_compact(MergeBufs b, const uint32_t* __restrict__ map_flag, const uint32_t* __restrict__ map_ex,
                                                    const uint32_t* __restrict__ upd_flag, const uint32_t* __restrict__ upd_ex) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  if (map_flag[g]) { b.em_row[map_ex[g]] = g; b.em_trig[map_ex[g]] = pack_id(b.ops.id_ctr[g], b.ops.id_actor[g]); }
  if (upd_flag[g]) b.upd_row[upd_ex[g]] = g;
}

__global__ __launch_bounds__(BLOCK) void ks_obj_keys(MergeBufs b, PatchIR ir, uint32_t n_obj, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint32_t i = gtid();  // object i + 1
  if (i + 1 >= n_obj) return;
  uint32_t row = ir.obj[i + 1].make_row;
  keys[i] = (uint64_t)b.ops.id_ctr[row] << b.bits_actor | b.ops.id_actor[row];
  vals[i] = i + 1;
}
__global__ __launch_bounds__(BLOCK) void ks_obj_rank(const uint32_t* __restrict__ sorted, uint32_t n_obj, uint32_t* __restrict__ obj_rank, uint32_t* __restrict__ rank_obj) {
  uint32_t i = gtid();
  if (i == 0) { obj_rank[0] = 0; rank_obj[0] = 0; }
  if (i + 1 >= n_obj) return;
  obj_rank[sorted[i]] = i + 1;
  rank_obj[i + 1] = sorted[i];
}

// first / one-past-last sorted map row of every object
__global__ __launch_bounds__(BLOCK) void ks_map_bounds(MergeBufs b, const uint32_t* __restrict__ perm, uint32_t n, uint32_t* __restrict__ begin, uint32_t* __restrict__ end) {
  uint32_t m = gtid();
  if (m >= n) return;
  uint32_t oi = obj_index_of(b, b.obj_row[b.em_row[perm[m]]]);
  if (m == 0 || obj_index_of(b, b.obj_row[b.em_row[perm[m - 1]]]) != oi) begin[oi] = m;
  if (m + 1 == n || obj_index_of(b, b.obj_row[b.em_row[perm[m + 1]]]) != oi) end[oi] = m + 1;
}

__global__ __launch_bounds__(BLOCK) void ks_list_counts(MergeBufs b, uint32_t n, const uint32_t* __restrict__ upd_cnt, uint32_t* __restrict__ pos_of, uint32_t* __restrict__ cnt) {
  uint32_t p = gtid();
  if (p > n) return;
  if (p == n) { cnt[p] = 0; return; }
  uint32_t v = b.order[p];
  pos_of[v] = p;
  cnt[p] = 1 + upd_cnt[v];
}
__global__ __launch_bounds__(BLOCK) void ks_list_bounds(MergeBufs b, uint32_t n, const uint32_t* __restrict__ list_off, uint32_t* __restrict__ begin, uint32_t* __restrict__ end) {
  uint32_t p = gtid();
  if (p >= n) return;
  uint32_t oi = obj_index_of(b, b.obj_row[b.order[p]]);
  if (p == 0 || obj_index_of(b, b.obj_row[b.order[p - 1]]) != oi) begin[oi] = list_off[p];
  if (p + 1 == n || obj_index_of(b, b.obj_row[b.order[p + 1]]) != oi) end[oi] = list_off[p + 1];
}
__global__ __launch_bounds__(BLOCK) void ks_upd_keys(MergeBufs b, uint32_t n, const uint32_t* __restrict__ pos_of, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint32_t i = gtid();
  if (i >= n) return;
  uint32_t g = b.upd_row[i];
  keys[i] = (uint64_t)pos_of[b.ref_row[g]] << (b.bits_ctr + b.bits_actor) | (uint64_t)b.ops.id_ctr[g] << b.bits_actor | b.ops.id_actor[g];
  vals[i] = g;
}
// rows of each object, in ascending object id order
__global__ __launch_bounds__(BLOCK) void ks_obj_counts(uint32_t n_obj, const uint32_t* __restrict__ rank_obj, const uint32_t* __restrict__ map_begin,
                                                       const uint32_t* __restrict__ map_end, const uint32_t* __restrict__ list_begin, const uint32_t* __restrict__ list_end,
                                                       uint32_t* __restrict__ cnt_by_rank) {
  uint32_t r = gtid();
  if (r > n_obj) return;
  uint32_t c = 0;
  if (r < n_obj) { uint32_t oi = rank_obj[r]; c = (map_end[oi] - map_begin[oi]) + (list_end[oi] - list_begin[oi]); }
  cnt_by_rank[r] = c;
}

__global__ __launch_bounds__(BLOCK) void ks_place_map(MergeBufs b, const uint32_t* __restrict__ perm, uint32_t n, const uint32_t* __restrict__ obj_rank,
                                                      const uint32_t* __restrict__ base_by_rank, const uint32_t* __restrict__ map_begin, uint32_t* __restrict__ final_pos,
                                                      uint32_t* __restrict__ src_of) {
  uint32_t m = gtid();
  if (m >= n) return;
  uint32_t g = b.em_row[perm[m]], oi = obj_index_of(b, b.obj_row[g]);
  uint32_t f = base_by_rank[obj_rank[oi]] + (m - map_begin[oi]);
  final_pos[g] = f;
  src_of[f] = g;
}
__global__ __launch_bounds__(BLOCK) void ks_place_ins(MergeBufs b, uint32_t n, const uint32_t* __restrict__ obj_rank, const uint32_t* __restrict__ base_by_rank,
                                                      const uint32_t* __restrict__ list_begin, const uint32_t* __restrict__ list_off, uint32_t* __restrict__ final_pos,
                                                      uint32_t* __restrict__ src_of) {
  uint32_t p = gtid();
  if (p >= n) return;
  uint32_t g = b.order[p], oi = obj_index_of(b, b.obj_row[g]);
  uint32_t f = base_by_rank[obj_rank[oi]] + (list_off[p] - list_begin[oi]);
  final_pos[g] = f;
  src_of[f] = g;
}
// sorted update j of the element at list position p sits at p + j + 1 of the chained list rows (p elements and j updates precede it)
__global__ __launch_bounds__(BLOCK) void ks_place_upd(MergeBufs b, const uint32_t* __restrict__ sorted, uint32_t n, const uint32_t* __restrict__ pos_of,
                                                      const uint32_t* __restrict__ obj_rank, const uint32_t* __restrict__ base_by_rank,
                                                      const uint32_t* __restrict__ list_begin, uint32_t* __restrict__ final_pos, uint32_t* __restrict__ src_of) {
  uint32_t j = gtid();
  if (j >= n) return;
  uint32_t g = sorted[j], oi = obj_index_of(b, b.obj_row[g]);
  uint32_t f = base_by_rank[obj_rank[oi]] + (pos_of[b.ref_row[g]] + j + 1 - list_begin[oi]);
  final_pos[g] = f;
  src_of[f] = g;
}

// canonical rows: gather + actor rank -> document actor index
__global__ __launch_bounds__(BLOCK) void ks_gather(MergeBufs b, uint32_t n_doc, const uint32_t* __restrict__ src_of, const uint32_t* __restrict__ doc_actor, OpCols out) {
  uint32_t f = gtid();
  if (f > n_doc) return;
  if (f == n_doc) { out.pred_num[f] = 0; return; }
  const OpCols& o = b.ops;
  uint32_t g = src_of[f];
  bool root = o.obj_actor[g] == NONE32;
  out.obj_actor[f] = root ? NONE32 : doc_actor[o.obj_actor[g]];
  out.obj_ctr[f] = root ? NONE32 : o.obj_ctr[g];
  out.key_actor[f] = o.key_actor[g] == NONE32 ? NONE32 : doc_actor[o.key_actor[g]];
  out.key_ctr[f] = o.key_ctr[g];
  out.key_off[f] = o.key_off[g];
  out.key_len[f] = o.key_len[g];
  out.id_actor[f] = doc_actor[o.id_actor[g]];
  out.id_ctr[f] = o.id_ctr[g];
  out.insert[f] = o.insert[g];
  out.action[f] = o.action[g];
  out.val_tl[f] = o.val_tl[g];
  out.val_off[f] = o.val_off[g];
  out.pred_num[f] = b.succ_cnt[g];
}

// one (target position, successor id) pair per pred entry of every op, `del`s included
__global__ __launch_bounds__(BLOCK) void ks_succ_pairs(MergeBufs b, const uint32_t* __restrict__ final_pos, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint32_t g = gtid();
  if (g >= b.n_ops) return;
  const OpCols& o = b.ops;
  uint32_t np = o.pred_num[g], pf = o.pred_first[g];
  for (uint32_t k = 0; k < np; k++) {
    uint32_t t = row_of(b, o.pred_actor[pf + k], o.pred_ctr[pf + k]);
    uint32_t f = t == NONE32 ? 0 : final_pos[t];
    keys[pf + k] = (uint64_t)f << (b.bits_ctr + b.bits_actor) | (uint64_t)o.id_ctr[g] << b.bits_actor | o.id_actor[g];
    vals[pf + k] = g;
  }
}
__global__ __launch_bounds__(BLOCK) void ks_succ_out(MergeBufs b, const uint32_t* __restrict__ sorted, uint32_t n, const uint32_t* __restrict__ doc_actor, OpCols out) {
  uint32_t j = gtid();
  if (j >= n) return;
  uint32_t g = sorted[j];
  out.pred_actor[j] = doc_actor[b.ops.id_actor[g]];
  out.pred_ctr[j] = b.ops.id_ctr[g];
}

void save_phase1(MergeBufs& b, SaveBufs& s, hipStream_t st) {
  uint32_t N = b.n_ops;
  (void)hipMemsetAsync(s.upd_cnt, 0, 4 * ((size_t)N + 1), st);
  (void)hipMemsetAsync(s.words, 0, 4 * 8, st);
  AM355_LAUNCH_INDEPENDENT(ks_classify, grid_for(N + 1), dim3(BLOCK), st, b, s.map_flag, s.upd_flag, s.upd_cnt, s.words + 2);
  exclusive_scan_u32(s.map_flag, s.map_ex, N + 1, s.words + 0, b.scan_ws, st);
  exclusive_scan_u32(s.upd_flag, s.upd_ex, N + 1, s.words + 1, b.scan_ws, st);
  if (N) AM355_LAUNCH_INDEPENDENT(ks_compact, grid_for(N), dim3(BLOCK), st, b, (const uint32_t*)s.map_flag, (const uint32_t*)s.map_ex, (const uint32_t*)s.upd_flag,
                                  (const uint32_t*)s.upd_ex);
}

// words (host copy of s.words): [0] map rows, [1] list update rows, [2] longest map key
void save_phase2(MergeBufs& b, PatchIR& ir, SaveBufs& s, const uint32_t* words, uint32_t n_obj, uint32_t n_ins, const uint32_t* doc_actor, hipStream_t st) {
  uint32_t N = b.n_ops, nm = words[0], nu = words[1], max_key = words[2];
  uint32_t n_doc = nm + nu + n_ins;
  // objects by id
  if (n_obj > 1) {
    AM355_LAUNCH_INDEPENDENT(ks_obj_keys, grid_for(n_obj), dim3(BLOCK), st, b, ir, n_obj, b.key_a, b.val_a);
    int res = radix_sort_pairs(b.key_a, b.val_a, b.key_b, b.val_b, n_obj - 1, 0, (int)(b.bits_ctr + b.bits_actor), b.sort_ws, st);
    AM355_LAUNCH_INDEPENDENT(ks_obj_rank, grid_for(n_obj), dim3(BLOCK), st, (const uint32_t*)(res ? b.val_b : b.val_a), n_obj, s.obj_rank, s.rank_obj);
  } else {
    (void)hipMemsetAsync(s.obj_rank, 0, 4, st);
    (void)hipMemsetAsync(s.rank_obj, 0, 4, st);
  }
  (void)hipMemsetAsync(s.map_begin, 0, 4 * 4 * ((size_t)n_obj + 1), st);  // map_begin, map_end, list_begin, list_end are contiguous
  // map rows by (object id, key, op id)
  const uint32_t* map_perm = b.val_a;
  if (nm) {
    uint32_t *perm_a = b.val_a, *perm_b = b.val_b;
    AM355_LAUNCH_INDEPENDENT(k_iota, grid_for(nm), dim3(BLOCK), st, perm_a, nm);
    int cur = 0;
    auto pass = [&](int mode, uint32_t chunk, int bits) {
      uint32_t* pin = cur ? perm_b : perm_a;
      uint64_t* kin = cur ? b.key_b : b.key_a;
      AM355_LAUNCH_INDEPENDENT(k_map_keys, grid_for(nm), dim3(BLOCK), st, b, (const uint32_t*)pin, kin, nm, mode, chunk, (const uint32_t*)s.obj_rank);
      int res = cur ? radix_sort_pairs(b.key_b, perm_b, b.key_a, perm_a, nm, 0, bits, b.sort_ws, st)
                    : radix_sort_pairs(b.key_a, perm_a, b.key_b, perm_b, nm, 0, bits, b.sort_ws, st);
      cur ^= res;
    };
    if (nm > 1 && nm <= MAP_SORT_SMALL) {
      hipLaunchKernelGGL(k_map_sort_small, grid_for(nm), dim3(BLOCK), 0, st, b, nm, (const uint32_t*)s.obj_rank, perm_b);
      cur = 1;
    } else if (nm > 1) {
      pass(MK_TRIGGER, 0, b.bits_ctr + b.bits_actor);
      pass(MK_LEN, 0, bits_for(max_key));
      uint32_t chunks = (max_key + 7) / 8;
      for (uint32_t c = chunks; c-- > 0;) pass(MK_CHUNK, c, 64);
      pass(MK_OBJECT, 0, bits_for(n_obj));
    }
    map_perm = cur ? perm_b : perm_a;
    // the permutation must survive the update sort below, which uses the same scratch: park it
    (void)hipMemcpyAsync(s.map_perm, map_perm, 4 * (size_t)nm, hipMemcpyDeviceToDevice, st);
    AM355_LAUNCH_INDEPENDENT(ks_map_bounds, grid_for(nm), dim3(BLOCK), st, b, (const uint32_t*)s.map_perm, nm, s.map_begin, s.map_end);
  }
  // list rows: elements in RGA order, each followed by its updates in ascending op id
  const uint32_t* upd_sorted = b.val_a;
  if (n_ins) {
    AM355_LAUNCH_INDEPENDENT(ks_list_counts, grid_for(n_ins + 1), dim3(BLOCK), st, b, n_ins, (const uint32_t*)s.upd_cnt, s.pos_of, s.list_off);
    exclusive_scan_u32(s.list_off, s.list_off, n_ins + 1, nullptr, b.scan_ws, st);
    AM355_LAUNCH_INDEPENDENT(ks_list_bounds, grid_for(n_ins), dim3(BLOCK), st, b, n_ins, (const uint32_t*)s.list_off, s.list_begin, s.list_end);
  }
  if (nu) {
    AM355_LAUNCH_INDEPENDENT(ks_upd_keys, grid_for(nu), dim3(BLOCK), st, b, nu, (const uint32_t*)s.pos_of, b.key_a, b.val_a);
    int res = radix_sort_pairs(b.key_a, b.val_a, b.key_b, b.val_b, nu, 0, bits_for(n_ins) + (int)(b.bits_ctr + b.bits_actor), b.sort_ws, st);
    upd_sorted = res ? b.val_b : b.val_a;
  }
  AM355_LAUNCH_INDEPENDENT(ks_obj_counts, grid_for(n_obj + 1), dim3(BLOCK), st, n_obj, (const uint32_t*)s.rank_obj, (const uint32_t*)s.map_begin, (const uint32_t*)s.map_end,
                           (const uint32_t*)s.list_begin, (const uint32_t*)s.list_end, s.base_by_rank);
  exclusive_scan_u32(s.base_by_rank, s.base_by_rank, n_obj + 1, nullptr, b.scan_ws, st);
  (void)hipMemsetAsync(s.final_pos, 0xff, 4 * ((size_t)N + 1), st);
  if (nm) AM355_LAUNCH_INDEPENDENT(ks_place_map, grid_for(nm), dim3(BLOCK), st, b, (const uint32_t*)s.map_perm, nm, (const uint32_t*)s.obj_rank, (const uint32_t*)s.base_by_rank,
                                   (const uint32_t*)s.map_begin, s.final_pos, s.src_of);
  if (n_ins) AM355_LAUNCH_INDEPENDENT(ks_place_ins, grid_for(n_ins), dim3(BLOCK), st, b, n_ins, (const uint32_t*)s.obj_rank, (const uint32_t*)s.base_by_rank,
                                      (const uint32_t*)s.list_begin, (const uint32_t*)s.list_off, s.final_pos, s.src_of);
  if (nu) AM355_LAUNCH_INDEPENDENT(ks_place_upd, grid_for(nu), dim3(BLOCK), st, b, upd_sorted, nu, (const uint32_t*)s.pos_of, (const uint32_t*)s.obj_rank,
                                   (const uint32_t*)s.base_by_rank, (const uint32_t*)s.list_begin, s.final_pos, s.src_of);
  AM355_LAUNCH_INDEPENDENT(ks_gather, grid_for(n_doc + 1), dim3(BLOCK), st, b, n_doc, (const uint32_t*)s.src_of, doc_actor, s.out);
  exclusive_scan_u32(s.out.pred_num, s.out.pred_first, n_doc + 1, nullptr, b.scan_ws, st);
  // succ lists: every pred entry names (row it overwrites, id of the overwriting op); sorted by canonical position, then id
  uint32_t P = b.n_preds;
  if (P) {
    AM355_LAUNCH_INDEPENDENT(ks_succ_pairs, grid_for(N), dim3(BLOCK), st, b, (const uint32_t*)s.final_pos, s.succ_key_a, s.succ_val_a);
    int res = radix_sort_pairs(s.succ_key_a, s.succ_val_a, s.succ_key_b, s.succ_val_b, P, 0, bits_for(n_doc) + (int)(b.bits_ctr + b.bits_actor), b.sort_ws, st);
    AM355_LAUNCH_INDEPENDENT(ks_succ_out, grid_for(P), dim3(BLOCK), st, b, (const uint32_t*)(res ? s.succ_val_b : s.succ_val_a), P, doc_actor, s.out);
  }
}

}  // namespace am355
