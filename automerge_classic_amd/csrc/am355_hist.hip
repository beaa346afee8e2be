// History reconstruction after Backend.load, device part.
//
// Reference: backend/new.js:1887-1912 computeHashGraph -> backend/columnar.js:1040-1047 decodeDocument, :876-943 groupChangeOps
// (ops -> changes by (actor, maxOp); `del` ops rebuilt from succ entries; preds = inverse of succ), :945-981 decodeDocumentChanges,
// :710-739 encodeChange, :370-444 encodeOps, :122-170 parseAllOpIds (change-local actor table: author first, the others sorted).
//
// A saved document holds the merged op rows, not its changes. The rows are in HBM after the load (am355_bigcol.hip); here:
//   1. ids -> slots: one bit per (actor, counter) that a row carries or a succ list names; popcount + prefix sum ranks the bits, which
//      numbers the ids per actor in counter order -- the order of ops inside a change and of changes inside an actor. An id no row
//      carries is a deletion: groupChangeOps rebuilds a `del` op for it from the first row that lists it (kh_mark, kh_popc, kh_slots);
//   2. preds = the inverse of the succ lists: counts per slot, prefix sum, fill, order by (counter, actor) (kh_pred_fill, kh_pred_sort);
//   3. a change owns the ids in (maxOp of its actor's previous change, maxOp]: its slot range by two rank queries (kh_changes);
//   4. the actors a change mentions (bitmap per change; the change-local index of an actor = its rank among them, author first) and
//      the twelve op columns of ALL changes as arrays by slot (kh_refs, kh_rows), validated as the reference's decode / re-encode
//      round trip would;
//   5. the column encoders of Backend.save (am355_encode.hip), segmented by change: one pass per column encodes that column of every
//      change (hist_stage2).
// The host (am355_history.cpp) reads the change metadata columns (a few thousand values), writes the headers, copies the column
// pieces into place and chains the hashes: a change's header holds the hashes of its dependencies, so SHA-256 runs level by level
// of the dependency graph, one dependent 64-byte block after the other per change -- 16 MB/s for a lane of this device against
// 2 GB/s for a host core with SHA extensions; the chain stays on the host threads.
#include "am355_hist.h"
#include "am355_prims.h"

namespace am355 {

static inline dim3 grid_for(uint32_t n) { return dim3((n + BLOCK - 1) / BLOCK); }
static size_t al256(size_t b) { return carve_round(b); }

static size_t hist_col_cap(int k, uint32_t M, uint32_t P, size_t key_bytes, size_t val_bytes) {
  // keyStr: headers + length prefixes + the key bytes. `key_bytes` is the caller's estimate -- the document's own key column holds a
  // repeated key once, the changes hold it once per change, so the rebuilt columns can be far longer than it; the encoder never
  // writes past this bound and reports the size it needs, with which the caller binds again (am355_doc_changes)
  if (k == 4) return enc_numbers_bound(M) + key_bytes + 16;
  if (k == 8) return val_bytes + 16;                          // valRaw
  return enc_numbers_bound(k >= 10 ? P : M);
}

size_t hist_bytes(uint32_t N, uint32_t P, uint32_t NC, uint32_t NA, uint32_t W, size_t key_bytes, size_t val_bytes) {
  const size_t M = (size_t)N + P + 1, PP = (size_t)P + 1, C = (size_t)NC + 1, AW = ((size_t)NA + 31) / 32, big = (M > PP ? M : PP) + 2;
  size_t b = al256(16) + al256(4 * ((size_t)NA + 1)) + al256(4 * (size_t)NA + 4) + 5 * al256(4 * C) + 4 * al256(4 * ((size_t)W + 2));
  b += 2 * al256(4 * M) + 3 * al256(4 * (M + 1)) + al256(4 * PP) + 2 * al256(4 * C) + al256(4 * C * AW);
  b += 12 * al256(4 * M) + al256(M) + 3 * al256(4 * PP) + 2 * al256(4 * (C + 1));
  b += al256(enc_work_bytes((uint32_t)big)) + al256(4 * big) + al256(big) + al256(4 * HIST_NCOL * 2 * (C + 1)) + al256(4 * HIST_NCOL);
  for (int k = 0; k < HIST_NCOL; k++) b += al256(hist_col_cap(k, (uint32_t)M, (uint32_t)PP, key_bytes, val_bytes));
  b += al256(scan_workspace_bytes((uint32_t)std::max<size_t>(big, (size_t)W + 2)));
  return b + 4096;
}

void hist_bind(HistBufs& h, void* block, uint32_t N, uint32_t P, uint32_t NC, uint32_t NA, uint32_t W, size_t key_bytes, size_t val_bytes) {
  canary_scope("history reconstruction (hist_bind)");
  canary_forget(block, hist_bytes(N, P, NC, NA, W, key_bytes, val_bytes));
  const size_t M = (size_t)N + P + 1, PP = (size_t)P + 1, C = (size_t)NC + 1, AW = ((size_t)NA + 31) / 32, big = (M > PP ? M : PP) + 2;
  uint8_t* p = (uint8_t*)block;
  auto take = [&](size_t bytes) { void* r = p; canary_note(p, bytes); p += al256(bytes); return r; };
  h.N = N; h.P = P; h.NC = NC; h.NA = NA; h.W = W; h.AW = (uint32_t)AW;
  h.flags = (uint32_t*)take(16);
  h.word_base = (uint32_t*)take(4 * ((size_t)NA + 1)); h.act_max = (uint32_t*)take(4 * (size_t)NA + 4);
  h.chg_actor = (uint32_t*)take(4 * C); h.chg_prev_max = (uint32_t*)take(4 * C); h.chg_max = (uint32_t*)take(4 * C);
  h.sorted_base = (uint32_t*)take(4 * C); h.sorted_chg = (uint32_t*)take(4 * C);
  h.all_bits = (uint32_t*)take(4 * ((size_t)W + 2)); h.row_bits = (uint32_t*)take(4 * ((size_t)W + 2));
  h.word_cnt = (uint32_t*)take(4 * ((size_t)W + 2)); h.word_rank = (uint32_t*)take(4 * ((size_t)W + 2));
  h.slot_row = (uint32_t*)take(4 * M); h.slot_ref = (uint32_t*)take(4 * M);
  h.pred_cnt = (uint32_t*)take(4 * (M + 1)); h.pred_first = (uint32_t*)take(4 * (M + 1)); h.pred_cur = (uint32_t*)take(4 * (M + 1));
  h.pred_row = (uint32_t*)take(4 * PP);
  h.chg_base = (uint32_t*)take(4 * C); h.chg_nops = (uint32_t*)take(4 * C);
  h.abits = (uint32_t*)take(4 * C * AW);
  h.seg = (uint32_t*)take(4 * M); h.chg_of = (uint32_t*)take(4 * M);
  uint32_t** vs[] = {&h.v_obj_actor, &h.v_obj_ctr, &h.v_key_actor, &h.v_key_ctr, &h.v_key_off, &h.v_key_len, &h.v_action, &h.v_val_tl, &h.v_val_off, &h.v_pred_num};
  for (uint32_t** v : vs) *v = (uint32_t*)take(4 * M);
  h.v_insert = (uint8_t*)take(M);
  h.p_actor = (uint32_t*)take(4 * PP); h.p_ctr = (uint32_t*)take(4 * PP); h.pseg = (uint32_t*)take(4 * PP);
  h.seg_base = (uint32_t*)take(4 * (C + 1)); h.pseg_base = (uint32_t*)take(4 * (C + 1));
  {
    void* w = take(enc_work_bytes((uint32_t)big));
    enc_carve(h.enc, w, (uint32_t)big);
    canary_scope("history reconstruction (hist_bind, behind the encoder work)");
  }
  h.deltas = (uint32_t*)take(4 * big);
  h.nullmask = (uint8_t*)take(big);
  h.col_off = (uint32_t*)take(4 * HIST_NCOL * 2 * (C + 1));
  h.col_len = (uint32_t*)take(4 * HIST_NCOL);
  for (int k = 0; k < HIST_NCOL; k++) {
    h.col_cap[k] = hist_col_cap(k, (uint32_t)M, (uint32_t)PP, key_bytes, val_bytes);
    h.col_out[k] = (uint8_t*)take(h.col_cap[k]);
  }
  h.scan_ws = take(scan_workspace_bytes((uint32_t)std::max<size_t>(big, (size_t)W + 2)));
}

// ---------------------------------------------------------------------------------------------------------
// ids -> slots
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool id_in_range(const HistBufs& h, uint32_t a, uint32_t ctr) { return a < h.NA && ctr != 0 && ctr <= h.act_max[a]; }

__global__ __launch_bounds__(BLOCK) void kh_mark(OpCols rows, HistBufs h) {
  const uint32_t i = gtid();
  if (i < h.N) {
    const uint32_t a = rows.id_actor[i], ctr = rows.id_ctr[i];
    if (!id_in_range(h, a, ctr)) atomicOr(&h.flags[0], (uint32_t)HF_INVALID);   // operation id outside of the range its actor's changes allow
    else {
      const uint32_t w = h.word_base[a] + ctr / 32, bit = 1u << (ctr % 32);
      if (atomicOr(&h.row_bits[w], bit) & bit) atomicOr(&h.flags[0], (uint32_t)HF_INVALID);   // two rows carry one operation id
      atomicOr(&h.all_bits[w], bit);
    }
    if ((uint64_t)rows.pred_first[i] + rows.pred_num[i] > h.P) atomicOr(&h.flags[0], (uint32_t)HF_INVALID);  // succ lists exceed the succ columns
  }
  if (i < h.P) {
    const uint32_t a = rows.pred_actor[i], ctr = rows.pred_ctr[i];   // (a document: the succ columns)
    if (!id_in_range(h, a, ctr)) atomicOr(&h.flags[0], (uint32_t)HF_INVALID);
    else atomicOr(&h.all_bits[h.word_base[a] + ctr / 32], 1u << (ctr % 32));
  }
}

__global__ __launch_bounds__(BLOCK) void kh_popc(HistBufs h) {
  const uint32_t w = gtid();
  if (w <= h.W) h.word_cnt[w] = w < h.W ? (uint32_t)__popc(h.all_bits[w]) : 0u;
}

// ids of actor a with a smaller counter, plus the actor's first slot = the slot of (a, ctr) when that id exists
__device__ __forceinline__ uint32_t slot_of(const HistBufs& h, uint32_t a, uint32_t ctr) {
  const uint32_t w = h.word_base[a] + ctr / 32;
  return h.word_rank[w] + (uint32_t)__popc(h.all_bits[w] & ((1u << (ctr % 32)) - 1u));
}

__global__ __launch_bounds__(BLOCK) void kh_slots(OpCols rows, HistBufs h) {
  const uint32_t r = gtid();
  if (r >= h.N) return;
  if (h.flags[0] & HF_INVALID) return;
  h.slot_row[slot_of(h, rows.id_actor[r], rows.id_ctr[r])] = r;
  const uint32_t f = rows.pred_first[r], n = rows.pred_num[r];
  for (uint32_t e = f; e < f + n; e++) {
    const uint32_t s = slot_of(h, rows.pred_actor[e], rows.pred_ctr[e]);
    atomicMin(&h.slot_ref[s], r);
    atomicAdd(&h.pred_cnt[s], 1u);
  }
}

__global__ __launch_bounds__(BLOCK) void kh_pred_fill(OpCols rows, HistBufs h) {
  const uint32_t r = gtid();
  if (r >= h.N) return;
  if (h.flags[0] & HF_INVALID) return;
  const uint32_t f = rows.pred_first[r], n = rows.pred_num[r];
  for (uint32_t e = f; e < f + n; e++) {
    const uint32_t s = slot_of(h, rows.pred_actor[e], rows.pred_ctr[e]);
    h.pred_row[h.pred_first[s] + atomicAdd(&h.pred_cur[s], 1u)] = r;
  }
}

// the preds of an op ascend by (counter, actor) (columnar.js:394-396 sorts them when a change is encoded)
__global__ __launch_bounds__(BLOCK) void kh_pred_sort(OpCols rows, HistBufs h, uint32_t M) {
  const uint32_t s = gtid();
  if (s >= M) return;
  if (h.flags[0] & HF_INVALID) return;
  const uint32_t lo = h.pred_first[s], hi = h.pred_first[s + 1];
  for (uint32_t i = lo + 1; i < hi; i++) {
    const uint32_t r = h.pred_row[i];
    const unsigned long long key = (unsigned long long)rows.id_ctr[r] << 32 | rows.id_actor[r];
    uint32_t j = i;
    for (; j > lo; j--) {
      const uint32_t q = h.pred_row[j - 1];
      if (((unsigned long long)rows.id_ctr[q] << 32 | rows.id_actor[q]) <= key) break;
      h.pred_row[j] = q;
    }
    h.pred_row[j] = r;
  }
}

__global__ __launch_bounds__(BLOCK) void kh_changes(HistBufs h) {
  const uint32_t k = gtid();
  if (k >= h.NC) return;
  if (h.flags[0] & HF_INVALID) { h.chg_base[k] = 0; h.chg_nops[k] = 0; return; }
  const uint32_t a = h.chg_actor[k], mx = h.chg_max[k], prev = h.chg_prev_max[k];
  const uint32_t base = slot_of(h, a, prev + 1), n = slot_of(h, a, mx + 1) - base;
  h.chg_base[k] = base;
  h.chg_nops[k] = n;
  bool bad = n > mx;   // more operations than maxOp allows
  // ids must be startOp .. maxOp without a gap (columnar.js:935-939)
  if (!bad && n && slot_of(h, a, mx - n + 1) != base) bad = true;
  if (bad) atomicOr(&h.flags[0], (uint32_t)HF_INVALID);
  if (k == 0) { h.flags[2] = h.word_rank[h.W]; h.flags[3] = h.pred_first[h.word_rank[h.W]]; }
}

void hist_stage1(const OpCols& rows, HistBufs& h, hipStream_t st) {
  const uint32_t Mcap = h.N + h.P + 1;
  (void)hipMemsetAsync(h.flags, 0, 16, st);
  (void)hipMemsetAsync(h.all_bits, 0, 4 * ((size_t)h.W + 2), st);
  (void)hipMemsetAsync(h.row_bits, 0, 4 * ((size_t)h.W + 2), st);
  (void)hipMemsetAsync(h.slot_row, 0xff, 4 * (size_t)Mcap, st);
  (void)hipMemsetAsync(h.slot_ref, 0xff, 4 * (size_t)Mcap, st);
  (void)hipMemsetAsync(h.pred_cnt, 0, 4 * ((size_t)Mcap + 1), st);
  (void)hipMemsetAsync(h.pred_cur, 0, 4 * ((size_t)Mcap + 1), st);
  const uint32_t np = h.N > h.P ? h.N : h.P;
  if (np) AM355_LAUNCH_INDEPENDENT(kh_mark, grid_for(np), dim3(BLOCK), st, rows, h);
  AM355_LAUNCH_INDEPENDENT(kh_popc, grid_for(h.W + 1), dim3(BLOCK), st, h);
  exclusive_scan_u32(h.word_cnt, h.word_rank, h.W + 1, nullptr, h.scan_ws, st);
  if (h.N) AM355_LAUNCH_INDEPENDENT(kh_slots, grid_for(h.N), dim3(BLOCK), st, rows, h);
  exclusive_scan_u32(h.pred_cnt, h.pred_first, Mcap + 1, nullptr, h.scan_ws, st);
  if (h.N) AM355_LAUNCH_INDEPENDENT(kh_pred_fill, grid_for(h.N), dim3(BLOCK), st, rows, h);
  AM355_LAUNCH_INDEPENDENT(kh_pred_sort, grid_for(Mcap), dim3(BLOCK), st, rows, h, Mcap);
  if (h.NC) AM355_LAUNCH_INDEPENDENT(kh_changes, grid_for(h.NC), dim3(BLOCK), st, h);
}

// ---------------------------------------------------------------------------------------------------------
// the changes' op columns
// ---------------------------------------------------------------------------------------------------------
// change that owns slot s: the last entry of sorted_base <= s (only changes that own slots are listed)
__device__ __forceinline__ uint32_t change_of_slot(const HistBufs& h, uint32_t n_sorted, uint32_t s) {
  uint32_t lo = 0, hi = n_sorted;
  while (lo + 1 < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (h.sorted_base[mid] <= s) lo = mid; else hi = mid;
  }
  return h.sorted_chg[lo];
}

__device__ __forceinline__ void touch(const HistBufs& h, uint32_t k, uint32_t a) { atomicOr(&h.abits[(size_t)k * h.AW + a / 32], 1u << (a % 32)); }

// index of actor a in the actor table of change k: the author first, the others in id order = rank order (columnar.js:154-157)
__device__ __forceinline__ uint32_t local_actor(const HistBufs& h, uint32_t k, uint32_t a) {
  const uint32_t author = h.chg_actor[k];
  if (a == author) return 0;
  const uint32_t* bits = h.abits + (size_t)k * h.AW;
  uint32_t n = 0;
  for (uint32_t w = 0; w < a / 32; w++) n += (uint32_t)__popc(bits[w]);
  n += (uint32_t)__popc(bits[a / 32] & ((1u << (a % 32)) - 1u));
  return author < a ? n : n + 1;   // (the author's own bit is set as well)
}

// pass 1 over the slots: which change, which witness row, which actors the change mentions
__global__ __launch_bounds__(BLOCK) void kh_refs(OpCols rows, HistBufs h, uint32_t n_sorted, uint32_t M) {
  const uint32_t s = gtid();
  if (s >= M) return;
  const uint32_t k = change_of_slot(h, n_sorted, s);
  h.chg_of[s] = k;
  h.seg[s] = h.chg_base[k];
  const uint32_t r = h.slot_row[s];
  const uint32_t q = r != NONE32 ? r : h.slot_ref[s];
  if (q == NONE32) { atomicOr(&h.flags[0], (uint32_t)HF_INVALID); return; }   // operation without a row
  touch(h, k, h.chg_actor[k]);
  if (rows.obj_actor[q] != NONE32) {
    if (rows.obj_actor[q] >= h.NA) { atomicOr(&h.flags[0], (uint32_t)HF_INVALID); return; }
    touch(h, k, rows.obj_actor[q]);
  }
  if (rows.key_len[q] == NONE32) {
    if (r == NONE32 && rows.insert[q]) touch(h, k, rows.id_actor[q]);
    else if (rows.key_ctr[q] != 0 && rows.key_ctr[q] != NONE32) {
      if (rows.key_actor[q] >= h.NA) { atomicOr(&h.flags[0], (uint32_t)HF_INVALID); return; }
      touch(h, k, rows.key_actor[q]);
    }
  }
  for (uint32_t e = h.pred_first[s]; e < h.pred_first[s + 1]; e++) touch(h, k, rows.id_actor[h.pred_row[e]]);
}

// valid UTF-8 that TextDecoder -> TextEncoder reproduces (a leading U+FEFF is dropped by the decoder, encoding.js:9-17)
__device__ bool hist_valid_utf8(const uint8_t* s, uint32_t n) {
  if (n >= 3 && s[0] == 0xef && s[1] == 0xbb && s[2] == 0xbf) return false;
  uint32_t i = 0;
  while (i < n) {
    const uint8_t c = s[i];
    if (c < 0x80) { i++; continue; }
    uint32_t extra, cp;
    if ((c & 0xe0) == 0xc0) { extra = 1; cp = c & 0x1f; }
    else if ((c & 0xf0) == 0xe0) { extra = 2; cp = c & 0x0f; }
    else if ((c & 0xf8) == 0xf0) { extra = 3; cp = c & 0x07; }
    else return false;
    if (i + extra >= n) return false;
    for (uint32_t j = 1; j <= extra; j++) {
      if ((s[i + j] & 0xc0) != 0x80) return false;
      cp = cp << 6 | (s[i + j] & 0x3f);
    }
    if ((extra == 1 && cp < 0x80) || (extra == 2 && cp < 0x800) || (extra == 3 && (cp < 0x10000 || cp > 0x10ffff)) || (cp >= 0xd800 && cp <= 0xdfff)) return false;
    i += extra + 1;
  }
  return true;
}

// A value the reference's decodeValue -> encodeValue round trip (columnar.js:259-329) reproduces byte for byte? 0 yes; HF_INVALID: the
// reference throws on it; HF_UNSUPPORTED: the reference writes something else (numbers not in minimal form, byte arrays -- encodeValue
// writes the whole underlying buffer of a decoded byte array --, unknown type tags)
__device__ uint32_t hist_value_round_trips(uint32_t tl, const uint8_t* b) {
  const uint32_t tag = tl & 15, len = tl >> 4;
  switch (tag) {
    case 0: case 1: case 2: return len ? (uint32_t)HF_UNSUPPORTED : 0u;
    case 3: case 4: case 8: case 9: {
      // one LEB128 number filling the value exactly, inside +-2^53, in its shortest form
      if (len == 0 || len > 10) return HF_INVALID;
      unsigned long long v = 0;
      uint32_t shift = 0, i = 0;
      for (; i < len; i++) {
        const uint8_t c = b[i];
        if (shift < 64) v |= (unsigned long long)(c & 0x7f) << shift;
        shift += 7;
        if (!(c & 0x80)) break;
      }
      if (i + 1 != len) return HF_INVALID;   // ends early or runs past the value
      if (tag == 3) {
        if (shift > 63 || v >= (1ull << 53)) return HF_INVALID;
        uint32_t need = 1;
        for (unsigned long long x = v; x >= 0x80; x >>= 7) need++;
        return need == len ? 0u : (uint32_t)HF_UNSUPPORTED;
      }
      long long sv = (long long)v;
      if ((b[len - 1] & 0x40) && shift < 64) sv = (long long)(v | (~0ull << shift));
      if (shift > 63 + 7 || sv >= (1ll << 53) || sv <= -(1ll << 53)) return HF_INVALID;
      uint32_t need = 1;
      for (long long x = sv;; need++) {
        const uint8_t c = (uint8_t)(x & 0x7f);
        x >>= 7;
        if ((x == 0 && !(c & 0x40)) || (x == -1 && (c & 0x40))) break;
      }
      return need == len ? 0u : (uint32_t)HF_UNSUPPORTED;
    }
    case 5: return len == 8 ? 0u : (uint32_t)HF_INVALID;
    case 6: return hist_valid_utf8(b, len) ? 0u : (uint32_t)HF_UNSUPPORTED;
    default: return HF_UNSUPPORTED;
  }
}

// pass 2 over the slots: the op of the slot as the values of its change's columns (encodeOps, columnar.js:370-444)
__global__ __launch_bounds__(BLOCK) void kh_rows(OpCols rows, const uint8_t* __restrict__ arena, unsigned long long arena_len, HistBufs h, uint32_t M) {
  const uint32_t s = gtid();
  if (s >= M) return;
  if (h.flags[0] & HF_INVALID) return;
  const uint32_t k = h.chg_of[s], r = h.slot_row[s];
  const bool del = r == NONE32;
  const uint32_t q = del ? h.slot_ref[s] : r;
  uint32_t err = 0;
  uint32_t oa = NONE32, oc = NONE32, ka = NONE32, kc = NONE32, koff = 0, klen = NONE32;
  if (rows.obj_actor[q] != NONE32) { oa = local_actor(h, k, rows.obj_actor[q]); oc = rows.obj_ctr[q]; }
  if (rows.key_len[q] != NONE32) {
    klen = rows.key_len[q]; koff = rows.key_off[q];
    if (klen == 0) err |= HF_UNSUPPORTED;                                           // empty map key
    else if ((unsigned long long)koff + klen > arena_len) err |= HF_INVALID;          // key outside the arena
    else if (!hist_valid_utf8(arena + koff, klen)) err |= HF_UNSUPPORTED;             // key is not valid UTF-8
  } else if (del && rows.insert[q]) {   // deleting the element the witness row inserted
    ka = local_actor(h, k, rows.id_actor[q]); kc = rows.id_ctr[q];
  } else if (rows.key_ctr[q] == 0) {
    if (del || !rows.insert[q]) err |= HF_INVALID;                                  // operation on _head that is not an insertion
    kc = 0;
  } else if (rows.key_ctr[q] != NONE32) {
    ka = local_actor(h, k, rows.key_actor[q]); kc = rows.key_ctr[q];
  } else err |= HF_INVALID;                                                          // operation without a key
  const uint32_t act = del ? 3u : rows.action[q];
  if (!del && act == 3) err |= HF_INVALID;                                           // a document holds no del operations
  if (act >= 7) err |= HF_UNSUPPORTED;                                               // link or unknown action
  uint32_t tl = 0, voff = 0;
  if (!del && (act == 1 || act == 5)) {
    tl = rows.val_tl[q]; voff = rows.val_off[q];
    const uint32_t len = tl >> 4;
    if (len && (unsigned long long)voff + len > arena_len) err |= HF_INVALID;         // value outside the arena
    else err |= hist_value_round_trips(tl, arena + voff);
  }
  h.v_obj_actor[s] = oa; h.v_obj_ctr[s] = oc; h.v_key_actor[s] = ka; h.v_key_ctr[s] = kc; h.v_key_off[s] = koff; h.v_key_len[s] = klen;
  h.v_insert[s] = del ? 0 : rows.insert[q];
  h.v_action[s] = act; h.v_val_tl[s] = tl; h.v_val_off[s] = voff;
  const uint32_t pf = h.pred_first[s], pn = h.pred_first[s + 1] - pf;
  h.v_pred_num[s] = pn;
  const uint32_t pbase = h.pred_first[h.chg_base[k]];
  for (uint32_t e = pf; e < pf + pn; e++) {
    const uint32_t pr = h.pred_row[e];
    h.p_actor[e] = local_actor(h, k, rows.id_actor[pr]);
    h.p_ctr[e] = rows.id_ctr[pr];
    h.pseg[e] = pbase;
  }
  if (err) atomicOr(&h.flags[0], err);
}

// first slot / first pred entry of every change in DOCUMENT order of the changes (a change without ops: an empty range)
__global__ __launch_bounds__(BLOCK) void kh_seg_bases(HistBufs h, uint32_t M) {
  const uint32_t k = gtid();
  if (k > h.NC) return;
  if (k == h.NC) { h.seg_base[k] = M; h.pseg_base[k] = h.pred_first[M]; return; }
  // (an empty change sits where the next op of its actor would: an empty stretch wherever it points)
  h.seg_base[k] = h.chg_base[k];
  h.pseg_base[k] = h.pred_first[h.chg_base[k]];
}

// byte range of change k in column c: [col_off[c][2k], col_off[c][2k + 1]) -- changes are not in slot order, so begin and end are
// taken separately from the offsets of the change's first slot and of the slot behind its last
__global__ __launch_bounds__(BLOCK) void kh_col_ranges(HistBufs h, int c, bool by_pred, bool raw, uint32_t n_vals) {
  const uint32_t k = gtid();
  if (k >= h.NC) return;
  const uint32_t b = by_pred ? h.pseg_base[k] : h.seg_base[k];
  const uint32_t n = by_pred ? h.pred_first[h.chg_base[k] + h.chg_nops[k]] - h.pred_first[h.chg_base[k]] : h.chg_nops[k];
  const uint32_t e = b + n;
  uint32_t *out = h.col_off + (size_t)c * 2 * (h.NC + 1);
  if (raw) { out[2 * k] = h.enc.off_ex[b]; out[2 * k + 1] = h.enc.off_ex[e]; }
  else { out[2 * k] = h.enc.off_ex[h.enc.run_ex[b]]; out[2 * k + 1] = h.enc.off_ex[h.enc.run_ex[e < n_vals ? e : n_vals]]; }
}

void hist_stage2(const OpCols& rows, const uint8_t* arena, size_t arena_len, HistBufs& h, uint32_t n_sorted, uint32_t M, hipStream_t st) {
  (void)hipMemsetAsync(h.abits, 0, 4 * (size_t)(h.NC + 1) * h.AW, st);
  (void)hipMemsetAsync(h.col_len, 0, 4 * HIST_NCOL, st);
  (void)hipMemsetAsync(h.col_off, 0, 4 * (size_t)HIST_NCOL * 2 * (h.NC + 1), st);
  if (M) {
    AM355_LAUNCH_INDEPENDENT(kh_refs, grid_for(M), dim3(BLOCK), st, rows, h, n_sorted, M);
    AM355_LAUNCH_INDEPENDENT(kh_rows, grid_for(M), dim3(BLOCK), st, rows, arena, (unsigned long long)arena_len, h, M);
  }
  AM355_LAUNCH_INDEPENDENT(kh_seg_bases, grid_for(h.NC + 1), dim3(BLOCK), st, h, M);
  if (!M) return;
  // the twelve columns, every one segmented by change; after each encode the byte range of every change in it
  const uint32_t PT = h.P;   // (pred entries in use: pred_first[M]; the arrays behind them are never read: every encode covers [0, n))
  auto ranges = [&](int c, bool by_pred, bool raw, uint32_t n_vals) {
    AM355_LAUNCH_INDEPENDENT(kh_col_ranges, grid_for(h.NC), dim3(BLOCK), st, h, c, by_pred, raw, n_vals);
  };
  (void)PT;
  enc_rle_numbers(h.v_obj_actor, nullptr, M, false, h.enc, h.col_out[0], h.col_len + 0, st, h.seg); ranges(0, false, false, M);
  enc_rle_numbers(h.v_obj_ctr, nullptr, M, false, h.enc, h.col_out[1], h.col_len + 1, st, h.seg); ranges(1, false, false, M);
  enc_rle_numbers(h.v_key_actor, nullptr, M, false, h.enc, h.col_out[2], h.col_len + 2, st, h.seg); ranges(2, false, false, M);
  enc_delta_prepare(h.v_key_ctr, M, h.deltas, h.nullmask, h.enc, st, h.seg);
  enc_rle_numbers(h.deltas, h.nullmask, M, true, h.enc, h.col_out[3], h.col_len + 3, st, h.seg); ranges(3, false, false, M);
  enc_rle_strings(arena, h.v_key_off, h.v_key_len, M, h.enc, h.col_out[4], h.col_len + 4, st, h.seg, (uint32_t)std::min<size_t>(h.col_cap[4], 0xffffffffu)); ranges(4, false, false, M);
  enc_boolean(h.v_insert, M, h.enc, h.col_out[5], h.col_len + 5, st, h.seg); ranges(5, false, false, M);
  enc_rle_numbers(h.v_action, nullptr, M, false, h.enc, h.col_out[6], h.col_len + 6, st, h.seg); ranges(6, false, false, M);
  enc_rle_numbers(h.v_val_tl, nullptr, M, false, h.enc, h.col_out[7], h.col_len + 7, st, h.seg); ranges(7, false, false, M);
  enc_raw_values(arena, h.v_val_off, h.v_val_tl, M, h.enc, h.col_out[8], h.col_len + 8, st); ranges(8, false, true, M);
  enc_rle_numbers(h.v_pred_num, nullptr, M, false, h.enc, h.col_out[9], h.col_len + 9, st, h.seg); ranges(9, false, false, M);
  // (the pred columns run over the pred entries: their count sits in flags[3], the host passes it as h.P for this stage)
  if (h.P) {
    enc_rle_numbers(h.p_actor, nullptr, h.P, false, h.enc, h.col_out[10], h.col_len + 10, st, h.pseg); ranges(10, true, false, h.P);
    enc_delta_prepare(h.p_ctr, h.P, h.deltas, h.nullmask, h.enc, st, h.pseg);
    enc_rle_numbers(h.deltas, h.nullmask, h.P, true, h.enc, h.col_out[11], h.col_len + 11, st, h.pseg); ranges(11, true, false, h.P);
  }
}

}  // namespace am355
