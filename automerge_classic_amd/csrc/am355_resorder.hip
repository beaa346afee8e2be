// Resident list order: Backend.applyChanges onto a state the context holds, small batches (am355_replay.hip replay_resident).
//
// Reference: a new list element is placed by seekToOp / seekWithinBlock (backend/new.js:227-317, 50-192): from the op behind its
// reference element (or the head of the list) skip every following element with a GREATER opId and insert in front of the first
// one with a smaller opId (the RGA rule, :144-163). A child's id is greater than its parent's, so the skipped stretch is exactly the
// subtrees of the greater siblings. For a batch merged into an EXISTING order that gives, per new element x:
//   * reference element old (or the head): gap(x) = the first OLD position q behind it with id(order[q]) < id(x) (the end of the
//     object when there is none) -- one forward scan over the stored order, 64 positions per step;
//   * two such elements with the same gap stand in DESCENDING id order (the greater one is skipped by the smaller one's scan), each
//     followed by the new elements that hang below it (elements of two OBJECTS can share a gap -- the end of one object is the first
//     position of the next: the object in front first);
//   * reference element new: x follows it directly when it is its only new child (a typing run: the shape of nearly every batch).
//     A new element with two new children is left to the full ordering (flag), as is everything that is not a plain list edit.
// Plain map rows among the batch's rows (`set` / `del` on string keys) take no part in any of this: they stand like deletions here, and
// the caller runs the map half of the merge behind the list merge (words[3] / words[4]; a batch of map rows only is left to that half
// alone, replay_resident / merge_run_maps).
// The order after the batch is then a MERGE: an old element at position p moves up by the number of new elements with gap <= p, the
// k-th new element (by gap, root id descending, depth in its run) lands at gap + k.
#include "am355_resorder.h"
#include "am355_rows.h"
#include "am355_canary.h"

namespace am355 {

static size_t al256(size_t b) { return carve_round(b); }
constexpr uint16_t RO_NOT_ELEM = 0xfffe, RO_ROOT = 0xffff;   // r.par: not a new list element; a new element whose reference element is old (or a list head)

size_t resorder_bytes(uint32_t n_new, uint32_t n_obj) {
  return 3 * al256(4 * ((size_t)n_new + 1)) + al256(2 * ((size_t)n_new + 1)) + al256(4 * ((size_t)n_obj + 2)) + al256(64) + 256;
}

void resorder_bind(ResOrderBufs& r, void* block, uint32_t n_new, uint32_t n_obj) {
  canary_scope("resident list order (resorder_bind)");
  canary_forget(block, resorder_bytes(n_new, n_obj));
  uint8_t* p = (uint8_t*)block;
  auto take = [&](size_t bytes) { void* q = p; canary_note(p, bytes); p += al256(bytes); return q; };
  r.gap = (uint32_t*)take(4 * ((size_t)n_new + 1));
  r.srt_gap = (uint32_t*)take(4 * ((size_t)n_new + 1));
  r.srt_row = (uint32_t*)take(4 * ((size_t)n_new + 1));
  r.par = (uint16_t*)take(2 * ((size_t)n_new + 1));
  r.obj_add = (uint32_t*)take(4 * ((size_t)n_obj + 2));
  r.words = (uint32_t*)take(64);
  canary_allow(r.obj_add, (size_t)((uint8_t*)(r.words + 8) - (uint8_t*)r.obj_add));   // (cleared by one fill, replay_resident)
}

__global__ __launch_bounds__(BLOCK) void kr_positions(MergeBufs b, uint32_t n_list, uint32_t* __restrict__ pos_of) {
  uint32_t p = gtid();
  if (p < n_list) pos_of[b.order[p]] = p;
}

void resorder_positions(const MergeBufs& b, uint32_t n_list, uint32_t* pos_of, hipStream_t st) {
  if (n_list) AM355_LAUNCH_INDEPENDENT(kr_positions, dim3((n_list + BLOCK - 1) / BLOCK), dim3(BLOCK), st, b, n_list, pos_of);
}

// one wavefront per new row: is the row one this path serves; the gap of a new element whose reference element is old
constexpr uint32_t GAP_STEPS_MAX = 4096;   // 64 positions each: a scan past 262 k greater elements is left to the full ordering
__global__ __launch_bounds__(WAVE) void kr_gaps(MergeBufs b, ResOrderBufs r) {
  const uint32_t t = blockIdx.x, lane = threadIdx.x;
  if (t >= r.n_new) return;
  const uint32_t g = r.T0 + t;
  const OpCols& o = b.ops;
  const uint8_t kind = b.kind[g];
  const uint32_t a = o.action[g];
  // words[3]: some row of the batch is not a plain map row (`set` of a value / `del` on a string key). A batch of such rows only leaves
  // every list as it is: the caller then runs the map half of the merge alone (merge_run_maps). Every chunk's rows are looked at.
  // words[4]: some row IS a plain map row: it takes no part in the list order (a row like a deletion here), and the caller runs the map
  // half of the merge behind the in-place list merge.
  const bool plain_map = (kind == K_MAP && a == 1) || (kind == K_DEL && o.key_len[g] != NONE32);
  if (lane == 0) r.words[plain_map ? 4 : 3] = 1;
  // (a chunk behind one that was refused: the order it would scan was never written)
  if (r.chunk && r.words[0]) return;
  uint32_t gap = NONE32;
  bool refuse = false;
  const bool list_del = kind == K_DEL && o.key_len[g] == NONE32;
  if (plain_map && r.allow_maps) {}
  else if (!(kind == K_LIST_INS || kind == K_LIST_UPD || list_del)) refuse = true;   // other map rows (objects made, increments), foreign rows, rows k_resolve left without a kind
  else if (kind != K_DEL && a != 1) refuse = true;                               // child objects (the object table grows), increments, links
  else if (kind == K_LIST_INS) {
    const uint32_t parent = b.ref_row[g];
    const bool head = o.key_ctr[g] == 0;
    const uint32_t make_row = b.obj_row[g];
    if (!head && parent == NONE32) refuse = true;
    else if (make_row != NONE32 && make_row >= r.T0) refuse = true;   // (an object this batch makes: it has no index in the object table yet -- its make row refuses the batch anyway)
    else if (head || parent < r.T0) {
      const uint32_t oi = obj_index_of(b, make_row);
      const uint32_t first = b.obj_first_pos[oi], end = first + b.obj_n[oi];
      uint32_t q = head ? first : r.pos_of[parent] + 1;
      const unsigned long long my_id = pack_id(o.id_ctr[g], o.id_actor[g]);
      gap = end;
      if (q < first || q > end) refuse = true;   // (the stored positions do not describe this object: never expected)
      for (uint32_t step = 0; !refuse && q < end; step++, q += WAVE) {
        if (step >= GAP_STEPS_MAX) { refuse = true; break; }
        bool smaller = false;
        if (q + lane < end) {
          const uint32_t e = b.order[q + lane];
          smaller = pack_id(o.id_ctr[e], o.id_actor[e]) < my_id;
        }
        const unsigned long long m = __ballot(smaller);
        if (m) { gap = q + (uint32_t)__ffsll(m) - 1; break; }
      }
    }
  }
  if (lane == 0) {
    r.gap[t] = gap;
    // (kr_order: the reference element as an index into the batch)
    uint16_t par = RO_NOT_ELEM;
    if (kind == K_LIST_INS) {
      const uint32_t parent = b.ref_row[g];
      par = (o.key_ctr[g] != 0 && parent != NONE32 && parent >= r.T0) ? (uint16_t)(parent - r.T0) : RO_ROOT;
    }
    r.par[t] = par;
    if (refuse) r.words[0] = 1;
  }
}

// one workgroup: the new elements in their final order
constexpr uint32_t RO_THREADS = 1024;
static_assert(RESORDER_ROOTS_MAX <= RO_THREADS, "kr_order scans the root sizes one root per thread");
constexpr uint32_t RO_PER = (RESORDER_ROWS_MAX + RO_THREADS - 1) / RO_THREADS;   // rows per thread
// (LDS: 12288 rows x (root | depth 4 B + new child 2 B + slot 2 B) + the roots' tables = 130 KB of the CU's 160: a batch of 40 changes of 250
//  ops is ONE chunk. Round 6 first held two copies of root / depth as 32-bit words for 4096 rows; a 10 k-row batch then took three
//  chunks, each with its own pass over the whole stored order.
//  What one workgroup must do is the pointer jumping; everything per ROW that needs device memory is done by kr_gaps' wavefronts (the
//  parent of every row as a 16-bit word, r.par) and everything else is per RUN: its length is its last element's depth + 1, which
//  sizes the root's stretch and counts the object's new elements -- no atomics per row, no gathers from device memory per row.)
__global__ __launch_bounds__(RO_THREADS) void kr_order(MergeBufs b, ResOrderBufs r) {
  static_assert(RESORDER_ROWS_MAX < RO_NOT_ELEM && RESORDER_ROWS_MAX < 0x8000, "batch indexes are 16-bit words here, depths 15 bits");
  constexpr uint32_t NOT_ELEM_BIT = 1u << 31;
  __shared__ uint32_t s_rd[RESORDER_ROWS_MAX];       // root so far | depth below it << 16 | NOT_ELEM_BIT: not a new list element
  __shared__ uint16_t s_kid[RESORDER_ROWS_MAX];      // the new element that refers to this one (0xffff: none)
  __shared__ uint16_t s_slot[RESORDER_ROWS_MAX];     // of a root row: its slot in s_roots
  __shared__ uint32_t s_roots[RESORDER_ROOTS_MAX], s_rank_of_root[RESORDER_ROOTS_MAX], s_size[RESORDER_ROOTS_MAX], s_base[RESORDER_ROOTS_MAX + 1];
  __shared__ unsigned long long s_rid[RESORDER_ROOTS_MAX];
  __shared__ uint32_t s_rgap[RESORDER_ROOTS_MAX], s_roi[RESORDER_ROOTS_MAX];
  __shared__ uint32_t s_n_roots, s_bad, s_moved[2], s_wave_tot[RO_THREADS / WAVE];
  const uint32_t t0 = threadIdx.x, n = r.n_new;
  const OpCols& o = b.ops;
  if (t0 == 0) {
    s_n_roots = 0; s_bad = 0; s_moved[0] = 0;
    if (r.chunk) { r.words[2] += r.words[1]; r.words[1] = 0; }   // (the elements of the chunk in front are part of the order now)
    else r.words[2] = r.n_list;
  }
  for (uint32_t t = t0; t < n; t += RO_THREADS) s_kid[t] = 0xffff;
  __syncthreads();
  // (no row so far that is not a plain map row: nothing to merge -- left to the caller's map path, which runs no list kernel at all)
  if (r.words[0] || n > RESORDER_ROWS_MAX || r.words[3] == 0) { if (t0 == 0) r.words[0] = 1; return; }
  // ---- parents within the batch (kr_gaps), roots. A new element with two new children is not a run: both write their index at the
  //      parent and one of them does not find it there (plain stores -- atomics on a bit per row were 32 lanes on one word) ----
  uint16_t par[RO_PER];
#pragma unroll
  for (uint32_t j = 0; j < RO_PER; j++) {
    const uint32_t t = t0 + j * RO_THREADS;
    par[j] = t < n ? r.par[t] : RO_NOT_ELEM;
  }
  // (a typing run is a stretch of consecutive rows, each the child of the row in front: inside a wavefront -- 64 consecutive rows -- such a
  //  stretch is contracted at once, every row pointing at the reference element of the stretch's first row; the jumping below then takes
  //  log2 of the wavefronts a run spans, not of its length)
  const uint32_t lane = t0 & (WAVE - 1);
#pragma unroll
  for (uint32_t j = 0; j < RO_PER; j++) {
    const uint32_t t = t0 + j * RO_THREADS;
    const bool active = t < n;
    const uint16_t p = par[j];
    if (active) {
      if (p < RO_NOT_ELEM) {
        if (p >= n) s_bad = 1; else s_kid[p] = (uint16_t)t;
      } else if (p == RO_ROOT) {
        const uint32_t k = atomicAdd(&s_n_roots, 1u);
        if (k < RESORDER_ROOTS_MAX) s_roots[k] = t; else s_bad = 1;
      }
    }
    const bool follows = active && lane > 0 && (uint32_t)p + 1u == t;
    const unsigned long long heads = __ballot(!follows);
    const uint32_t hl = 63u - (uint32_t)__clzll(heads & (~0ull >> (63u - lane))), dl = lane - hl;   // (lane 0 is a head)
    const uint32_t hp = (uint32_t)__shfl((int)p, (int)hl);   // the reference element of the stretch's first row
    if (active) {
      uint32_t rd;
      if (p >= RO_NOT_ELEM || p >= n) rd = p == RO_ROOT ? t : t | NOT_ELEM_BIT;
      else if (hp < RO_NOT_ELEM) rd = hp | (dl + 1u) << 16;
      else { rd = (t - dl) | dl << 16; if (hp != RO_ROOT) s_bad = 1; }   // (the stretch hangs below a root of this wavefront)
      s_rd[t] = rd;
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < RO_PER; j++) {
    const uint32_t t = t0 + j * RO_THREADS;
    if (t < n && par[j] < RO_NOT_ELEM && par[j] < n && s_kid[par[j]] != t) s_bad = 1;
  }
  __syncthreads();
  if (s_bad) { if (t0 == 0) r.words[0] = 1; return; }
  // ---- root and depth of every new element: pointer jumping over the runs, in place (a step reads into registers, then writes); it ends
  //      with the first step that moves nothing -- log2 of the LONGEST run steps, not of the batch ----
  for (uint32_t span = 1, round = 0; span < n; span <<= 1, round++) {
    uint32_t nw[RO_PER];
    bool moved = false;
#pragma unroll
    for (uint32_t j = 0; j < RO_PER; j++) {
      const uint32_t t = t0 + j * RO_THREADS;
      if (t < n) {
        const uint32_t rd = s_rd[t], up = rd & 0xffffu, ru = s_rd[up];
        nw[j] = (ru & 0xffffu) | ((rd >> 16) + (up != t ? ru >> 16 : 0u)) << 16;
        moved |= (ru & 0xffffu) != up;
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < RO_PER; j++) {
      const uint32_t t = t0 + j * RO_THREADS;
      if (t < n) s_rd[t] = nw[j];
    }
    if (moved) s_moved[round & 1u] = 1;
    if (t0 == 0) s_moved[(round & 1u) ^ 1u] = 0;   // (the next step's word: every thread has read it a barrier ago)
    __syncthreads();
    if (!s_moved[round & 1u]) break;
  }
  // ---- roots by (gap, id descending): rank by counting (a batch of a few changes has a few roots) ----
  const uint32_t R = s_n_roots;
  for (uint32_t k = t0; k < R; k += RO_THREADS) {
    const uint32_t t = s_roots[k], g = r.T0 + t;
    s_rgap[k] = r.gap[t];
    s_roi[k] = obj_index_of(b, b.obj_row[g]);
    s_rid[k] = pack_id(o.id_ctr[g], o.id_actor[g]);
    s_slot[t] = (uint16_t)k;   // root index of a batch row -> its slot k in s_roots
  }
  __syncthreads();
  for (uint32_t k = t0; k < R; k += RO_THREADS) {
    const uint32_t gap_k = s_rgap[k], oi_k = s_roi[k];
    const unsigned long long id_k = s_rid[k];
    uint32_t rank = 0;
    // (one gap can belong to two objects: the end of one is the first position of the next -- the object in front first; objects lie
    //  in the order of their indexes, kr_apply's object table counts on the same)
    for (uint32_t j = 0; j < R; j++)
      rank += (s_rgap[j] < gap_k || (s_rgap[j] == gap_k && (s_roi[j] < oi_k || (s_roi[j] == oi_k && s_rid[j] > id_k)))) ? 1u : 0u;   // (ids are unique: j == k counts nothing)
    s_rank_of_root[k] = rank;
  }
  __syncthreads();
  // the length of a run: the depth of its last element (the one no new element refers to) + 1; sizes in RANK order
  for (uint32_t t = t0; t < n; t += RO_THREADS) {
    const uint32_t rd = s_rd[t];
    if (!(rd & NOT_ELEM_BIT) && s_kid[t] == 0xffff) s_size[s_rank_of_root[s_slot[rd & 0xffffu]]] = (rd >> 16) + 1u;
  }
  __syncthreads();
  {
    // s_base = exclusive prefix of the sizes (R <= RESORDER_ROOTS_MAX = RO_THREADS: one root per thread; a single thread walking a
    // thousand dependent LDS reads was 30 us of this kernel)
    const uint32_t lane = t0 & (WAVE - 1), wv = t0 / WAVE;
    const uint32_t v = t0 < R ? s_size[t0] : 0u;
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == WAVE - 1) s_wave_tot[wv] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t k = 0; k < RO_THREADS / WAVE; k++) { const uint32_t x = s_wave_tot[k]; before += k < wv ? x : 0u; all += x; }
    if (t0 < R) s_base[t0] = before + incl - v;
    if (t0 == 0) { s_base[R] = all; r.words[1] = all; }
  }
  __syncthreads();
  for (uint32_t t = t0; t < n; t += RO_THREADS) {
    const uint32_t rd = s_rd[t], k = s_slot[rd & 0xffffu];
    if (rd & NOT_ELEM_BIT) continue;
    const uint32_t at = s_base[s_rank_of_root[k]] + (rd >> 16);
    r.srt_gap[at] = s_rgap[k];
    r.srt_row[at] = r.T0 + t;
  }
  // new elements per object, run by run: a batch names few objects -- usually ONE --, and device-scope atomics on one word execute one
  // after another (~12 ns each). The lanes of a wavefront (a root each) whose runs lie in one object share one.
  {
    const uint32_t lane = t0 & (WAVE - 1);
    const bool mine = t0 < R;
    const uint32_t oi = mine ? s_roi[t0] : NONE32, sz = mine ? s_size[s_rank_of_root[t0]] : 0u;
    unsigned long long left = __ballot(mine);
    while (left) {
      const uint32_t leader = (uint32_t)__ffsll(left) - 1;
      const uint32_t loi = __shfl(oi, (int)leader);
      const bool same = mine && oi == loi;
      const uint32_t sum = __shfl(wave_incl_scan_u32(same ? sz : 0u, lane), WAVE - 1);
      if (lane == leader) atomicAdd(&r.obj_add[loi], sum);
      left &= ~__ballot(same);
    }
  }
}

// ---- what follows from kr_order's result, in ONE launch (three launches and a signalling one before: the host's launch rate, ~4 us per
// launch, is what a call of 0.3 ms is made of). Workgroups [0, shift_blocks) write the merged order, workgroup shift_blocks the object
// table, the rest the visibility verdicts; none reads what another writes. The verdict words the host waits for are final since
// kr_order: the first thread sends them on its way in (r.sig), so the host enqueues the next stage while this one runs. ----

// the merged order: old elements move up by the new elements in front of them, the k-th new element lands at gap + k
__device__ __forceinline__ void shift_item(const MergeBufs& b, const ResOrderBufs& r, uint32_t i) {
  const uint32_t K = r.words[1], n_list = r.words[2];
  if (i < n_list) {
    uint32_t lo = 0, hi = K;   // new elements with gap <= i
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (r.srt_gap[mid] <= i) lo = mid + 1; else hi = mid; }
    const uint32_t e = b.order[i];
    r.order_new[i + lo] = e;
    r.pos_of[e] = i + lo;
  }
  if (i < K) {
    const uint32_t at = r.srt_gap[i] + i, row = r.srt_row[i];
    r.order_new[at] = row;
    r.pos_of[row] = at;
  }
}

// per object: its new elements; first positions move up by the new elements of the objects in front (one workgroup; objects are few)
__device__ __forceinline__ void objects_block(const MergeBufs& b, const ResOrderBufs& r) {
  __shared__ uint32_t s_red[BLOCK / WAVE];
  uint32_t carry = 0;
  for (uint32_t base = 0; base <= r.n_obj; base += BLOCK) {
    const uint32_t oi = base + threadIdx.x;
    const uint32_t add = oi <= r.n_obj ? r.obj_add[oi] : 0u;
    if (oi <= r.n_obj && add) r.obj_add[oi] = 0;   // (the next chunk counts anew)
    uint32_t total;
    const uint32_t ex = block_exclusive_scan_u32(add, s_red, &total);
    if (oi <= r.n_obj) {
      b.obj_first_pos[oi] += carry + ex;
      b.obj_n[oi] += add;
    }
    carry += total;
  }
}

// the verdict "its own value is visible" (k_emit) of the batch's elements and of the elements its rows overwrite or delete
__device__ __forceinline__ void kinds_item(const MergeBufs& b, const ResOrderBufs& r, uint32_t t) {
  if (t >= r.n_new) return;
  const uint32_t g = r.T0 + t;
  const OpCols& o = b.ops;
  const uint8_t kind = b.kind[g];
  if (kind == K_LIST_INS) { if (b.succ_cnt[g] == 0) b.kind[g] = K_LIST_INS_VIS; return; }   // (valued: kr_gaps admitted `set` rows only)
  for (uint32_t k = 0; k < o.pred_num[g]; k++) {
    const uint32_t pr = row_of(b, o.pred_actor[o.pred_first[g] + k], o.pred_ctr[o.pred_first[g] + k]);
    if (pr != NONE32 && b.kind[pr] == K_LIST_INS_VIS) b.kind[pr] = K_LIST_INS;   // (it has a successor now: this row)
  }
}

__global__ __launch_bounds__(BLOCK) void kr_apply(MergeBufs b, ResOrderBufs r, uint32_t shift_blocks) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && r.sig) {
    uint32_t v[9];
    for (int k = 0; k < 8; k++) v[k] = r.words[k];
    v[8] = b.counts->flags;
    signal_host(r.sig->resorder, v, 9, &r.sig->resorder_seq, r.sig_seq);
  }
  if (r.words[0]) return;
  if (blockIdx.x < shift_blocks) shift_item(b, r, gtid());
  else if (blockIdx.x == shift_blocks) objects_block(b, r);
  else kinds_item(b, r, (blockIdx.x - shift_blocks - 1) * BLOCK + threadIdx.x);
}

uint32_t resorder_chunk_rows() {
  const char* e = getenv("AM355_RESORDER_CHUNK");
  const long v = e ? atol(e) : 0;
  return v > 0 && v < (long)RESORDER_ROWS_MAX ? (uint32_t)v : RESORDER_ROWS_MAX;
}

void resorder_run(MergeBufs& b, ResOrderBufs& r, hipStream_t st, bool* final_in_new) {
  if (final_in_new) *final_in_new = false;
  if (!r.n_new) return;
  const uint32_t rows = resorder_chunk_rows();
  const uint32_t T0 = r.T0, n_all = r.n_new, chunks = (n_all + rows - 1) / rows;
  HostSignals* sig = r.sig;
  uint32_t* src = b.order;
  uint32_t* dst = r.order_new;
  MergeBufs bb = b;
  ResOrderBufs rr = r;
  for (uint32_t c = 0; c < chunks; c++) {
    rr.chunk = c;
    rr.T0 = T0 + c * rows;
    rr.n_new = std::min(rows, n_all - c * rows);
    rr.sig = c + 1 == chunks ? sig : nullptr;   // (the last chunk's kr_apply tells the host)
    bb.order = src;
    rr.order_new = dst;
    hipLaunchKernelGGL(kr_gaps, dim3(rr.n_new), dim3(WAVE), 0, st, bb, rr);
    hipLaunchKernelGGL(kr_order, dim3(1), dim3(RO_THREADS), 0, st, bb, rr);
    const uint32_t n_list_most = r.n_list + c * rows;   // (what the running count can have reached)
    const uint32_t most = n_list_most > rr.n_new ? n_list_most : rr.n_new;
    const uint32_t shift_blocks = (most + BLOCK - 1) / BLOCK, kind_blocks = (rr.n_new + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(kr_apply, dim3(shift_blocks + 1 + kind_blocks), dim3(BLOCK), 0, st, bb, rr, shift_blocks);
    std::swap(src, dst);
  }
  if (final_in_new) *final_in_new = (chunks & 1u) != 0;
}

}  // namespace am355
