// Device helpers over the op rows and the merge results that more than one stage uses (am355_merge.hip, am355_delta.hip).
#pragma once
#include "am355_merge.h"

namespace am355 {

// K_LIST_INS_VIS: an insert row whose own value is visible (set by k_emit: the element's value count is val_cnt + this bit, so the
// common case -- one visible value per element, its insert -- costs no atomic)
// K_FOREIGN: objectId sharding (MergeBufs.shard_world > 1) -- the row belongs to an object another rank owns. It takes no part in
// this rank's merge (no validation, no succ counting, no emission) except that a make row still enters the object table, so
// that object indexes are the same on every rank (the fragments of the patch IR are stitched by object index).
enum Kind : uint8_t { K_NONE = 0, K_MAP = 1, K_LIST_INS = 2, K_LIST_UPD = 3, K_DEL = 4, K_LIST_INS_VIS = 5, K_FOREIGN = 6 };


__device__ __forceinline__ uint32_t row_of(const MergeBufs& b, uint32_t actor, uint32_t ctr) {
  if (actor >= b.n_actors) return NONE32;
  uint32_t first = b.actor_tab_off[actor], lo = first, hi = b.actor_tab_off[actor + 1];
  while (lo < hi) {  // first span with start_op > ctr
    uint32_t mid = (lo + hi) >> 1;
    if (b.spans[mid].start_op <= ctr) lo = mid + 1; else hi = mid;
  }
  if (lo == first) return NONE32;
  ActorSpan s = b.spans[lo - 1];
  uint32_t d = ctr - s.start_op;
  return d < s.n_ops ? s.op_base + d : NONE32;
}

__device__ __forceinline__ unsigned long long pack_id(uint32_t ctr, uint32_t actor) { return (unsigned long long)ctr << 32 | actor; }

// element a non-insert list row refers to / an insert row creates
__device__ __forceinline__ uint32_t elem_of(const MergeBufs& b, uint32_t row) {
  if (b.ops.insert[row]) return row;
  if (b.ops.key_ctr[row] == NONE32 || b.ops.key_ctr[row] == 0) return NONE32;
  return row_of(b, b.ops.key_actor[row], b.ops.key_ctr[row]);
}

__device__ __forceinline__ bool same_obj(const MergeBufs& b, uint32_t r1, uint32_t r2) {
  return b.ops.obj_actor[r1] == b.ops.obj_actor[r2] && b.ops.obj_ctr[r1] == b.ops.obj_ctr[r2];
}

__device__ __forceinline__ bool same_key(const MergeBufs& b, uint32_t r1, uint32_t r2) {
  uint32_t l1 = b.ops.key_len[r1], l2 = b.ops.key_len[r2];
  if (l1 == NONE32 || l1 != l2) return false;
  const uint8_t *p = b.arena + b.ops.key_off[r1], *q = b.arena + b.ops.key_off[r2];
  for (uint32_t k = 0; k < l1; k++)
    if (p[k] != q[k]) return false;
  return true;
}

__device__ __forceinline__ uint32_t obj_index_of(const MergeBufs& b, uint32_t make_row) { return make_row == NONE32 ? 0 : b.obj_index[make_row]; }

// JS compares strings by UTF-16 code units (new.js:84). On valid UTF-8 that equals byte order except that
// supplementary-plane characters (lead bytes F0..F4) sort below U+E000..U+FFFF (lead bytes EE, EF): remap.
__device__ __forceinline__ uint32_t utf16_order_byte(uint32_t x) {
  if (x >= 0xf0 && x <= 0xf4) return x - 2;
  if (x == 0xee || x == 0xef) return x + 5;
  return x;
}

// value class compared by appendEdit's `datatype` and `typeof` tests (new.js:759-760, 768-769)
__device__ __forceinline__ uint32_t value_class(uint32_t tl) {
  uint32_t tag = tl & 15;
  if (tl == 0) return 0;
  if (tl == 1 || tl == 2) return 1;
  if (tag == 6 || tag == 3 || tag == 4 || tag == 5 || tag == 8 || tag == 9) return tag;
  return 16 + tag;
}

// sLEB / uLEB value of an integer-typed op value (tags 3 uint, 4 int, 8 counter, 9 timestamp); columnar.js:300-329
__device__ __forceinline__ bool int_value(const MergeBufs& b, uint32_t row, long long& out) {
  uint32_t tl = b.ops.val_tl[row], tag = tl & 15, len = tl >> 4;
  if (!(tag == 3 || tag == 4 || tag == 8 || tag == 9) || len == 0 || len > 10) return false;
  const uint8_t* p = b.arena + b.ops.val_off[row];
  unsigned long long v = 0;
  int shift = 0;
  for (uint32_t k = 0; k < len; k++) {
    uint32_t byte = p[k];
    v |= (unsigned long long)(byte & 0x7f) << shift;
    shift += 7;
    if (!(byte & 0x80)) {
      if (tag != 3 && (byte & 0x40) && shift < 64) v |= ~0ull << shift;
      out = (long long)v;
      return true;
    }
  }
  return false;
}

}  // namespace am355
