// Resident list order (am355_apply_changes onto a state the context holds): the new elements of a small batch are ranked against the
// STORED document order instead of ordering every list from scratch. See am355_resorder.hip.
#pragma once
#include "am355_merge.h"

namespace am355 {

constexpr uint32_t RESORDER_ROWS_MAX = 12288;   // new rows the single ordering workgroup holds in LDS: one CHUNK of a batch
constexpr uint32_t RESORDER_CHUNKS_MAX = 3;     // a larger batch is merged chunk by chunk, each against the order the chunks in front left (resorder_run)
constexpr uint32_t RESORDER_ROOTS_MAX = 1024;   // new elements whose reference element is old (or a list head)

struct ResOrderBufs {
  uint32_t T0, n_new;        // the batch's rows: [T0, T0 + n_new) (resorder_run: of the chunk it is launching)
  uint32_t n_list;           // elements in b.order before the call (the kernels read the running count, words[2])
  uint32_t chunk;            // 0 .. chunks - 1 (resorder_run)
  uint32_t allow_maps;       // plain map rows among the batch's rows do not refuse it (the caller runs the map half of the merge behind the list merge)
  uint32_t n_obj;            // objects including _root (obj_n / obj_first_pos hold n_obj + 1 entries)
  uint32_t* pos_of;          // [row capacity] position of every element row in b.order (kept between calls)
  uint32_t* order_new;       // [row capacity + 2] the order after the call (the caller swaps it with b.order)
  uint32_t* gap;             // [n_new] a new element whose reference element is old / a head: the old position it goes in front of
  uint16_t* par;             // [n_new] the reference element of a new element as an index into the batch (kr_gaps -> kr_order)
  uint32_t* srt_gap;         // [n_new] the new elements in their final order: gap ...
  uint32_t* srt_row;         // [n_new] ... and row
  uint32_t* obj_add;         // [n_obj + 1] new elements per object (cleared by the caller)
  uint32_t* words;           // [8] (cleared by the caller): [0] != 0: not served here (the caller orders all lists anew), [1] new
                             // elements of the last chunk, [2] elements in front of them: the order holds [1] + [2] after the call;
                             // [3] != 0: some row of the batch is not a plain map row, [4] != 0: some row is one (kr_gaps)
  HostSignals* sig;          // the words + Counts.flags for the host through pinned memory (HostSignals.resorder), nullptr: the caller copies them
  uint32_t sig_seq;
};

size_t resorder_bytes(uint32_t n_new, uint32_t n_obj);
void resorder_bind(ResOrderBufs& r, void* block, uint32_t n_new, uint32_t n_obj);

// pos_of[b.order[p]] = p for the n_list elements (after a full ordering)
void resorder_positions(const MergeBufs& b, uint32_t n_list, uint32_t* pos_of, hipStream_t st);
// Ranks the batch's new list elements against the stored order (k_resolve of the batch has run): r.words[0] tells whether the batch is
// one this path serves -- list rows only (inserts, deletions, assignments of plain values), no new object, at most one new child per
// new element (typing runs), <= RESORDER_ROWS_MAX rows, <= RESORDER_ROOTS_MAX roots --; if so r.order_new / r.pos_of / b.obj_n /
// b.obj_first_pos / b.kind describe the state after the batch. The caller reads r.words back (8 words; r.sig: signalled) before it relies on them.
// A batch of more than RESORDER_ROWS_MAX rows goes chunk by chunk (rows are in application order: a chunk refers to nothing behind it),
// the order ping-ponging between b.order and r.order_new: *final_in_new tells where it ends up. A refusal in a later chunk leaves the
// earlier chunks merged -- the full ordering the caller then runs starts from the rows, not from these arrays.
void resorder_run(MergeBufs& b, ResOrderBufs& r, hipStream_t st, bool* final_in_new = nullptr);
uint32_t resorder_chunk_rows();   // RESORDER_ROWS_MAX, or less: AM355_RESORDER_CHUNK (tests: small batches in several chunks), read per call

}  // namespace am355
