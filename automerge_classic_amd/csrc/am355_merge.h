// Stage 2 of the replay engine: op-set merge and whole-document patch IR (see am355_merge.hip).
#pragma once
#include "am355_internal.h"
#include "am355_scan.h"
#include "am355_prims.h"
#include "../../include/am355.h"
#include <stddef.h>

namespace am355 {

// counters produced on the device and read back once by the host to size the later launches
struct Counts {
  uint32_t flags;        // OR of Flag bits raised by any kernel
  uint32_t n_map_emit;   // visible map/table values
  uint32_t n_list_ins;   // insert rows in list/text objects (RGA tree nodes)
  uint32_t n_list_upd;   // visible non-insert values on list elements
  uint32_t n_objects;    // make* rows (+1 for _root)
  uint32_t max_key_len;  // longest map key among emitted values
  uint32_t n_edits;      // list edit records
  uint32_t pad;          // a list element has more children than the in-place sibling ordering handles: redo with the radix sort
  uint32_t n_runs;       // typing runs of the insertion forest (list ranking works on 2 x runs + 1 tour entries)
  uint32_t euler_done;   // the single-workgroup LDS list ranking handled the tour
  uint32_t n_erecs;      // edit records (a multi-insert run counts once)
  uint32_t n_head_children;  // insert rows whose reference element is _head (all list objects)
  uint32_t n_quirk;      // list rows that need the reference's counter / `remove` rules (new.js:937-965, 1010-1033): counters completed by
                         // increments, visible rows without a value. 0 = the ordinary edits (merge_run then skips k_quirk_rows)
  uint32_t n_list_inc;   // increments on list elements seen by k_resolve: only then does k_emit look for counters among the invisible `set` rows of lists
  uint32_t map_group_big; // k_map_group_rank met more than MAP_GROUP_MAX values on one key: the host orders the emissions again, by radix passes over the trigger ids
  uint32_t reserved[1];
};

// What the keys of the map emissions have in common (k_emit; behind Counts in its device block, cleared with it, signalled with it):
// bit i of or_b / or_inv_b word w = some key has / lacks that bit in byte 4 w + (3 - i / 8) ... -- stored as the big-endian words the
// sort keys are made of (map_key_of): byte j of a key (zero beyond its length, UTF-16 order remap applied) sits in word j / 4 at bits
// [24 - 8 (j % 4), 32 - 8 (j % 4)). A byte position is the same in every key iff (or & or_inv) is zero there: its radix pass is the
// identity and is not run; likewise the pass over the key lengths when every key has the same length.
struct MapKeyStats {
  uint32_t or_len, or_inv_len;
  uint32_t or_b[4], or_inv_b[4];   // the first 16 key bytes
};
constexpr uint32_t MAP_KEY_STATS_WORD = 16;  // word offset behind Counts (sizeof(Counts) / 4)

// Device buffers of the merge stage. N = op rows, P = preds. Everything is uint32 unless noted.
struct MergeBufs {
  // inputs
  const uint8_t* arena;
  OpCols ops;
  uint32_t n_ops, n_preds, n_actors;
  uint32_t shard_rank, shard_world;  // objectId sharding: this rank merges the objects it owns (shard_owner); world 1 = everything
  // Resident state (am355_apply_changes onto a state the context holds, am355_replay.hip replay_resident): rows [0, first_row) were
  // decoded and resolved by an earlier replay and are where they were -- with their obj_row / ref_row / kind and with the successor /
  // increment accumulators the rows so far left in them; k_resolve then runs over the rows >= first_row only. seed_list_inc: what
  // Counts.n_list_inc held after that earlier replay (the counter block is cleared per replay). row_stride: words between the per-row
  // arrays of this block (capacity + 1; 0 = n_ops + 1): the fills go by n_ops, not by the stride.
  uint32_t first_row, seed_list_inc, row_stride;
  const uint32_t* actor_tab_off;  // [n_actors + 1] into spans
  const ActorSpan* spans;
  uint32_t bits_ctr, bits_actor;  // key widths: bits(max op counter), bits(n_actors)
  // per-row results
  uint32_t *obj_row, *ref_row, *succ_cnt, *inc_cnt, *val_cnt, *obj_index;
  unsigned long long *inc_sum, *last_inc;
  uint8_t* kind;
  // compaction targets
  uint32_t *em_row;                 // [N] map emissions (rows), later sorted
  unsigned long long* em_trig;      // [N] trigger op id of each map emission (packed ctr<<32|actor)
  uint32_t *ins_row;                // [N] list insert rows
  uint32_t *upd_row;                // [N] visible list update rows
  // sort scratch (sized for N pairs)
  uint64_t *key_a, *key_b;
  uint32_t *val_a, *val_b;
  void* sort_ws;
  // RGA
  uint32_t *first_child;            // [2N+1] indexed by parent slot: element row, or N + make row for a list head
  uint32_t *next_sib;               // [N]
  uint32_t *child_head;             // [2N+1] unordered child list per parent slot (atomic push), aliases the Euler scratch
  uint32_t *child_next;             // [N]
  uint32_t *head_child;             // [HEAD_CHILD_MAX] children of list heads (ordered by one workgroup from LDS)
  uint32_t *obj_n;                  // [n_objects + 1] list elements (insert rows) per object
  uint32_t *run_heads, *row_run;    // typing runs: first insert-list index of each run; run of a row (run heads and tails only)
  uint32_t *list_vis, *list_cnt;    // [N] per list position: element visible, number of its edits (scanned into scan_a / scan_b)
  CarryScan cs_ins, cs_make, cs_runs, cs_vis, cs_cnt, cs_erec;  // carried scans (am355_scan.h); their group sums follow `counts`
  unsigned long long *euler_a, *euler_b;        // [2N+2] Euler tour list ranking: (weight-to-end << 32 | successor)
  uint32_t *order;                  // [N] node rows in document order (all list objects chained)
  uint32_t *scan_a, *scan_b;        // [N+1] prefix sums over `order`
  uint32_t *obj_first_pos;          // [n_objects] position in `order` of the first element of each list object
  void* scan_ws;
  Counts* counts;                   // device; followed by the group sums of the carried scans (cleared with it)
  size_t counts_bytes;              // Counts + group sums
  void* zero_base;                  // succ_cnt .. last_inc are contiguous: one memset per replay
  size_t zero_bytes;
  HostSignals* sig;                 // pinned host memory the last kernels of each phase write their counters to (may be null: documents)
  uint32_t sig_seq;                 // sequence number of this replay
  void* fill_base;                  // order | first_child | child_head are contiguous: one 0xff fill per replay
  size_t fill_bytes;
};

// bytes of the device block that holds Counts and the carried scans' group sums for N op rows
size_t merge_counts_bytes(uint32_t n_ops);
// points b.counts / b.cs_*.group_sum into that block (b.cs_*.wg_sum are carved by the caller: carry_words(N) words each)
void merge_bind_counts(MergeBufs& b, void* d_counts_block);

// ---- patch IR in device memory (the output of the hot path) ---------------------------------------------
// The four record tables of include/am355.h (objects, map records, edit records, values) are written by the device in their
// final layout and copied to the host as they are. The per-element arrays e_* are the intermediate the edit records are
// packed from (one entry per visible list value, in document order).
struct PatchIR {
  am355_ir_object* obj;     // [n_objects] index 0 is _root, then make rows in row order (make_row: NONE32 for root)
  am355_ir_map* map;        // [n_map_emit] sorted by (object, key, trigger op)
  am355_ir_edit* edit;      // [n_erecs + 1] one record per edit -- per uniform stretch of a multi-insert run -- + sentinel
  uint32_t *e_row;          // [n_edits] row holding the value (its id is the edit's opId)
  uint32_t *e_elem;         // [n_edits] row of the element (its id is the elemId)
  uint32_t *e_index;        // [n_edits] list index
  uint32_t *e_flags;        // bit0: update (else insert), bit1: continues the multi-insert run of the previous edit, bit2: child
                            // object, bit8 / bit9: first / last edit of its list object, bit10: its value does not follow the
                            // previous one in the arena with the same type/length word (starts a new record of the same run)
};

size_t merge_scratch_pairs(uint32_t n_ops);

// host side of the device -> host signalling (HostSignals): spins until *seq_word == seq
bool wait_host_signal(volatile uint32_t* seq_word, uint32_t seq, hipStream_t st);  // false: no signal (stream drained instead)

// Zero-fills the merge stage needs (succ / counter accumulators, child lists, list order): independent of the decode kernels,
// so the caller issues them on a second stream beside the decode and joins before merge_run.
// (what: the accumulators of the rows k_resolve is about to run over | what the kernels of merge_run from k_emit on fill and count in:
// replay_resident clears the first before its k_resolve and the second only when it does not merge the list order in place)
enum { MERGE_FILL_ROWS = 1, MERGE_FILL_TABLES = 2 };
void merge_prepare(MergeBufs& b, hipStream_t aux, int what = MERGE_FILL_ROWS | MERGE_FILL_TABLES, const FillRanges* extra = nullptr);   // extra: clears of the caller's, in the same launch
// resolve -> emit -> compaction -> object table, map emission order, RGA order (sibling ordering, typing runs, list ranking),
// edits. `b.counts` (cleared by the caller BEFORE the decode kernels, whose validity flags it already holds) is read back
// twice without draining the stream (ev_counts, ev_runs: the host sizes the later launches while the device works through
// the earlier ones) and once at the end (synchronises st). Returns the counters in *h_counts.
void merge_run(MergeBufs& b, PatchIR& ir, Counts* h_counts, hipStream_t st, hipEvent_t ev_counts, hipEvent_t ev_runs, bool resolved = false);
// the map half alone (rows resolved; a batch of plain map rows onto a kept state): see am355_merge.hip
void merge_run_maps(MergeBufs& b, PatchIR& ir, Counts* h_counts, hipStream_t st);
// k_resolve alone over the rows >= b.first_row (replay_resident: the list order of a small batch is then updated in place,
// am355_resorder.hip, or merge_run(..., resolved = true) goes on from k_emit)
void merge_resolve(MergeBufs& b, hipStream_t st);

// Document load: whole-document patch of rows already in canonical order (pred_* arrays = succ lists). `b.counts` must be
// cleared by the caller before the decode kernels run.
void doc_patch(MergeBufs& b, PatchIR& ir, Counts* h_counts, hipStream_t st);

// ---- Backend.save after a replay: op rows in saved-document order (am355_merge.hip, "Backend.save") -------------------
struct SaveBufs {
  uint32_t *map_flag, *map_ex, *upd_flag, *upd_ex, *upd_cnt, *pos_of, *list_off, *final_pos, *src_of, *map_perm;  // [N + 2]
  uint32_t *obj_rank, *rank_obj, *base_by_rank;            // [n_objects + 2]
  uint32_t *map_begin, *map_end, *list_begin, *list_end;   // [n_objects + 1] each, contiguous
  uint64_t *succ_key_a, *succ_key_b;                       // [P + 1]
  uint32_t *succ_val_a, *succ_val_b;                       // [P + 1]
  uint32_t* words;                                         // [8] device: map rows, update rows, longest key
  OpCols out;                                              // canonical rows (actor fields: document actor index; pred_* = succ lists)
};
// counts map rows / list update rows (s.words), compacts them; the caller reads s.words back before phase 2
void save_phase1(MergeBufs& b, SaveBufs& s, hipStream_t st);
// n_obj includes _root; doc_actor[rank] = index in the document's actor table (device)
void save_phase2(MergeBufs& b, PatchIR& ir, SaveBufs& s, const uint32_t* words, uint32_t n_obj, uint32_t n_ins, const uint32_t* doc_actor, hipStream_t st);

}  // namespace am355
