// Incremental patch of Backend.applyChanges: device stage that runs after a replay of (earlier changes + the new
// batch) and derives, from the merged state and from which rows are new, what the reference's sequential patch state machine
// reports for the batch (see am355_delta.hip).
#pragma once
#include "am355_merge.h"

namespace am355 {

// per object (index as in PatchIR.obj): where it hangs in its parent, whether this call touched it, its ranges in the delta tables
struct ObjLink {
  uint32_t parent;                // object index of the parent (NONE32: _root)
  uint32_t flags;                 // OL_* bits
  uint32_t key_off, key_len;      // map / table parent: key of the make op (arena bytes)
  uint32_t elem_ctr, elem_actor;  // list / text parent: id of the element that holds the object
  uint32_t touch;                 // first new row (counted from the first new row) that names this object, NONE32: untouched
  uint32_t map_begin, map_end;    // its records in the delta map table
  uint32_t edit_begin, edit_end;  // its records in the delta edit table
  uint32_t pad;
};
enum : uint32_t {
  OL_LIST_PARENT = 1u,  // the parent is a list / text object
  OL_VISIBLE = 2u,      // the make op has no successor (the object is a visible value of its parent's property)
  OL_ELEM_NEW = 4u      // list parent: the element was inserted by this call
};

struct DeltaCounts {
  uint32_t flags;      // Flag bits (F_UNSUPPORTED: outside the subset served here)
  uint32_t n_items;    // list edit items (inserts of new elements + events: an element removed, assigned to, or brought back)
  uint32_t n_kept;     // visible values of touched map keys
  uint32_t n_place;    // touched map keys left without a visible value
  uint32_t n_erecs;    // delta edit records
  uint32_t n_slots;    // touched map keys
  uint32_t hazard;     // a key that holds a visible child object lost values to the merge loop's skipping rule (see kd_placeholders)
  uint32_t reason;     // DR_*: why the call is refused (the smallest code raised), NONE32: not refused by this stage
  uint32_t rec_extra;  // edit records beyond one per item: an update / re-insert item writes one record per visible value (kd_events)
  uint32_t deferred;   // kd_edit_small left the second half of the stage to the host (more items / records than one workgroup takes, map records)
};
enum : uint32_t {
  DR_FOREIGN_ROW = 1,     // a row of an object another shard owns
  DR_KEY_TABLE,           // touched-key table full
  DR_ELEM_ROWS,           // more value rows on a touched list element than the stage walks
  DR_ELEM_NOT_PLAIN,      // an assigned list element holds something that is neither a value nor a child object (inc, link)
  DR_SAME_ELEM_CALL,      // two ops on one element in one merge call
  DR_GAP_WALK,            // too many later insertions between two elements of one merge call
  DR_LAGGING_UPDATE,      // the first update edit of a conflict would sit at the reference's lagging index
  DR_POP_WALK,            // too many update edits at one index in a row
  DR_AMBIGUOUS_DEL,       // a deletion whose place in the merge loop's work list is ambiguous
  DR_CHILD_HAZARD,        // values skipped on a key that holds a child object
  DR_EDIT_TABLE,          // more edit records than the edit table holds and no memory to grow it
  DR_INTERNAL
};
const char* delta_reason_text(uint32_t reason);

// Device memory of the delta stage, carved by the caller (delta_bytes / delta_bind). N = op rows, NN = new rows, NM = map records
// of the whole-document IR, NO = objects.
struct DeltaBufs {
  uint32_t T0;            // first new row
  uint32_t n_new;         // N - T0
  uint32_t n_obj;         // objects including _root
  uint32_t n_map;         // records of ir.map
  uint32_t n_list;        // list elements (positions of MergeBufs.order)
  uint32_t bits_new;      // bits of a row number counted from T0
  uint32_t n_pass;        // scheduling passes after the first that begin inside the new rows
  const uint32_t* pass_rows;  // [n_pass] first row of each such pass
  uint32_t n_breaks;          // rows at which an op stream began: a call of applyChanges or a scheduling pass of one (all calls so far)
  const uint32_t* breaks;     // [n_breaks] ascending
  uint32_t T_doc;             // rows [0, T_doc) are the rebuilt history of a document the lineage began with (Backend.load); 0: none
  uint32_t breaks_exact;      // 0: where EARLIER calls ended is not known (the staged changes were replayed in one go)
  DeltaCounts* counts;
  ObjLink* link;          // [NO]
  // rows: first_del / new_succ per VALUE row (first row of the batch that overwrites or deletes it, how many do); upd_*: the
  // assignment rows (K_LIST_UPD) of every list element, grouped by element
  uint32_t *first_del, *new_succ, *upd_n, *pos_of;    // [N + 1]
  uint32_t *first_kill;                               // [N + 1] first row of the batch that names the row as pred and is NOT an increment of the
                                                      // counter the row sets (an increment does not take a counter away: new.js:937-965)
  uint32_t *upd_off, *upd_cur, *upd_rows;             // [N + 2]
  // per new row that deletes from or assigns to a list element: what the patch shows for it (EV_*), was the element visible before
  // it, how many values it shows afterwards
  uint32_t *ev_kind, *ev_before, *ev_nafter;          // [NN + 1]
  // list positions
  uint32_t *v0, *icnt, *v0_ex, *item_ex, *icur;       // [n_list + 2]
  // items (ping-pong), [NN + 1]
  uint32_t *tk[2], *elem[2], *acc[2], *lo[2], *hi[2];
  uint32_t *zf, *zw, *zf_ex, *zw_ex;                  // [NN + 2]
  uint32_t *e_index, *e_flags, *e_head, *e_head_ex;   // [NN + 2] (e_head: records of the item)
  uint32_t *e_val, *e_val_ex;                         // [NN + 2] values of the item (count of a remove)
  am355_ir_edit* edit;                                // [edit_cap] (NN + 2 from the block; delta_run moves it when the records need more)
  uint32_t edit_cap;
  // touched map keys: open-addressing table of cap slots (power of two)
  uint32_t key_mask;
  uint32_t *slot_rep, *slot_first, *slot_last, *slot_cont, *slot_cnt, *slot_child, *slot_drop, *place, *place_ex;  // [cap + 1]
  unsigned long long* slot_L;                         // [cap]
  // map records
  uint32_t *keep, *keep_ex, *rec_slot;                // [NM + 1]
  uint64_t *pair_key[2];                              // [NM + cap + 1]
  uint32_t *pair_val[2];
  am355_ir_map* map;                                  // [NM + cap + 1]
  void* scan_ws;
  void* sort_ws;
  // counters to the host through pinned words instead of a copy + a blocking wait (am355_internal.h HostSignals; nullptr: copies)
  HostSignals* sig;
  uint32_t sig_seq;
  // kd_edit_small leaves the tables the host wants -- d.link [n_obj], d.edit [n_erecs + 1] -- in pinned host memory itself and signals
  // behind them (nullptr: the caller copies): two copy dispatches and a signalling launch less
  ObjLink* host_link;
  am355_ir_edit* host_edit;
  uint32_t list_only;   // the caller knows that no new row is a map row (am355_resorder.hip served the batch): the stage's map kernels are not launched
};

enum : uint32_t { EV_NONE = 0, EV_INSERT = 1, EV_REMOVE = 2, EV_UPDATE = 3 };

size_t delta_bytes(uint32_t n_ops, uint32_t n_new, uint32_t n_map, uint32_t n_obj, uint32_t n_list);
void delta_bind(DeltaBufs& d, void* block, uint32_t n_ops, uint32_t n_new, uint32_t n_map, uint32_t n_obj, uint32_t n_list);

// Runs the stage on `st` and synchronises it. On return *hc holds the counters (hc->flags != 0: refused / invalid), the delta tables
// d.edit [n_erecs + 1], d.map [n_kept + n_place] and d.link [n_obj] are complete in device memory.
// check_only: stop after the checks over the touched map keys (hc->hazard / hc->flags); no tables are produced. Used on a state
// that one am355_load_changes + am355_replay built (T0 = 0: every row is "new"), before the first am355_apply_changes onto it.
// grow_edit(records): device memory for that many edit records (the caller owns it), nullptr = out of memory. An item of a conflicted
// list element writes one record per visible value, so the records can outnumber the new rows the block was carved for.
typedef am355_ir_edit* (*DeltaGrowEdit)(void* user, size_t records);
// before_end(mid, rec_bound): called when every kernel of the stage is enqueued and nothing has been waited for since the first half:
// *mid holds the first half's counters (n_kept, n_place: final), rec_bound >= n_erecs + 1. The caller may enqueue the copies of the
// tables it wants (d.link, d.map, at most rec_bound records of d.edit) on `st`: the stage's last wait then covers them too.
typedef void (*DeltaBeforeEnd)(void* user, const DeltaCounts* mid, size_t rec_bound);
void delta_run(MergeBufs& b, PatchIR& ir, DeltaBufs& d, DeltaCounts* hc, hipStream_t st, bool check_only = false, DeltaGrowEdit grow_edit = nullptr,
               void* grow_user = nullptr, DeltaBeforeEnd before_end = nullptr);

// What the reference's objectMeta holds in `children[key]` for the property (map key or list element) that holds -- or held -- each
// of the given objects: the visible values, or nothing (KH_DEAD). It is refreshed only while it is non-empty or a child object is
// visible (new.js:916-931), row by row of every merge call that visits the property: once a visit leaves it empty, later plain values
// do not bring it back; and it lists the values the LAST visit saw, which a call that went on to a greater key did not all look at
// (kd_slots). One thread replays the visits of a property from the rows on it (their row numbers are the times). `objects` (host,
// object indexes) -> `out` (host): KH_LIVE with the listed values / KH_DEAD / KH_UNKNOWN (more rows than the walk takes, or a visit
// the walk does not model). Synchronises `st`.
enum : uint8_t { KH_UNKNOWN = 0, KH_LIVE = 1, KH_DEAD = 2 };
constexpr uint32_t KH_VALUES_MAX = 16;  // values of one property the walk reports (more: KH_UNKNOWN)
struct KeyHistory {
  uint8_t state = KH_UNKNOWN;
  uint32_t n = 0;                         // KH_LIVE: the op ids of the listed values, ascending
  uint32_t ctr[KH_VALUES_MAX], actor[KH_VALUES_MAX];
};
int delta_key_history(MergeBufs& b, PatchIR& ir, DeltaBufs& d, const uint32_t* objects, uint32_t n, KeyHistory* out, hipStream_t st);

}  // namespace am355
