// Shared host/device data layout of the replay engine.
#pragma once
#include "am355_device.h"

namespace am355 {

// column slots of a change (reference: backend/columnar.js:56-78 CHANGE_COLUMNS)
// (a saved document stores the same columns plus the op ids, and succ lists instead of pred lists -- columnar.js:80-84;
//  when a document is decoded the three C_PRED_* slots hold its succNum / succActor / succCtr columns)
enum ColSlot { C_OBJ_ACTOR, C_OBJ_CTR, C_KEY_ACTOR, C_KEY_CTR, C_KEY_STR, C_INSERT, C_ACTION, C_VAL_LEN, C_VAL_RAW, C_PRED_NUM, C_PRED_ACTOR, C_PRED_CTR,
               C_ID_ACTOR, C_ID_CTR, C_NUM };

// One record per binary change, written by k_parse_changes (device) and read by the host scheduler.
// Offsets are relative to `base` (the change's first byte in the raw arena).
struct ChangeMeta {
  uint64_t base;
  uint32_t len;
  uint32_t flags;
  uint64_t seq, start_op;
  uint32_t n_entries;    // actor table entries: author + others
  uint32_t author_slot;  // slot of the author in the device actor table (k_actor_intern)
  uint32_t max_first;    // max over its actor entries of "first change index authored by that actor" (k_actor_check)
  uint32_t pad;  // bit 0: the change has non-empty columns outside the modelled set
  uint32_t n_deps, deps_off;
  uint32_t actor_off, actor_len;
  uint32_t n_other, others_off;
  uint32_t col_off[C_NUM], col_len[C_NUM];
  uint32_t n_ops, n_preds;
};

// What the host's in-order fast path needs per change (32 B instead of the full ChangeMeta), written by k_actor_check
struct ChangeBrief {
  uint64_t seq;
  uint32_t start_op, n_ops, n_preds, n_entries, author_slot;
  uint32_t flags_fits;  // validity flags of the change; bit 31: the columns fit the wave-per-change decoder, bit 30: its small LDS class, bit 29: unmodelled columns present
};

// bits of the "fast path" word: any bit set => the host runs the general scheduler (new.js:1550-1597) itself
enum FastFlag : uint32_t {
  FF_DUP_HASH = 1u << 0,      // the same change appears twice
  FF_MISSING_DEP = 1u << 1,   // a dependency is not in the batch
  FF_LATE_DEP = 1u << 2,      // a dependency appears later in the batch than its dependent
  FF_LATE_ACTOR = 1u << 3,    // a change mentions an actor whose first change comes later in the batch
  FF_CAPACITY = 1u << 4       // actor-table staging buffer too small (retry after growing it)
};

// Per-change launch parameters produced by the host scheduler for the decode kernels (applied changes only)
struct ChangePlan {
  uint32_t change;     // index into ChangeMeta[]
  uint32_t op_base;    // first row of this change in the op arrays
  uint32_t pred_base;  // first entry in the pred arrays
  uint32_t amap_base;  // first entry of its local->global actor translation table
  uint32_t author;     // global actor rank of the author
  uint32_t n_actors;   // entries in its actor table
};

// What k_plan reports to the host: the host sizes buffers and launch grids with it while it is still checking sequence numbers and
// building the per-actor tables itself
struct PlanTotals {
  uint32_t n_ops, n_preds, n_entries;
  uint32_t n_small, n_large, n_serial;  // plans per decoder class (changes without ops have no plan)
  uint32_t max_op;
  uint32_t fallback;                    // 1: more distinct actors / longer actor ids than the device ranking handles, or sums beyond 32 bits -- the host plans
  uint32_t flags_a, fast_a, total_entries, n_distinct;  // the stage-1 words the host decides on (validity flags, fast-path word, actor-table entries, distinct actors)
  uint32_t reserved[4];
};

// The decode kernels of the in-order path are enqueued behind k_plan_apply BEFORE the host has read these totals (the host used to
// sit between the two with a signal, its buffer carving and a launch: 50 us of idle device in a 400 us replay). They run over rows
// carved for a CAPACITY (the previous replay's size or an estimate from the encoded bytes) and do nothing unless this test holds --
// evaluated by every wavefront on the device copy of the totals and by the host on its own copy: in order, well-formed, planned on the
// device, and inside the capacity. Otherwise the host carves for the real totals and launches as before.
__host__ __device__ inline bool decode_gate_open(const PlanTotals& t, uint32_t cap_ops, uint32_t cap_preds, uint32_t cap_distinct) {
  return !t.fallback && !t.flags_a && !t.fast_a && t.n_ops <= cap_ops && t.n_preds <= cap_preds && t.n_distinct <= cap_distinct;
}
struct DecodeGate {
  const PlanTotals* totals = nullptr;   // null: an ordinary launch (grid = plans of the class)
  uint32_t cap_ops = 0, cap_preds = 0, cap_distinct = 0, n_changes = 0, large = 0;
};

// Device -> host signalling without a copy and without a blocking wait: the LAST kernel of a phase writes its few result words
// straight into pinned host memory (fine-grained, device-visible), fences at system scope and then publishes the sequence number
// of the replay; the host spins on that word. A D2H copy + hipStreamSynchronize costs a copy dispatch (~15 us until it runs, ~5 us
// on the stream) and an interrupt wake-up (~30 us) on a path whose kernels take 5-60 us each.
struct HostSignals {
  volatile uint32_t plan_seq;    uint32_t pad0[15];
  PlanTotals plan;
  volatile uint32_t counts_seq;  uint32_t pad1[15];
  uint32_t counts[32];           // Counts after k_compact_rows, MapKeyStats behind it
  volatile uint32_t runs_seq;    uint32_t pad2[15];
  uint32_t runs[16];             // Counts after k_run_heads
  volatile uint32_t final_seq;   uint32_t pad3[15];
  uint32_t final_counts[16];     // Counts after k_edit_pack
  // Backend.applyChanges (signalled by a one-thread launch behind the phase, am355_prims.h launch_signal_words)
  volatile uint32_t resorder_seq;  uint32_t pad4[15];
  uint32_t resorder[16];         // ResOrderBufs.words [8] | Counts.flags after the in-place list order merge
  volatile uint32_t delta_mid_seq; uint32_t pad5[15];
  uint32_t delta_mid[16];        // DeltaCounts after the first half of the delta stage
  volatile uint32_t delta_end_seq; uint32_t pad6[15];
  uint32_t delta_end[16];        // DeltaCounts after the stage (behind the copies of its tables, when the caller enqueued them)
};

// Per-actor lookup table entry for opId -> row resolution: the applied changes of one actor, ascending start_op
struct ActorSpan {
  uint32_t start_op, n_ops, op_base;
};

// Fixed-width op rows (structure of arrays, one u32 per field per op; reference row layout new.js:10-12).
// Actor fields hold GLOBAL actor ranks (lexicographic rank of the raw actor id among all actors of the
// batch), so comparing (ctr, actor) pairs numerically equals the reference's (counter, actorId string) order.
struct OpCols {
  uint32_t *obj_actor, *obj_ctr;      // obj_actor == NONE32: _root
  uint32_t *key_actor, *key_ctr;      // key_ctr == NONE32: no element key; key_ctr == 0: _head
  uint32_t *key_off, *key_len;        // key_len == NONE32: no string key; key_off is an absolute arena offset
  uint32_t *action, *val_tl, *val_off; // val_tl = (len << 4) | type tag; val_off absolute arena offset
  uint32_t *pred_first, *pred_num;
  uint32_t *id_ctr, *id_actor;
  uint8_t* insert;
  uint32_t *pred_actor, *pred_ctr;    // flattened pred lists
};

// Owner of an object's rows: _root on rank 0, every other object by its id (ordering and pred / succ resolution
// never cross objects, new.js:1141-1145, 1173-1176; the one cross-object link, make row -> child object, is the object index).
__host__ __device__ __forceinline__ uint32_t shard_owner(uint32_t obj_actor, uint32_t obj_ctr, uint32_t world) {
  return obj_actor == NONE32 ? 0u : (obj_ctr + obj_actor) % world;
}

}  // namespace am355
