// General causal scheduler on the device (reference backend/new.js:1550-1597 `applyChanges` inside the retry loop
// of BackendDoc.applyChanges :1822-1841): any delivery order, duplicates, missing dependencies. See am355_sched.hip.
#pragma once
#include "am355_internal.h"

namespace am355 {

constexpr uint32_t SCHED_NEVER = 0xfffffffeu;  // pass of a change that is never applied (missing dependency, later copy of a duplicate)
constexpr uint32_t SCHED_UNSET = 0xffffffffu;

// result words of the scheduler (device; the last kernel also sends them to the host inside PlanTotals.reserved)
enum SchedWord : uint32_t {
  SW_N_APPLIED = 0,   // changes that are applied
  SW_MAX_PASS,        // number of the last scheduling pass
  SW_UNFINISHED,      // 1: the relaxation ran out of sweeps (pathological dependency chains): the host schedules
  SW_FLAGS,           // F_UNKNOWN_ACTOR_DEV: a change mentions an actor without an earlier change (new.js:1442-1449)
  SW_MAX_OP,
  SW_NUM = 16
};

struct SchedBufs {
  uint32_t n;
  uint32_t* pass;        // [n] scheduling pass of every change, SCHED_NEVER: never applied
  uint32_t *cursor, *curmax, *dfirst, *dcnt;  // [n] global-memory form of ks_pass (batches beyond its LDS size): wait-on / work list / dependency ranges
  uint32_t *apos, *left;      // [n] per group of copies of one change (at its first copy): position of the copy that is applied | undecided copies
  unsigned long long* best;   // [n] min over a group's decided copies of (pass << 32 | position)
  uint64_t *key_a, *key_b;    // [n] sort keys (pass; n for changes never applied)
  uint32_t *val_a, *val_b;    // [n] change indexes
  void* sort_ws;
  uint32_t* rank_of;     // [n] application rank of a change (NONE32: never applied)
  uint32_t* first_rank;  // [slot_mask + 1] first application rank of a change authored by the actor in that slot
  uint8_t* is_head;      // [n] applied and no applied change depends on it
  uint32_t* words;       // [SW_NUM]
  unsigned long long* block_sums;  // plan_block_sums_bytes(n)
};
size_t sched_bytes(uint32_t n, uint32_t slot_mask);
void sched_bind(SchedBufs& s, void* block, uint32_t n, uint32_t slot_mask);

// Enqueues the whole scheduler on `st`:
//   pass(c) = max over the dependencies d of c of  pass(d) + [d stands behind c in the queue]   (0 without dependencies; never, if a
//   dependency is not in the batch or is never applied itself; later copies of a change are never applied) -- the pass of the
//   reference's retry loop in which c is applied; application order = (pass, position in the queue);
//   then the decode plans in THAT order (row / pred bases by prefix sums over the applied changes, decoder classes as launch_plan
//   does for the in-order path), the heads, the actor rule, and the totals to the host through HostSignals.plan (PlanTotals.reserved
//   [1] = applied changes, [2] = last pass, [3] = 1 if unfinished).
// dep_idx / self_idx: launch_deps_resolve. amap / amap_base: the changes' actor tables as slots (launch_actor_intern). `order` (device,
// [n]) receives the change indexes in application order (the first n_applied entries).
void launch_sched_general(const ChangeMeta* metas, const ChangeBrief* briefs, uint32_t n, const uint32_t* dep_idx, const uint32_t* self_idx,
                          const uint32_t* amap, const uint32_t* amap_base, uint32_t amap_cap, const uint32_t* slot_rank, uint32_t slot_mask, SchedBufs& s,
                          uint32_t** order, ChangePlan* plans, ChangePlan* plans_serial, const uint32_t* stage_words, const uint32_t* distinct, HostSignals* sig, uint32_t seq,
                          hipStream_t st);

}  // namespace am355
