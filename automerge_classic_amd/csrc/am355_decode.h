// Stage 1 launchers (see am355_decode.hip).
#pragma once
#include "am355_internal.h"

namespace am355 {
// word ranges k_parse_changes clears on its way in (value per range); n = 0: none
struct ParseFills {
  uint32_t* p[5];
  uint64_t n_words[5];
  uint32_t value[5];
  uint32_t n;
};
// the same records computed by the host from the bytes it holds (parse_change compiled for the host; small batches)
void parse_changes_host(const uint8_t* arena, const uint64_t* offsets, uint32_t n_changes, ChangeMeta* metas, uint32_t* n_entries);
void launch_parse_changes(const uint8_t* arena, const uint64_t* offsets, uint32_t n_changes, ChangeMeta* metas, uint32_t* n_entries, const ParseFills& fills,
                          hipStream_t st, bool fat = false);
// decoder class of a parsed change for a host-built plan (what k_actor_check writes into ChangeBrief.flags_fits): 2 small wave class,
// 1 large wave class, 0 lane-serial
int change_wave_class(const ChangeMeta& m);
void launch_hash_changes(const uint8_t* arena, const uint64_t* offsets, uint32_t n, uint8_t* hashes, uint32_t* min_idx, uint32_t* hash_tab,
                         uint32_t tab_mask, uint32_t* flags, hipStream_t st);
void launch_deps_resolve(const uint8_t* arena, const ChangeMeta* metas, const uint8_t* hashes, uint32_t n, const uint32_t* hash_tab, uint32_t tab_mask,
                         const uint32_t* min_idx, uint8_t* has_dependent, uint32_t* fast_flags, uint32_t* dep_idx, uint32_t* self_idx, hipStream_t st);
// (dep_idx[(arena offset of a dependency hash) / 32] = index of the first change of the batch with that hash, NONE32 if none;
//  self_idx[c] = first change with c's own hash: what the host's general scheduler works on instead of hash lookups)
// distinct: [0] count of claimed actor-table slots, [1..] their indexes (capacity distinct_capacity()); briefs: per-change digest for the host.
// k_actor_check also starts the device half of the in-order plan: per-workgroup sums of ops / preds / actor entries / plans per decoder
// class (block_sums, plan_block_sums_bytes(n) bytes), lexicographic ranks of the distinct actor ids (slot_rank[slot]) and plan_words
// ([0] fallback, [1] max op, [2] OR of change flags, [3] unknown columns; cleared by the caller).
// rank_ids: rank_ids_bytes() bytes of device memory (the ids of the distinct actors as the ranking workgroup reads them; needs no clearing)
void launch_actor_intern(const uint8_t* arena, ChangeMeta* metas, uint32_t n, const uint32_t* amap_base, uint32_t* amap, uint32_t amap_cap,
                         unsigned long long* slots, uint32_t slot_mask, uint32_t* first_idx, uint32_t* flags, uint32_t* fast_flags, uint32_t* distinct,
                         void* rank_ids, ChangeBrief* briefs, uint32_t* slot_rank, unsigned long long* block_sums, uint32_t* plan_words, hipStream_t st,
                         ChangeBrief* host_briefs = nullptr);
size_t rank_ids_bytes();
size_t plan_block_sums_bytes(uint32_t n);
uint32_t distinct_capacity();
// ... and k_plan_apply finishes it: for every change with ops its ChangePlan (row / pred / actor-table bases by prefix sums in input order,
// author rank) by decoder class (`plans`: small class from the front, large class from the back; `plans_serial`: the rest) -- what the
// decode kernels need, so that they can start while the host is still validating sequence numbers and building the per-actor span
// tables. The totals go to the host through `sig` (HostSignals.plan).
void launch_plan(const ChangeBrief* briefs, uint32_t n, const uint32_t* distinct, const uint32_t* slot_rank, uint32_t slot_mask, const unsigned long long* block_sums,
                 ChangePlan* plans, ChangePlan* plans_serial, const uint32_t* words, const uint32_t* plan_words, HostSignals* sig, uint32_t seq, hipStream_t st,
                 PlanTotals* dev_totals = nullptr, uint32_t* host_s1 = nullptr);
void launch_decode_speculative(const uint8_t* arena, const ChangeMeta* metas, const ChangePlan* plans, uint32_t n_changes, const PlanTotals* totals, uint32_t cap_ops,
                               uint32_t cap_preds, uint32_t cap_distinct, const uint32_t* amap, const uint32_t* slot_rank, OpCols cols, uint32_t* flags, hipStream_t st,
                               hipStream_t aux, bool with_large, uint32_t shard_rank = 0, uint32_t shard_world = 1);
void launch_decode_planned(const uint8_t* arena, const ChangeMeta* metas, const ChangePlan* plans, const ChangePlan* plans_serial, uint32_t n_changes,
                           uint32_t n_small, uint32_t n_large, uint32_t n_serial, const uint32_t* amap, const uint32_t* slot_rank, OpCols cols, uint32_t* flags,
                           hipStream_t st, hipStream_t aux, uint32_t shard_rank = 0, uint32_t shard_world = 1);
// slot_rank == nullptr: `amap` already holds global actor ranks. plans = [n_small | n_large wave-decodable | n_serial others]
// (ChangeBrief.flags_fits bit 30: small wave class, bit 31: any wave class). `aux`: a stream the caller forked from `st` and joins
// afterwards (the second decoder class runs there)
void launch_decode_columns(const uint8_t* arena, const ChangeMeta* metas, const ChangePlan* plans, uint32_t n_small, uint32_t n_large, uint32_t n_serial,
                           const uint32_t* amap, const uint32_t* slot_rank, OpCols cols, uint32_t* flags, hipStream_t st, hipStream_t aux, uint32_t shard_rank = 0,
                           uint32_t shard_world = 1);
// (shard_world > 1, objectId sharding: the wave decoder stops behind the object columns for a change that holds no row of an object
// this rank owns -- what k_resolve reads of foreign rows is decoded by then)
// documents: count rows / succ entries into meta->n_ops / n_preds, then decode all op columns of the one pseudo-change
void launch_doc_count(const uint8_t* arena, ChangeMeta* meta, hipStream_t st);
void launch_decode_document(const uint8_t* arena, const ChangeMeta* meta, const ChangePlan* plan, const uint32_t* actor_rank, OpCols cols,
                            uint32_t* flags, hipStream_t st);  // host: can this change use the wave-per-change decoder?
// document keyStr column: key_off / key_len of every row from the run table built by keystr_index (am355_bigcol.hip)
void launch_keystr_expand(const uint32_t* run_start, const uint32_t* run_off, const uint32_t* run_len, const uint32_t* n_runs, uint32_t n_rows,
                          uint32_t* key_off, uint32_t* key_len, hipStream_t st);
}  // namespace am355
