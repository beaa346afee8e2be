// Parallel decode of the multi-megabyte columns of a saved document (see am355_bigcol.hip).
#pragma once
#include "am355_internal.h"
#include <stddef.h>

namespace am355 {

constexpr int BIG_NCOL = 12;
// order of the tokenisable columns at the front of the document arena
enum BigCol { BC_OBJ_ACTOR, BC_OBJ_CTR, BC_KEY_ACTOR, BC_KEY_CTR, BC_ID_ACTOR, BC_ID_CTR, BC_INSERT, BC_ACTION, BC_VAL_LEN, BC_SUCC_NUM, BC_SUCC_ACTOR, BC_SUCC_CTR };
enum BigKind : uint32_t { BK_UINT = 0, BK_DELTA = 1, BK_BOOL = 2 };

struct BigColDesc {
  uint32_t off[BIG_NCOL], len[BIG_NCOL], kind[BIG_NCOL];
  uint32_t tok_bytes;  // the columns occupy arena[0, tok_bytes)
};

// filled on the device, read back once by the host
struct BigColInfo {
  uint32_t t0[BIG_NCOL], t1[BIG_NCOL];  // token range of each column
  uint32_t r0[BIG_NCOL], r1[BIG_NCOL];  // record range
  uint32_t rows[BIG_NCOL];              // values the column holds
  uint32_t n_tokens, n_records, flags, n_succ;  // n_succ = sum of succNum over the rows
};

// all arrays are sized by the byte count of the columns (tokens <= bytes, records <= tokens), so nothing here waits for a
// device-side count; cap = tok_bytes + 2
struct BigColWork {
  uint32_t *term_ex, *tok_end, *tok_lo, *tok_hi, *jump_a, *jump_b, *mark, *rec_ex, *rec_tok, *rec_rows, *rec_start;  // [cap]
  uint16_t* tok_meta;                                                                                           // [cap] byte count | last byte << 8
  void* scan_ws;                                                                                                // scan_workspace_bytes(cap)
  void* chain_ws;                                                                                               // chain_work_bytes(cap)
  BigColInfo* info;                                                                                             // device
};
size_t bigcol_work_bytes(uint32_t tok_bytes);
void bigcol_carve(BigColWork& w, void* base, uint32_t tok_bytes);

// value arrays: NONE32 = null for the plain columns; delta columns carry a separate null mask
struct BigColVals {
  uint32_t* v[BIG_NCOL];       // [rows + 1]
  uint8_t* key_ctr_null;       // [N + 1]
  uint32_t *val_off, *succ_first;  // [N + 1] exclusive prefix sums of value lengths / succ counts
  uint32_t* tmp;               // [max(N, P) + 1]
  void* scan_ws;               // scan_workspace_bytes(max(N, P) + 1)
  uint32_t* tile_rec;          // [BIG_NCOL x tile_stride] record that holds the first row of every stretch of 1024 rows (kb_tile_recs)
  uint32_t tile_stride;
};
size_t bigcol_vals_bytes(uint32_t n_rows, uint32_t n_succ);
void bigcol_carve_vals(BigColVals& v, void* base, uint32_t n_rows, uint32_t n_succ);

// numbers (info->n_tokens, column token ranges), then -- with n_tokens read back by the caller -- records and row starts
// (column record ranges, rows per column in *w.info)
void bigcol_index_tokens(const uint8_t* arena, const BigColDesc& d, BigColWork& w, hipStream_t st);
void bigcol_index_records(const uint8_t* arena, const BigColDesc& d, BigColWork& w, uint32_t n_tokens, hipStream_t st);
// values of every row of every column, delta / offset prefix sums; info->n_succ
void bigcol_expand(const BigColDesc& d, const BigColWork& w, const BigColInfo& h, BigColVals& v, uint32_t n_rows, uint32_t n_succ_cap, hipStream_t st);
// fixed-width op rows from the value arrays (key strings are filled by launch_keystr_expand)
void bigcol_assemble(const BigColVals& v, uint32_t n_rows, uint32_t n_succ, const uint32_t* actor_rank, uint32_t n_actors, uint32_t val_raw_abs,
                     uint32_t val_raw_len, OpCols o, uint32_t* flags, hipStream_t st);

// keyStr column -> run table (run_start has n_runs + 1 entries; run_len NONE32 = null run), fully parallel; scratch in `work`
// (keystr_work_bytes(col_len) bytes). `begin` (vnext / hnext of every position with the k-th successors of literal headers resolved
// tile by tile in LDS, then the true headers) and `finish` (literal items, run table) are enqueued back to back: no host decision
// in between. d_unresolved: four device words the caller cleared ([0], [1] pending counts; flags[3] of `finish` = 1 when the TRUE parse
// reached a literal the walker had cut off after KeyStage.max_jumps windows: the caller repeats the load with max_jumps = 0).
struct KeyWork {
  uint32_t *vnext, *hnext, *kk, *ja, *jb, *mark_h, *mark_v, *item_ex, *rows;  // [L + 2]
  uint32_t *run_start, *run_off, *run_len, *run_kind;                        // [L + 2] (items <= bytes)
  uint32_t* n_runs;                                                          // device word
  void* scan_ws;
};
struct KeyStage {
  KeyWork k;
  void* chain_ws;
  const uint8_t *col, *arena;
  uint32_t col_abs, L;
  uint32_t max_jumps = 64;   // windows the continuation walker follows a literal through (0: no bound); set before keystr_index_begin
};
size_t keystr_work_bytes(uint32_t col_len);
void keystr_index_begin(const uint8_t* arena, uint32_t col_abs, uint32_t col_len, void* work, KeyStage& s, uint32_t* n_runs, uint32_t* d_unresolved,
                        hipStream_t st);
void keystr_index_finish(KeyStage& s, bool unresolved, uint32_t** run_start, uint32_t** run_off, uint32_t** run_len, uint32_t* flags, hipStream_t st);

}  // namespace am355
