// History reconstruction after Backend.load on the device: see am355_hist.hip.
#pragma once
#include "am355_encode.h"
#include "am355_internal.h"

namespace am355 {

enum : uint32_t { HF_INVALID = 1u, HF_UNSUPPORTED = 2u };  // HistBufs.flags: what the reference does with this document (throws / served by the JS path)
constexpr int HIST_NCOL = 12;  // change columns in id order: objActor objCtr keyActor keyCtr keyStr insert action valLen valRaw predNum predActor predCtr

// Device memory of the reconstruction. N = document rows, P = succ entries, NC = changes, NA = actors, W = 32-bit words of the id
// bitmaps (one bit per (actor, counter) up to the actor's last maxOp). Every id the document mentions -- row ids and ids in succ lists --
// gets a SLOT: ids numbered per actor in counter order = the order of ops inside a change and of changes inside an actor. M <= N + P.
struct HistBufs {
  uint32_t N, P, NC, NA, W, AW;    // AW = words of a change's actor bitmap
  uint32_t* flags;                 // [4]: HF_* | -- | M (slots) | total preds
  // uploaded by the host
  uint32_t *word_base, *act_max;   // [NA + 1], [NA]
  uint32_t *chg_actor, *chg_prev_max, *chg_max;   // [NC]
  uint32_t *sorted_base, *sorted_chg;             // [NCs] the changes that own slots, ascending by first slot (second upload)
  // ids -> slots
  uint32_t *all_bits, *row_bits, *word_cnt, *word_rank;   // [W + 1]
  uint32_t *slot_row, *slot_ref;   // [M]: row that carries the id | first row that lists it in its succ list (NONE32: none)
  uint32_t *pred_cnt, *pred_first, *pred_cur;    // [M + 1]
  uint32_t* pred_row;              // [P]: rows overwritten by the op in the slot (the inverse of the succ lists), ascending by id per slot
  // changes -> slot ranges
  uint32_t *chg_base, *chg_nops;   // [NC]
  uint32_t* abits;                 // [NC x AW] actors a change mentions
  // the changes' op columns, by slot (seg: first slot of the slot's change) and by pred entry (pseg)
  uint32_t *seg, *chg_of;          // [M]
  uint32_t *v_obj_actor, *v_obj_ctr, *v_key_actor, *v_key_ctr, *v_key_off, *v_key_len, *v_action, *v_val_tl, *v_val_off, *v_pred_num;   // [M]
  uint8_t* v_insert;               // [M]
  uint32_t *p_actor, *p_ctr, *pseg;   // [P]
  uint32_t *seg_base, *pseg_base;  // [NC + 1] first slot / first pred entry of every change (document order of changes)
  // encoder work + output
  EncWork enc;
  uint32_t* deltas;                // [max(M, P) + 2]
  uint8_t* nullmask;               // [max(M, P) + 2]
  uint8_t* col_out[HIST_NCOL];     // encoded bytes of a column, all changes back to back
  size_t col_cap[HIST_NCOL];
  uint32_t* col_off;               // [HIST_NCOL][2 x (NC + 1)]: begin / end of change k in column q at [q][2k], [q][2k + 1]
  uint32_t* col_len;               // [HIST_NCOL] total bytes
  void* scan_ws;
};
size_t hist_bytes(uint32_t N, uint32_t P, uint32_t NC, uint32_t NA, uint32_t W, size_t key_bytes, size_t val_bytes);
void hist_bind(HistBufs& h, void* block, uint32_t N, uint32_t P, uint32_t NC, uint32_t NA, uint32_t W, size_t key_bytes, size_t val_bytes);

// Stage 1 (enqueued on st): id bitmaps, slots, preds by slot, slot range of every change. Afterwards the host reads h.flags,
// h.chg_base, h.chg_nops (it needs the changes that own slots in slot order for stage 2, and the op counts for the headers).
void hist_stage1(const OpCols& rows, HistBufs& h, hipStream_t st);
// Stage 2: actor tables, the op columns of every change (values by slot), the twelve column encodes segmented by change.
// n_sorted: entries of h.sorted_base / h.sorted_chg. Afterwards h.col_out / h.col_off / h.col_len / h.abits / h.flags are complete.
void hist_stage2(const OpCols& rows, const uint8_t* arena, size_t arena_len, HistBufs& h, uint32_t n_sorted, uint32_t M, hipStream_t st);

}  // namespace am355
