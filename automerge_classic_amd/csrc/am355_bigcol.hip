// Parallel decode of big RLE / delta / boolean columns (document load, BASELINE config 5).
//
// The columns of a saved document run to megabytes, so neither the wave-per-change decoder (LDS staging) nor a lane per
// column will do. The scheme (reference codecs: backend/encoding.js:789-920 RLE, 1004-1051 delta, 1141-1207 boolean):
//   A. tokens   every byte with bit 7 clear ends a LEB128 number: flag + prefix sum over the bytes of all columns at once
//               gives each number's index; one lane per number assembles it.
//   B. records  seen as a record header, token t names its successor (t+2 for a repetition or null run, t+1+k for a
//               literal of k values; t+1 in a boolean column). The real headers are the orbit of the column's first
//               token under that map: marked by chain_mark (am355_prims.hip: tile exits in LDS, compacted entry graph),
//               all columns together.
//   C. rows     rows per record -> prefix sum -> row start of every record; every row binary-searches its record and reads
//               its value (repetition: the record's value token; literal: value token + offset; null run: null; boolean:
//               record parity). Delta columns, value offsets and succ-list offsets are further prefix sums.
// All of it is streaming integer work over HBM-resident arrays (prefix sums, gathers); no MFMA.
#include "am355_bigcol.h"
#include "am355_prims.h"
#include <cstdio>
#include <cstdlib>

namespace am355 {

static inline dim3 grid_for(uint32_t n) { return dim3((n + BLOCK - 1) / BLOCK); }
constexpr uint64_t MAX_SAFE_BIG = 9007199254740991ull;

constexpr uint32_t EXPAND_TILE = 1024;   // rows per entry of the coarse row -> record table (kb_tile_recs)
static size_t al256(size_t b) { return carve_round(b); }
size_t bigcol_work_bytes(uint32_t tok_bytes) {
  size_t cap = (size_t)tok_bytes + 2;
  return 11 * al256(4 * cap) + al256(2 * cap) + al256(scan_workspace_bytes((uint32_t)cap)) + al256(sizeof(BigColInfo)) + al256(chain_work_bytes((uint32_t)cap));
}
void bigcol_carve(BigColWork& w, void* base, uint32_t tok_bytes) {
  canary_scope("document token work (bigcol_carve)");
  size_t cap = (size_t)tok_bytes + 2;
  uint8_t* p = (uint8_t*)base;
  auto take = [&](size_t bytes) { void* r = p; canary_note(p, bytes); p += al256(bytes); return r; };
  uint32_t** arrs[] = {&w.term_ex, &w.tok_end, &w.tok_lo, &w.tok_hi, &w.jump_a, &w.jump_b, &w.mark, &w.rec_ex, &w.rec_tok, &w.rec_rows, &w.rec_start};
  for (uint32_t** a : arrs) *a = (uint32_t*)take(4 * cap);
  w.tok_meta = (uint16_t*)take(2 * cap);
  w.scan_ws = take(scan_workspace_bytes((uint32_t)cap));
  w.info = (BigColInfo*)take(sizeof(BigColInfo));
  w.chain_ws = take(chain_work_bytes((uint32_t)cap));
}
size_t bigcol_vals_bytes(uint32_t n_rows, uint32_t n_succ) {
  size_t n = (size_t)n_rows + 1, p = (size_t)n_succ + 1;
  return 12 * al256(4 * n) + 2 * al256(4 * p) + al256(n) + al256(4 * (n > p ? n : p)) + al256(scan_workspace_bytes((uint32_t)(n > p ? n : p))) +
         al256(4 * BIG_NCOL * ((n > p ? n : p) / EXPAND_TILE + 3));
}
void bigcol_carve_vals(BigColVals& v, void* base, uint32_t n_rows, uint32_t n_succ) {
  canary_scope("document column values (bigcol_carve_vals)");
  size_t n = (size_t)n_rows + 1, pn = (size_t)n_succ + 1;
  uint8_t* p = (uint8_t*)base;
  auto take = [&](size_t bytes) { void* r = p; canary_note(p, bytes); p += al256(bytes); return r; };
  for (int c = 0; c < BIG_NCOL; c++) v.v[c] = (uint32_t*)take(4 * (c >= BC_SUCC_ACTOR ? pn : n));
  v.val_off = (uint32_t*)take(4 * n);
  v.succ_first = (uint32_t*)take(4 * n);
  v.key_ctr_null = (uint8_t*)take(n);
  v.tmp = (uint32_t*)take(4 * (n > pn ? n : pn));
  v.scan_ws = take(scan_workspace_bytes((uint32_t)(n > pn ? n : pn)));
  v.tile_stride = (uint32_t)((n > pn ? n : pn) / EXPAND_TILE + 3);
  v.tile_rec = (uint32_t*)take(4 * BIG_NCOL * (size_t)v.tile_stride);
}

// ---- tokens -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void kb_token_ends(const uint8_t* __restrict__ arena, uint32_t n, const uint32_t* __restrict__ term_ex,
                                                       uint32_t* __restrict__ tok_end) {
  uint32_t i = gtid();
  if (i < n && !(arena[i] & 0x80)) tok_end[term_ex[i]] = i;
}

__global__ __launch_bounds__(WAVE) void kb_col_ranges(BigColDesc d, const uint32_t* __restrict__ term_ex, BigColInfo* __restrict__ info) {
  uint32_t c = threadIdx.x;
  if (c >= BIG_NCOL) return;
  info->t0[c] = term_ex[d.off[c]];
  info->t1[c] = term_ex[d.off[c] + d.len[c]];
}

__device__ __forceinline__ bool big_tok_sint(uint32_t lo, uint32_t hi, uint32_t meta, int64_t& out) {
  uint32_t nb = meta & 0xff, last = meta >> 8;
  if (nb == 0 || nb > 10 || (nb == 10 && last != 0 && last != 0x7f)) return false;
  uint64_t v = (uint64_t)hi << 32 | lo;
  if ((last & 0x40) && 7 * nb < 64) v |= ~0ull << (7 * nb);
  out = (int64_t)v;
  return out <= (int64_t)MAX_SAFE_BIG && out >= -(int64_t)MAX_SAFE_BIG;
}
__device__ __forceinline__ bool big_tok_uint(uint32_t lo, uint32_t hi, uint32_t meta, uint64_t& out) {
  uint32_t nb = meta & 0xff, last = meta >> 8;
  if (nb == 0 || nb > 10 || (nb == 10 && (last & 0xfe))) return false;
  out = (uint64_t)hi << 32 | lo;
  return out <= MAX_SAFE_BIG;
}
__device__ __forceinline__ int big_col_of(const BigColInfo* info, uint32_t t) {
  int c = 0;
  for (int k = 0; k < BIG_NCOL; k++)
    if (t >= info->t0[k] && t < info->t1[k]) c = k;
  return c;
}

// one lane per token: assemble its value; as a would-be record header, where does the next header sit?
__global__ __launch_bounds__(BLOCK) void kb_token_values(const uint8_t* __restrict__ arena, BigColDesc d, uint32_t cap, BigColWork w) {
  uint32_t t = gtid();
  if (t >= cap) return;
  uint32_t n_tokens = w.info->n_tokens;
  w.rec_rows[t] = 0;
  if (t >= n_tokens) { w.jump_a[t] = NONE32; w.mark[t] = 0; return; }
  uint32_t end = w.tok_end[t], start = t ? w.tok_end[t - 1] + 1 : 0;
  uint32_t nb = end - start + 1;
  uint64_t v = 0;
  if (nb <= 10)
    for (uint32_t k = 0; k < nb; k++) v |= (uint64_t)(arena[start + k] & 0x7f) << (7 * k);
  uint32_t meta = (nb <= 10 ? nb : 0xffu) | (uint32_t)arena[end] << 8;
  w.tok_lo[t] = (uint32_t)v;
  w.tok_hi[t] = (uint32_t)(v >> 32);
  w.tok_meta[t] = (uint16_t)meta;
  int c = big_col_of(w.info, t);
  uint32_t t0 = w.info->t0[c], t1 = w.info->t1[c];
  uint32_t next;
  if (d.kind[c] == BK_BOOL) next = t + 1;
  else {
    int64_t cnt;
    if (!big_tok_sint((uint32_t)v, (uint32_t)(v >> 32), meta, cnt)) next = t + 1;  // not a count: never the header of a well-formed column
    else if (cnt < 0) next = (uint64_t)(-cnt) + 1 > (uint64_t)(t1 - t) ? t1 : t + 1 + (uint32_t)(-cnt);
    else next = t + 2;
  }
  w.jump_a[t] = next >= t1 ? NONE32 : next;
  w.mark[t] = t == t0 ? 1u : 0u;
}

// ---- records ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void kb_records(BigColDesc d, uint32_t cap, BigColWork w) {
  uint32_t t = gtid();
  if (t < BIG_NCOL) {
    w.info->r0[t] = w.rec_ex[w.info->t0[t]];
    w.info->r1[t] = w.rec_ex[w.info->t1[t]];
  }
  if (t >= w.info->n_tokens || !w.mark[t]) return;
  int c = big_col_of(w.info, t);
  uint32_t t0 = w.info->t0[c], t1 = w.info->t1[c];
  uint32_t r = w.rec_ex[t], err = 0;
  uint64_t rows = 0;
  uint32_t lo = w.tok_lo[t], hi = w.tok_hi[t], meta = w.tok_meta[t];
  if (d.kind[c] == BK_BOOL) {
    if (!big_tok_uint(lo, hi, meta, rows)) err |= F_BAD_LEB;
    if (rows == 0 && t != t0) err |= F_BAD_RLE;  // a zero-length run only as the leading `false` run (encoding.js:1180-1184)
  } else {
    int64_t cnt;
    if (!big_tok_sint(lo, hi, meta, cnt)) err |= F_BAD_LEB;
    else if (cnt > 1) {
      rows = (uint64_t)cnt;
      if (t + 1 >= t1) err |= F_BAD_LEB;
    } else if (cnt == 1) err |= F_BAD_RLE;
    else if (cnt < 0) {
      rows = (uint64_t)(-cnt);
      if (rows > (uint64_t)(t1 - t - 1)) { err |= F_BAD_LEB; rows = 0; }
    } else {
      if (t + 1 >= t1 || !big_tok_uint(w.tok_lo[t + 1], w.tok_hi[t + 1], w.tok_meta[t + 1], rows)) err |= F_BAD_LEB;
      else if (rows == 0) err |= F_BAD_RLE;
    }
    // successive literals / successive null runs are not canonical (encoding.js:870-887)
    if (!err && t != t0 && cnt <= 0) {
      // the previous header is not known here; kb_record_pairs checks neighbours once rec_tok is complete
    }
  }
  if (rows > 0x7ffffff0ull) { err |= F_OVERFLOW; rows = 0; }
  w.rec_tok[r] = t;
  w.rec_rows[r] = (uint32_t)rows;
  if (err) atomicOr(&w.info->flags, err);
}

// neighbouring records of one column: two literals or two null runs in a row are rejected by the reference decoder
__global__ __launch_bounds__(BLOCK) void kb_record_pairs(BigColDesc d, BigColWork w) {
  uint32_t r = gtid();
  if (r == 0 || r >= w.info->n_records) return;
  uint32_t t = w.rec_tok[r], tp = w.rec_tok[r - 1];
  int c = big_col_of(w.info, t);
  if (d.kind[c] == BK_BOOL || tp < w.info->t0[c]) return;
  int64_t a = 0, b = 0;
  big_tok_sint(w.tok_lo[tp], w.tok_hi[tp], w.tok_meta[tp], a);
  big_tok_sint(w.tok_lo[t], w.tok_hi[t], w.tok_meta[t], b);
  if ((a < 0 && b < 0) || (a == 0 && b == 0)) atomicOr(&w.info->flags, (uint32_t)F_BAD_RLE);
}

__global__ __launch_bounds__(WAVE) void kb_col_rows(BigColWork w) {
  uint32_t c = threadIdx.x;
  if (c >= BIG_NCOL) return;
  w.info->rows[c] = w.rec_start[w.info->r1[c]] - w.rec_start[w.info->r0[c]];
}

// Step 1: numbers. Leaves info->n_tokens for the host, so that everything after runs over the numbers (about two thirds of
// the byte count for these columns) instead of over the bytes.
void bigcol_index_tokens(const uint8_t* arena, const BigColDesc& d, BigColWork& w, hipStream_t st) {
  uint32_t L = d.tok_bytes;
  (void)hipMemsetAsync(w.info, 0, sizeof(BigColInfo), st);
  exclusive_scan_terminators(arena, L, w.term_ex, &w.info->n_tokens, w.scan_ws, st);  // (index of every number = terminators in front of it)
  AM355_LAUNCH_INDEPENDENT(kb_col_ranges, dim3(1), dim3(WAVE), st, d, (const uint32_t*)w.term_ex, w.info);
  AM355_LAUNCH_INDEPENDENT(kb_token_ends, grid_for(L), dim3(BLOCK), st, arena, L, (const uint32_t*)w.term_ex, w.tok_end);
}

// Step 2: records and row starts, over n_tokens numbers (read back by the caller).
void bigcol_index_records(const uint8_t* arena, const BigColDesc& d, BigColWork& w, uint32_t n_tokens, hipStream_t st) {
  uint32_t cap = n_tokens + 2;
  AM355_LAUNCH_INDEPENDENT(kb_token_values, grid_for(cap), dim3(BLOCK), st, arena, d, cap, w);
  chain_mark(w.jump_a, cap, w.mark, w.chain_ws, st);  // record headers = orbit of each column's first number
  exclusive_scan_u32(w.mark, w.rec_ex, cap, &w.info->n_records, w.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(kb_records, grid_for(cap), dim3(BLOCK), st, d, cap, w);
  AM355_LAUNCH_INDEPENDENT(kb_record_pairs, grid_for(cap), dim3(BLOCK), st, d, w);
  exclusive_scan_u32(w.rec_rows, w.rec_start, cap, nullptr, w.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(kb_col_rows, dim3(1), dim3(WAVE), st, w);
}

// ---- rows ---------------------------------------------------------------------------------------------------
// coarse table for kb_expand: the record that holds the first row of every stretch of EXPAND_TILE rows, per column (one binary
// search over the column's records per stretch instead of one per row: a row then looks among the few records of its stretch)
__global__ __launch_bounds__(BLOCK) void kb_tile_recs(BigColWork w, BigColVals v, uint32_t n_rows, uint32_t n_succ) {
  const uint32_t tile = gtid();
  const uint32_t c = blockIdx.y;
  const uint32_t n = c >= BC_SUCC_ACTOR ? n_succ : n_rows;
  const uint32_t tiles = n / EXPAND_TILE + 2;
  if (tile >= tiles) return;
  const uint32_t r0 = w.info->r0[c], r1 = w.info->r1[c];
  const uint32_t base = w.rec_start[r0], avail = w.rec_start[r1] - base;
  const uint64_t row = (uint64_t)tile * EXPAND_TILE;
  uint32_t lo = r0, hi = r1;  // last record whose first row is <= row
  if (row >= avail) lo = r1 > r0 ? r1 - 1 : r0;
  else
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (w.rec_start[mid] - base <= row) lo = mid; else hi = mid;
    }
  v.tile_rec[(size_t)c * v.tile_stride + tile] = lo;
}

// one lane per row of one column (blockIdx.y = the column: the twelve columns of a document in ONE launch -- they were twelve, each
// with its own ramp and tail). Delta columns: out = the delta (0 for null) and null_out = 1 for null.
struct ExpandCols {
  uint32_t kind[BIG_NCOL], r0[BIG_NCOL], r1[BIG_NCOL], n[BIG_NCOL];
  uint32_t* out[BIG_NCOL];
  const uint32_t* tile_rec[BIG_NCOL];
  uint8_t* key_ctr_null;
};
__global__ __launch_bounds__(BLOCK) void kb_expand(BigColWork w, ExpandCols cols) {
  const uint32_t c = blockIdx.y;
  const uint32_t kind = cols.kind[c], r0 = cols.r0[c], r1 = cols.r1[c], n_rows = cols.n[c];
  uint32_t* __restrict__ out = cols.out[c];
  uint8_t* __restrict__ null_out = c == BC_KEY_CTR ? cols.key_ctr_null : nullptr;
  const uint32_t* __restrict__ tile_rec = cols.tile_rec[c];
  uint32_t row = gtid();
  if (row >= n_rows) return;
  uint32_t base = w.rec_start[r0];
  uint32_t avail = w.rec_start[r1] - base;
  uint32_t val = kind == BK_DELTA ? 0 : NONE32;
  bool nul = true;
  if (row < avail) {
    // last record whose first row is <= row: between the records of the row's stretch and of the next one
    uint32_t lo = tile_rec[row / EXPAND_TILE], hi = tile_rec[row / EXPAND_TILE + 1] + 1;
    if (hi > r1) hi = r1;
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (w.rec_start[mid] - base <= row) lo = mid; else hi = mid;
    }
  uint32_t t = w.rec_tok[lo], off = row - (w.rec_start[lo] - base);
    if (kind == BK_BOOL) { val = (lo - r0) & 1; nul = false; }
    else {
      int64_t cnt = 0;
      big_tok_sint(w.tok_lo[t], w.tok_hi[t], w.tok_meta[t], cnt);
      if (cnt != 0) {
        uint32_t vt = cnt > 1 ? t + 1 : t + 1 + off;
        uint32_t vlo = w.tok_lo[vt], vhi = w.tok_hi[vt], vmeta = w.tok_meta[vt];
        nul = false;
        if (kind == BK_DELTA) {
          int64_t dv = 0;
          if (!big_tok_sint(vlo, vhi, vmeta, dv)) atomicOr(&w.info->flags, (uint32_t)F_BAD_LEB);
          if (dv > 0x7fffffffll || dv < -0x7fffffffll) { atomicOr(&w.info->flags, (uint32_t)F_OVERFLOW); dv = 0; }
          val = (uint32_t)(int32_t)dv;  // summed modulo 2^32; every partial sum is checked to be a 32-bit counter
        } else {
          uint64_t uv = 0;
          if (!big_tok_uint(vlo, vhi, vmeta, uv)) atomicOr(&w.info->flags, (uint32_t)F_BAD_LEB);
          if (uv >= NONE32) { atomicOr(&w.info->flags, (uint32_t)F_OVERFLOW); uv = 0; }
          val = (uint32_t)uv;
        }
        // a literal never repeats its predecessor (encoding.js:826), and the first value of a record never continues the last
        // value of the record in front of it -- a repetition after a repetition or a literal (encoding.js:866-868), a literal
        // after a repetition (lastValue survives into the literal, :826); a null run in between resets it
        uint32_t pt = NONE32;
        if (off > 0) { if (cnt < 0) pt = vt - 1; }
        else if (lo > r0) {
          uint32_t tp = w.rec_tok[lo - 1];
          int64_t pcnt = 0;
          big_tok_sint(w.tok_lo[tp], w.tok_hi[tp], w.tok_meta[tp], pcnt);
          if (pcnt > 1) pt = tp + 1;
          else if (pcnt < 0) pt = t - 1;
        }
        if (pt != NONE32) {  // (equal NUMBERS: an over-long encoding of the same value is the same value to readRawValue)
          bool same;
          if (kind == BK_DELTA) {
            int64_t a = 0, b = 1;
            same = big_tok_sint(w.tok_lo[pt], w.tok_hi[pt], w.tok_meta[pt], a) && big_tok_sint(vlo, vhi, vmeta, b) && a == b;
          } else {
            uint64_t a = 0, b = 1;
            same = big_tok_uint(w.tok_lo[pt], w.tok_hi[pt], w.tok_meta[pt], a) && big_tok_uint(vlo, vhi, vmeta, b) && a == b;
          }
          if (same) atomicOr(&w.info->flags, (uint32_t)F_BAD_RLE);
        }
      }
    }
  }
  out[row] = val;
  if (null_out) null_out[row] = nul ? 1 : 0;
}

__global__ __launch_bounds__(BLOCK) void kb_shift4(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
  uint32_t i = gtid();
  if (i <= n) out[i] = i < n && in[i] != NONE32 ? in[i] >> 4 : 0;
}
__global__ __launch_bounds__(BLOCK) void kb_null_to_zero(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
  uint32_t i = gtid();
  if (i <= n) out[i] = i < n && in[i] != NONE32 ? in[i] : 0;
}
// absolute value of a delta column: exclusive prefix + own delta
__global__ __launch_bounds__(BLOCK) void kb_delta_abs(const uint32_t* __restrict__ ex, uint32_t n, uint32_t* __restrict__ delta_inout, uint32_t* __restrict__ flags) {
  uint32_t i = gtid();
  if (i >= n) return;
  uint32_t a = ex[i] + delta_inout[i];
  if (a >= 0x80000000u) atomicOr(flags, (uint32_t)F_OVERFLOW);  // negative or beyond 2^31: not an op counter this engine represents
  delta_inout[i] = a;
}

void bigcol_expand(const BigColDesc& d, const BigColWork& w, const BigColInfo& h, BigColVals& v, uint32_t n_rows, uint32_t n_succ_cap, hipStream_t st) {
  {
    const uint32_t big = n_rows > n_succ_cap ? n_rows : n_succ_cap;
    AM355_LAUNCH_INDEPENDENT(kb_tile_recs, dim3((big / EXPAND_TILE + 2 + BLOCK - 1) / BLOCK, BIG_NCOL), dim3(BLOCK), st, w, v, n_rows, n_succ_cap);
  }
  {
    ExpandCols ec{};
    uint32_t n_max = 0;
    for (int c = 0; c < BIG_NCOL; c++) {
      ec.kind[c] = d.kind[c]; ec.r0[c] = h.r0[c]; ec.r1[c] = h.r1[c];
      ec.n[c] = c >= BC_SUCC_ACTOR ? n_succ_cap : n_rows;
      ec.out[c] = v.v[c];
      ec.tile_rec[c] = v.tile_rec + (size_t)c * v.tile_stride;
      n_max = ec.n[c] > n_max ? ec.n[c] : n_max;
    }
    ec.key_ctr_null = v.key_ctr_null;
    if (n_max) AM355_LAUNCH_INDEPENDENT(kb_expand, dim3(grid_for(n_max).x, BIG_NCOL), dim3(BLOCK), st, w, ec);
  }
  // value offsets, succ list offsets
  AM355_LAUNCH_INDEPENDENT(kb_shift4, grid_for(n_rows + 1), dim3(BLOCK), st, (const uint32_t*)v.v[BC_VAL_LEN], n_rows, v.val_off);
  exclusive_scan_u32(v.val_off, v.val_off, n_rows + 1, nullptr, v.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(kb_null_to_zero, grid_for(n_rows + 1), dim3(BLOCK), st, (const uint32_t*)v.v[BC_SUCC_NUM], n_rows, v.succ_first);
  exclusive_scan_u32(v.succ_first, v.succ_first, n_rows + 1, &w.info->n_succ, v.scan_ws, st);
  // delta columns -> absolute counters
  struct { int c; uint32_t n; } deltas[] = {{BC_KEY_CTR, n_rows}, {BC_ID_CTR, n_rows}, {BC_SUCC_CTR, n_succ_cap}};
  for (auto& dl : deltas) {
    if (!dl.n) continue;
    uint32_t* ex = v.tmp;
    exclusive_scan_u32(v.v[dl.c], ex, dl.n, nullptr, v.scan_ws, st);
    AM355_LAUNCH_INDEPENDENT(kb_delta_abs, grid_for(dl.n), dim3(BLOCK), st, (const uint32_t*)ex, dl.n, v.v[dl.c], &w.info->flags);
  }
}

// ---- op rows --------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t big_rank(const uint32_t* __restrict__ rank, uint32_t n_actors, uint32_t idx, uint32_t& err) {
  if (idx >= n_actors) { err |= F_BAD_ROW; return 0; }
  return rank[idx];
}

// same row checks as the lane-serial decoder (am355_decode.hip k_decode_columns; new.js:715-723)
__global__ __launch_bounds__(BLOCK) void kb_assemble(BigColVals v, uint32_t n_rows, const uint32_t* __restrict__ rank, uint32_t n_actors, uint32_t val_raw_abs,
                                                     uint32_t val_raw_len, OpCols o, uint32_t* __restrict__ flags) {
  uint32_t i = gtid();
  if (i >= n_rows) return;
  uint32_t err = 0;
  uint32_t oa = v.v[BC_OBJ_ACTOR][i], oc = v.v[BC_OBJ_CTR][i];
  if ((oa == NONE32) != (oc == NONE32)) err |= F_BAD_ROW;
  o.obj_actor[i] = oa == NONE32 ? NONE32 : big_rank(rank, n_actors, oa, err);
  o.obj_ctr[i] = oc == NONE32 ? 0 : oc;
  uint32_t ka = v.v[BC_KEY_ACTOR][i], kc = v.v[BC_KEY_CTR][i];
  bool kc_null = v.key_ctr_null[i] != 0;
  if ((kc_null && ka != NONE32) || (!kc_null && kc == 0 && ka != NONE32) || (!kc_null && kc > 0 && ka == NONE32)) err |= F_BAD_ROW;
  o.key_actor[i] = ka == NONE32 ? NONE32 : big_rank(rank, n_actors, ka, err);
  o.key_ctr[i] = kc_null ? NONE32 : kc;
  uint32_t ia = v.v[BC_ID_ACTOR][i], ic = v.v[BC_ID_CTR][i];
  if (ia == NONE32 || ic == 0) err |= F_BAD_ROW;
  o.id_actor[i] = ia == NONE32 ? 0 : big_rank(rank, n_actors, ia, err);
  o.id_ctr[i] = ic;
  uint32_t ins = v.v[BC_INSERT][i];
  o.insert[i] = ins == NONE32 ? 0 : (uint8_t)ins;
  uint32_t act = v.v[BC_ACTION][i];
  if (act == NONE32) { err |= F_UNSUPPORTED; act = 0; }
  o.action[i] = act;
  uint32_t tl = v.v[BC_VAL_LEN][i];
  if (tl == NONE32) tl = 0;
  uint32_t used = v.val_off[i];
  if ((uint64_t)used + (tl >> 4) > val_raw_len) err |= F_BAD_CHUNK;  // readRawBytes past the column
  o.val_tl[i] = tl;
  o.val_off[i] = val_raw_abs + used;
  uint32_t sn = v.v[BC_SUCC_NUM][i];
  o.pred_num[i] = sn == NONE32 ? 0 : sn;
  o.pred_first[i] = v.succ_first[i];
  if (err) atomicOr(flags, err);
}

__global__ __launch_bounds__(BLOCK) void kb_assemble_succ(BigColVals v, uint32_t n_succ, const uint32_t* __restrict__ rank, uint32_t n_actors, OpCols o,
                                                          uint32_t* __restrict__ flags) {
  uint32_t i = gtid();
  if (i >= n_succ) return;
  uint32_t err = 0;
  uint32_t a = v.v[BC_SUCC_ACTOR][i];
  if (a == NONE32) { err |= F_UNSUPPORTED; a = 0; }
  o.pred_actor[i] = big_rank(rank, n_actors, a, err);
  o.pred_ctr[i] = v.v[BC_SUCC_CTR][i];
  if (err) atomicOr(flags, err);
}

void bigcol_assemble(const BigColVals& v, uint32_t n_rows, uint32_t n_succ, const uint32_t* actor_rank, uint32_t n_actors, uint32_t val_raw_abs,
                     uint32_t val_raw_len, OpCols o, uint32_t* flags, hipStream_t st) {
  if (n_rows) AM355_LAUNCH_INDEPENDENT(kb_assemble, grid_for(n_rows), dim3(BLOCK), st, v, n_rows, actor_rank, n_actors, val_raw_abs, val_raw_len, o, flags);
  if (n_succ) AM355_LAUNCH_INDEPENDENT(kb_assemble_succ, grid_for(n_succ), dim3(BLOCK), st, v, n_succ, actor_rank, n_actors, o, flags);
}


// ---- key strings ------------------------------------------------------------------------------------------------
// The keyStr column (UTF-8 RLE, encoding.js:789-920 with type 'utf8') holds length-prefixed strings between the record
// headers, so its numbers cannot be found by their terminating bytes. Every byte position is parsed speculatively in both
// roles instead:
//   as a string   [len][bytes]          -> vnext[i]: where the next string of a literal would start
//   as a header   count>1 [len][bytes]  -> hnext[i] = end of the string
//                 count==0 [n]          -> hnext[i] = after n
//                 count<0 (k strings)   -> hnext[i] = k-th vnext-successor of the first string
// 1. k-th successors for all positions at once: pointer doubling over vnext, each position applying round r when bit r of
//    its k is set (jumps commute, so no per-level tables are kept).
// 2. true headers = orbit of position 0 under hnext (pointer doubling, as for the token columns).
// 3. true literal items = vnext-orbits of the first string of every true literal, cut at the next true header.
// 4. items (repetition / null run / literal value) -> prefix sums -> run table (first row, offset, length).
// Position L (one past the column) is the regular end of the parse; NONE32 is "ran off the column".

__device__ __forceinline__ bool key_uleb(const uint8_t* __restrict__ p, uint32_t i, uint32_t L, uint64_t& v, uint32_t& nb) {
  v = 0;
  for (nb = 0; nb < 10 && i + nb < L; nb++) {
    uint32_t b = p[i + nb];
    if (nb == 9 && (b & 0xfe)) return false;
    v |= (uint64_t)(b & 0x7f) << (7 * nb);
    if (!(b & 0x80)) { nb++; return v <= MAX_SAFE_BIG; }
  }
  return false;
}
__device__ __forceinline__ bool key_sleb(const uint8_t* __restrict__ p, uint32_t i, uint32_t L, int64_t& out, uint32_t& nb) {
  uint64_t v = 0;
  for (nb = 0; nb < 10 && i + nb < L; nb++) {
    uint32_t b = p[i + nb];
    if (nb == 9 && b != 0 && b != 0x7f) return false;
    v |= (uint64_t)(b & 0x7f) << (7 * nb);
    if (!(b & 0x80)) {
      nb++;
      if ((b & 0x40) && 7 * nb < 64) v |= ~0ull << (7 * nb);
      out = (int64_t)v;
      return out <= (int64_t)MAX_SAFE_BIG && out >= -(int64_t)MAX_SAFE_BIG;
    }
  }
  return false;
}
// where the string starting at i ends (NONE32: malformed / beyond the column)
__device__ __forceinline__ uint32_t key_string_end(const uint8_t* __restrict__ p, uint32_t i, uint32_t L, uint32_t* off, uint32_t* len) {
  uint64_t v;
  uint32_t nb;
  if (i >= L || !key_uleb(p, i, L, v, nb) || v > (uint64_t)(L - i - nb)) return NONE32;
  if (off) { *off = i + nb; *len = (uint32_t)v; }
  return i + nb + (uint32_t)v;
}

// ---- 1. vnext / hnext of every position, k-th successors included, tile by tile in LDS ------------------------------------------
// One workgroup takes KEY_TILE consecutive positions. vnext chains only move forward, so inside the tile they are followed by pointer
// doubling over 16-bit LOCAL indexes: after round r, P[i] = the 2^r-th successor of i, or EXIT once the chain has left the tile, in
// which case H[i] = the number of hops that took and E[i] = the position (anywhere behind the tile) it left to. A literal header at i
// with k strings walks the same tables -- it applies round r when bit r of k is set; jumps along one chain commute -- and either
// stays inside the tile (hnext = where it stands after the last round) or leaves it, which happens exactly when k >= H[first
// string]: then hnext / rem hold the CONTINUATION (E[first string], k - H[first string]) that kk_kth_cont finishes from the
// per-position (hops to the tile's end, exit position) pairs every tile leaves in global memory.
// (Before: twelve rounds over ALL positions in global memory, three 4-byte arrays read and two written per round and position --
// 3.7 ms of the 34 MB key column of the config-5 document, plus a host decision in the middle for literals beyond 4096 strings.)
// A workgroup OWNS `tile` positions and follows chains through a WINDOW of `win` = tile + halo positions: the strings of the garbage
// "literals" that most byte positions are when read as headers (a letter is a count of a few dozen) span a few hundred bytes, so with
// a halo of a thousand positions only real, long literals leave the window and need the walker -- with the window equal to the tile
// every twentieth position did, and the walker (dependent loads, a lane at a time) took 2.2 ms of the config-5 key column.
constexpr uint32_t KEY_WIN = 4096;            // table entries (LDS: 16 bytes each)
constexpr uint32_t KEY_TILE = 3072;           // positions a workgroup owns by default; the rest of the window is halo
constexpr uint32_t KEY_TILE_THREADS = 1024;
constexpr uint32_t KEY_PER = KEY_WIN / KEY_TILE_THREADS;
constexpr uint32_t KEY_EXIT = 0xffffu;

__global__ __launch_bounds__(KEY_TILE_THREADS) void kk_tile(const uint8_t* __restrict__ col, uint32_t L, KeyWork k, uint32_t tile, uint32_t win,
                                                            uint32_t* __restrict__ pend, uint32_t* __restrict__ n_pend) {
  __shared__ uint16_t P[2][KEY_WIN], H[2][KEY_WIN];
  __shared__ uint32_t E[2][KEY_WIN];
  const uint32_t base = blockIdx.x * tile, t = threadIdx.x;
  const uint32_t end = base + win;   // (positions >= L + 2 do not exist; L and L + 1 are the two end markers)
  uint32_t kc_[KEY_PER], cur_[KEY_PER], hn_[KEY_PER], s0_[KEY_PER];
  // ---- parse: every position of the window as a string, the owned ones as a header too ----
#pragma unroll
  for (uint32_t j = 0; j < KEY_PER; j++) {
    const uint32_t l = t + j * KEY_TILE_THREADS, i = base + l;
    kc_[j] = 0; cur_[j] = KEY_EXIT; hn_[j] = NONE32; s0_[j] = NONE32;
    if (l >= win) continue;
    uint32_t vn = NONE32;
    if (i < L) vn = key_string_end(col, i, L, nullptr, nullptr);
    if (l < tile) {
      if (i < L) {
        int64_t cnt;
        uint32_t hb;
        if (key_sleb(col, i, L, cnt, hb)) {
          const uint32_t q = i + hb;
          if (cnt > 1) hn_[j] = key_string_end(col, q, L, nullptr, nullptr);
          else if (cnt == 0) {
            uint64_t n;
            uint32_t nb;
            if (key_uleb(col, q, L, n, nb) && n > 0) hn_[j] = q + nb;
          } else if (cnt < 0 && (uint64_t)(-cnt) <= (uint64_t)L && q < L) {
            kc_[j] = (uint32_t)(-cnt);
            s0_[j] = q;   // the first string: advanced kc times below
          }
        }
      }
      if (i <= L + 1) {
        k.vnext[i] = vn;
        k.mark_h[i] = (i == 0 && L > 0) ? 1u : 0u;
        k.mark_v[i] = 0;
        k.rows[i] = 0;
        k.kk[i] = kc_[j];
      }
    }
    // local successor: inside the window, or EXIT to vn (NONE32: a dead end -- malformed, or beyond the column)
    const bool inside = vn != NONE32 && vn < end && vn < L;
    P[0][l] = inside ? (uint16_t)(vn - base) : (uint16_t)KEY_EXIT;
    H[0][l] = 1;
    E[0][l] = vn;
  }
  __syncthreads();
  // a literal whose first string lies behind the window, or which holds more strings than the window positions, leaves it anyway
#pragma unroll
  for (uint32_t j = 0; j < KEY_PER; j++)
    if (kc_[j] && s0_[j] < end && kc_[j] < win) cur_[j] = s0_[j] - base;
  int pp = 0;
  for (uint32_t r = 0; (1u << r) < win * 2 && r < 13; r++) {
    // the header walks first (they read table r), then the table doubles
#pragma unroll
    for (uint32_t j = 0; j < KEY_PER; j++)
      if (cur_[j] != KEY_EXIT && (kc_[j] >> r & 1u)) cur_[j] = P[pp][cur_[j]];
#pragma unroll
    for (uint32_t j = 0; j < KEY_PER; j++) {
      const uint32_t l = t + j * KEY_TILE_THREADS;
      if (l >= win) continue;
      uint16_t p = P[pp][l], h = H[pp][l];
      uint32_t e = E[pp][l];
      if (p != KEY_EXIT) {
        const uint16_t p2 = P[pp][p];
        h = (uint16_t)(h + H[pp][p]);
        if (p2 == KEY_EXIT) e = E[pp][p];
        p = p2;
      }
      P[pp ^ 1][l] = p; H[pp ^ 1][l] = h; E[pp ^ 1][l] = e;
    }
    __syncthreads();
    pp ^= 1;
  }
  // ---- results for the owned positions: (hops to the window's end, exit position); hnext, or the continuation of a literal that left ----
#pragma unroll
  for (uint32_t j = 0; j < KEY_PER; j++) {
    const uint32_t l = t + j * KEY_TILE_THREADS, i = base + l;
    if (l >= tile || i > L + 1) continue;
    k.item_ex[i] = H[pp][l];   // (free until the item flags are built)
    k.ja[i] = E[pp][l];        // (kk_item_init writes its own jumps there afterwards)
    uint32_t hn = hn_[j], rem = 0;
    if (kc_[j]) {
      if (cur_[j] != KEY_EXIT) hn = base + cur_[j];   // all kc hops inside the window
      else if (s0_[j] >= end) { hn = s0_[j]; rem = kc_[j]; }   // the first string lies behind the window: everything is left to do
      else {
        const uint32_t sl = s0_[j] - base, hops = H[pp][sl];
        // kc >= hops (else the walk had stayed inside): leave the window with the rest
        hn = E[pp][sl];
        rem = kc_[j] >= hops ? kc_[j] - hops : 0u;
      }
    }
    k.hnext[i] = hn;
    k.jb[i] = rem;
    if (rem) pend[atomicAdd(n_pend, 1u)] = i;
  }
}

// literals that left their window: window by window along (hops to the end, exit position), then the last hops -- fewer than the
// window's -- one by one: up to KEY_CONT_SMALL of them right here (the garbage "literals", a few dozen strings), more than that
// (real literals: hundreds of strings in their last window) by kk_kth_big from a copy of the window in LDS
constexpr uint32_t KEY_CONT_SMALL = 48;
__global__ __launch_bounds__(BLOCK) void kk_kth_cont(uint32_t L, KeyWork k, const uint32_t* __restrict__ pend, const uint32_t* __restrict__ n_pend,
                                                     uint32_t* __restrict__ big, uint32_t* __restrict__ n_big, uint32_t small, uint32_t max_jumps) {
  const uint32_t total = *n_pend;
  for (uint32_t w = gtid(); w < total; w += gridDim.x * BLOCK) {
    const uint32_t i = pend[w];
    uint32_t rem = k.jb[i], pos = k.hnext[i], jumps = 0;
    bool cut = false;
    while (rem) {
      if (pos == NONE32 || pos >= L) { pos = NONE32; rem = 0; break; }   // (L is the regular end of the parse; nothing follows it)
      const uint32_t hops = k.item_ex[pos];
      if (rem >= hops) {
        // Garbage "literals" of millions of strings exist (any bytes that read as a long negative count) and would walk window by
        // window to the end of the column -- two dependent loads per window, 3.3 ms for a lane of the config-5 column, and the kernel
        // waits for its slowest lane. The walk is cut after max_jumps windows: the header becomes a dead end with its rest kept in jb.
        // Should the TRUE parse reach such a header (kk_item_init sees it), the load is repeated without the bound (am355_replay.hip).
        if (max_jumps && ++jumps > max_jumps) { cut = true; break; }
        rem -= hops; pos = k.ja[pos];
      }
      else if (rem <= small) {
        for (; rem; rem--) {
          pos = k.vnext[pos];
          if (pos == NONE32) break;
        }
        rem = 0;
      } else break;   // a long last stretch: a chain of dependent loads from global memory would hold this kernel up (3.3 ms for the config-5 column)
    }
    k.hnext[i] = cut ? NONE32 : pos;
    k.jb[i] = rem;
    if (rem && !cut) big[atomicAdd(n_big, 1u)] = i;
  }
}

// one wavefront per literal with a long last stretch: the rest of its window comes to LDS in one round of loads, lane 0 walks there
__global__ __launch_bounds__(WAVE) void kk_kth_big(const uint8_t* __restrict__ col, uint32_t L, KeyWork k, uint32_t tile, uint32_t win,
                                                   const uint32_t* __restrict__ big, const uint32_t* __restrict__ n_big) {
  __shared__ uint8_t bytes[KEY_WIN + 64];
  const uint32_t total = *n_big, lane = threadIdx.x;
  for (uint32_t w = blockIdx.x; w < total; w += gridDim.x) {
    const uint32_t i = big[w];
    uint32_t pos = k.hnext[i], rem = k.jb[i];
    // the next `rem` strings start inside the window of pos (rem < its hops to the window's end); a length prefix may reach 10 bytes past it
    const uint32_t wend0 = (pos / tile) * tile + win + 16, wend = wend0 < L ? wend0 : L;
    __syncthreads();
    for (uint32_t q = lane; pos + q < wend; q += WAVE) bytes[q] = col[pos + q];
    __syncthreads();
    if (lane == 0) {
      const uint8_t* lp = bytes - pos;   // (indexed by column position)
      uint32_t p = pos;
      for (; rem; rem--) {
        p = key_string_end(lp, p, wend, nullptr, nullptr);
        if (p == NONE32) break;   // (cannot happen: the stretch stays inside the window by construction; kept as a dead end)
      }
      k.hnext[i] = p;
      k.jb[i] = 0;
    }
  }
}

// literal items: start marks on the first string of every true literal; jumps follow vnext but stop at a true header
__global__ __launch_bounds__(BLOCK) void kk_item_init(const uint8_t* __restrict__ col, uint32_t L, KeyWork k, uint32_t* __restrict__ flags) {
  uint32_t i = gtid();
  if (i >= L) { if (i <= L + 1) k.ja[i] = NONE32; return; }
  uint32_t vn = k.vnext[i];
  k.ja[i] = (vn < L && !k.mark_h[vn]) ? vn : NONE32;
  if (k.mark_h[i]) {
    if (k.jb[i]) { flags[3] = 1; }  // a true literal the walker cut off (kk_kth_cont): the caller repeats the load without the bound
    else if (k.hnext[i] == NONE32 || k.hnext[i] > L) atomicOr(flags, (uint32_t)F_BAD_LEB);  // header that is not a well-formed record, or runs off the column
    if (k.kk[i]) {
      int64_t cnt;
      uint32_t hb;
      key_sleb(col, i, L, cnt, hb);
      k.mark_v[i + hb] = 2;  // 2 = first value of its literal
    }
  }
}

// 1 for every run of the final table: repetitions and null runs sit on their header, literal values on their string
// (four positions per thread, as 16-byte loads and one 16-byte store where a whole stretch lies inside the column: one position per
// thread was half a million wavefronts of three loads each for a 34 MB column -- 1.6 ms, bound by launching them)
__global__ __launch_bounds__(BLOCK) void kk_item_flags(uint32_t L, KeyWork k) {
  const uint32_t i0 = gtid() * 4;
  if (i0 > L + 1) return;
  const bool wide = i0 + 4 <= L && ((((uintptr_t)k.mark_h | (uintptr_t)k.kk | (uintptr_t)k.mark_v | (uintptr_t)k.item_ex) & 15) == 0);
  if (wide) {
    const uint4 h = *(const uint4*)(k.mark_h + i0), q = *(const uint4*)(k.kk + i0), v = *(const uint4*)(k.mark_v + i0);
    *(uint4*)(k.item_ex + i0) = uint4{(h.x && !q.x) || v.x ? 1u : 0u, (h.y && !q.y) || v.y ? 1u : 0u, (h.z && !q.z) || v.z ? 1u : 0u,
                                      (h.w && !q.w) || v.w ? 1u : 0u};
    return;
  }
  for (uint32_t i = i0; i < i0 + 4 && i <= L + 1; i++) {
    uint32_t f = 0;
    if (i < L) f = (k.mark_h[i] && !k.kk[i]) || k.mark_v[i] ? 1u : 0u;
    k.item_ex[i] = f;
  }
}

__global__ __launch_bounds__(BLOCK) void kk_items(const uint8_t* __restrict__ col, uint32_t col_abs, uint32_t L, KeyWork k, uint32_t* __restrict__ flags) {
  uint32_t i = gtid();
  if (i >= L) return;
  bool is_h = k.mark_h[i] && !k.kk[i], is_v = k.mark_v[i] != 0;
  if (!is_h && !is_v) return;
  uint32_t idx = k.item_ex[i];
  uint32_t off = 0, len = NONE32, rows = 1;
  if (is_v) key_string_end(col, i, L, &off, &len);
  else {
    int64_t cnt = 0;
    uint32_t hb = 0;
    key_sleb(col, i, L, cnt, hb);
    if (cnt > 1) {
      key_string_end(col, i + hb, L, &off, &len);
      rows = cnt > 0x7ffffff0ll ? 0 : (uint32_t)cnt;
      if (!rows) atomicOr(flags, (uint32_t)F_OVERFLOW);
    } else {
      uint64_t n = 0;
      uint32_t nb;
      key_uleb(col, i + hb, L, n, nb);
      rows = n > 0x7ffffff0ull ? 0 : (uint32_t)n;
      if (!rows) atomicOr(flags, (uint32_t)F_OVERFLOW);
    }
  }
  k.rows[idx] = rows;
  k.run_off[idx] = len == NONE32 ? 0 : col_abs + off;
  k.run_len[idx] = len;
  k.run_kind[idx] = is_v ? k.mark_v[i] : 0;  // 0 record on its own, 1 later literal value, 2 first literal value
  // canonical form (encoding.js:865-887): no two null runs in a row, no value equal to its predecessor -- checked
  // between neighbouring runs by kk_run_pairs once the table is complete
}

__global__ __launch_bounds__(BLOCK) void kk_run_pairs(const uint8_t* __restrict__ arena, KeyWork k, uint32_t* __restrict__ flags) {
  uint32_t r = gtid();
  uint32_t n = *k.n_runs;
  if (r == 0 || r >= n) return;
  uint32_t la = k.run_len[r - 1], lb = k.run_len[r];
  if (k.run_kind[r] == 2 && k.run_kind[r - 1] != 0) { atomicOr(flags, (uint32_t)F_BAD_RLE); return; }  // two literals in a row
  if (la == NONE32 && lb == NONE32) { atomicOr(flags, (uint32_t)F_BAD_RLE); return; }
  if (la != lb || la == NONE32) return;
  const uint8_t *a = arena + k.run_off[r - 1], *b = arena + k.run_off[r];
  for (uint32_t j = 0; j < la; j++)
    if (a[j] != b[j]) return;
  atomicOr(flags, (uint32_t)F_BAD_RLE);
}

size_t keystr_work_bytes(uint32_t col_len) {
  size_t cap = (size_t)col_len + 2;
  return 13 * al256(4 * cap) + al256(scan_workspace_bytes((uint32_t)cap)) + al256(chain_work_bytes((uint32_t)cap)) + 256;
}

void keystr_index_begin(const uint8_t* arena, uint32_t col_abs, uint32_t col_len, void* work, KeyStage& s, uint32_t* n_runs, uint32_t* d_unresolved,
                        hipStream_t st) {
  uint32_t L = col_len, cap = L + 2;
  KeyWork& k = s.k;
  uint8_t* p = (uint8_t*)work;
  canary_scope("document key strings (keystr_index_begin)");
  auto take = [&](size_t bytes) { void* r = p; canary_note(p, bytes); p += al256(bytes); return r; };
  uint32_t** arrs[] = {&k.vnext, &k.hnext, &k.kk, &k.ja, &k.jb, &k.mark_h, &k.mark_v, &k.item_ex, &k.rows, &k.run_start, &k.run_off, &k.run_len, &k.run_kind};
  for (uint32_t** a : arrs) *a = (uint32_t*)take(4 * (size_t)cap);
  k.scan_ws = take(scan_workspace_bytes(cap));
  s.chain_ws = take(chain_work_bytes(cap));
  canary_arm();
  k.n_runs = n_runs;
  s.col = arena + col_abs;
  s.arena = arena;
  s.col_abs = col_abs;
  s.L = L;
  // (d_unresolved[0], [1]: two device words the caller cleared, count the literals that leave their window and those of them with a
  // long last stretch; their positions go to run_start / run_off, which are free until the run table is built)
  uint32_t tile = KEY_TILE;
  if (const char* e = getenv("AM355_KEY_TILE")) { int v = atoi(e); if (v >= 16 && (uint32_t)v <= KEY_TILE) tile = (uint32_t)v; }  // (tests: small tiles, so that small documents cross them)
  const uint32_t win = tile + tile / 3 <= KEY_WIN ? tile + tile / 3 : KEY_WIN;
  uint32_t cont_small = KEY_CONT_SMALL;
  if (const char* e = getenv("AM355_KEY_CONT_SMALL")) cont_small = (uint32_t)atoi(e);   // (tests: 0 sends every last stretch through kk_kth_big)
  if (s.max_jumps) { if (const char* e = getenv("AM355_KEY_JUMPS")) s.max_jumps = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : s.max_jumps; }  // (tests: 1 cuts every long literal)
  hipLaunchKernelGGL(kk_tile, dim3((cap + tile - 1) / tile), dim3(KEY_TILE_THREADS), 0, st, s.col, L, k, tile, win, k.run_start, d_unresolved);
  // (a lane per pending literal, all lanes of a wavefront busy: the walk is a chain of dependent loads, and what hides it is the number
  // of wavefronts in flight -- 512 workgroups striding over the list took 3.3 ms for the config-5 column, a lane per POSITION with one
  // lane in twenty-five busy 2.2 ms; the grid covers a literal for every eighth position and strides beyond that)
  AM355_LAUNCH_INDEPENDENT(kk_kth_cont, dim3((cap / 8 + BLOCK - 1) / BLOCK + 1), dim3(BLOCK), st, L, k, (const uint32_t*)k.run_start, (const uint32_t*)d_unresolved, k.run_off,
                           d_unresolved + 1, cont_small, s.max_jumps);
  hipLaunchKernelGGL(kk_kth_big, dim3(2048), dim3(WAVE), 0, st, s.col, L, k, tile, win, (const uint32_t*)k.run_off, (const uint32_t*)(d_unresolved + 1));
  if (getenv("AM355_KEY_DIAG")) {  // (diagnostic: how many literals left their window / had a long last stretch)
    uint32_t w[2] = {0, 0};
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(w, d_unresolved, 8, hipMemcpyDeviceToHost);
    fprintf(stderr, "keystr: %u positions, tile %u window %u: %u literals left their window, %u with a long last stretch\n", L, tile, win, w[0], w[1]);
  }
  chain_mark(k.hnext, L, k.mark_h, s.chain_ws, st);   // 2. true headers
}

void keystr_index_finish(KeyStage& s, bool unresolved, uint32_t** run_start, uint32_t** run_off, uint32_t** run_len, uint32_t* flags, hipStream_t st) {
  KeyWork& k = s.k;
  uint32_t L = s.L, cap = L + 2;
  (void)unresolved;
  *run_start = k.run_start; *run_off = k.run_off; *run_len = k.run_len;
  AM355_LAUNCH_INDEPENDENT(kk_item_init, grid_for(cap), dim3(BLOCK), st, s.col, L, k, flags);
  chain_mark(k.ja, L, k.mark_v, s.chain_ws, st);     // 3. literal items
  AM355_LAUNCH_INDEPENDENT(kk_item_flags, grid_for((cap + 3) / 4), dim3(BLOCK), st, L, k);
  exclusive_scan_u32(k.item_ex, k.item_ex, cap, k.n_runs, k.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(kk_items, grid_for(L), dim3(BLOCK), st, s.col, s.col_abs, L, k, flags);
  exclusive_scan_u32(k.rows, k.run_start, cap, nullptr, k.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(kk_run_pairs, grid_for(cap), dim3(BLOCK), st, s.arena, k, flags);
}

}  // namespace am355
