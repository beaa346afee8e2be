// Device-side column encoders for Backend.save: the reference's RLE / delta / boolean encoders
// (backend/encoding.js:558-783, 932-948, 1061-1135) restated as data-parallel passes.
//
// The reference encoders are state machines fed one value at a time, but what they emit is a function of the value
// sequence alone: a maximal run of nulls becomes (0, n); a maximal run of >= 2 equal values (n, v); every maximal stretch
// of lone values one literal (-k, v1..vk); a column of nothing but nulls is empty. So: run boundaries by comparing
// neighbours, run ids / lengths by a prefix sum, literal groups by a second one, byte sizes per run, a third prefix
// sum for the offsets, and one lane per run writes its header and value. HBM-bound streaming work; no MFMA.
#include "am355_encode.h"
#include "am355_prims.h"

namespace am355 {

static inline dim3 grid_for(uint32_t n) { return dim3((n + BLOCK - 1) / BLOCK); }
static size_t al256(size_t b) { return carve_round(b); }

size_t enc_work_bytes(uint32_t n) { return 8 * al256(4 * ((size_t)n + 2)) + al256(scan_workspace_bytes(n + 2)); }
void enc_carve(EncWork& w, void* base, uint32_t n) {
  canary_scope("column encoder work (enc_carve)");
  uint8_t* p = (uint8_t*)base;
  auto take = [&](size_t bytes) { void* r = p; canary_note(p, bytes); p += al256(bytes); return r; };
  uint32_t** arrs[] = {&w.flag, &w.run_ex, &w.run_first, &w.grp_flag, &w.grp_ex, &w.grp_first, &w.size, &w.off_ex};
  for (uint32_t** a : arrs) *a = (uint32_t*)take(4 * ((size_t)n + 2));
  w.scan_ws = take(scan_workspace_bytes(n + 2));
}

__device__ __forceinline__ uint32_t uleb_size(uint64_t v) {
  uint32_t n = 1;
  while (v >= 0x80) { v >>= 7; n++; }
  return n;
}
__device__ __forceinline__ uint32_t sleb_size(int64_t v) {
  uint32_t n = 1;
  while (!((v >= -64) && (v < 64))) { v >>= 7; n++; }
  return n;
}
__device__ __forceinline__ uint8_t* put_uleb(uint8_t* p, uint64_t v) {
  while (v >= 0x80) { *p++ = (uint8_t)(v | 0x80); v >>= 7; }
  *p++ = (uint8_t)v;
  return p;
}
__device__ __forceinline__ uint8_t* put_sleb(uint8_t* p, int64_t v) {
  for (;;) {
    uint8_t b = (uint8_t)(v & 0x7f);
    v >>= 7;  // arithmetic
    if ((v == 0 && !(b & 0x40)) || (v == -1 && (b & 0x40))) { *p++ = b; return p; }
    *p++ = b | 0x80;
  }
}

// value sources -------------------------------------------------------------------------------------------------
// `seg` (all sources; may be null = one segment): seg[i] = index of the first value of the segment value i belongs to. The history
// reconstruction (am355_hist.hip) encodes the columns of ALL changes of a document in one go: a segment is one change, and what
// the reference's encoder emits for a change is what it emits for the change's values alone -- runs, literal stretches, null runs
// and delta chains end at a segment's end.
struct NumSrc {
  const uint32_t* vals;
  const uint8_t* mask;
  bool is_signed;
  const uint32_t* seg;
  __device__ __forceinline__ bool seg_start(uint32_t i) const { return i == 0 || (seg && seg[i] == i); }
  __device__ __forceinline__ bool is_null(uint32_t i) const { return mask ? mask[i] != 0 : vals[i] == NONE32; }
  __device__ __forceinline__ bool equal(uint32_t i, uint32_t j) const {
    bool ni = is_null(i), nj = is_null(j);
    return ni == nj && (ni || vals[i] == vals[j]);
  }
  __device__ __forceinline__ uint32_t vsize(uint32_t i) const { return is_signed ? sleb_size((int64_t)(int32_t)vals[i]) : uleb_size(vals[i]); }
  __device__ __forceinline__ uint8_t* write(uint8_t* p, uint32_t i) const { return is_signed ? put_sleb(p, (int64_t)(int32_t)vals[i]) : put_uleb(p, vals[i]); }
};
struct StrSrc {
  const uint8_t* arena;
  const uint32_t *off, *len;
  const uint32_t* seg;
  __device__ __forceinline__ bool seg_start(uint32_t i) const { return i == 0 || (seg && seg[i] == i); }
  __device__ __forceinline__ bool is_null(uint32_t i) const { return len[i] == NONE32; }
  __device__ __forceinline__ bool equal(uint32_t i, uint32_t j) const {
    uint32_t li = len[i], lj = len[j];
    if (li != lj) return false;
    if (li == NONE32 || off[i] == off[j]) return true;
    const uint8_t *a = arena + off[i], *b = arena + off[j];
    for (uint32_t k = 0; k < li; k++)
      if (a[k] != b[k]) return false;
    return true;
  }
  __device__ __forceinline__ uint32_t vsize(uint32_t i) const { return uleb_size(len[i]) + len[i]; }
  __device__ __forceinline__ uint8_t* write(uint8_t* p, uint32_t i) const {
    p = put_uleb(p, len[i]);
    const uint8_t* a = arena + off[i];
    for (uint32_t k = 0; k < len[i]; k++) p[k] = a[k];
    return p + len[i];
  }
};

// passes ----------------------------------------------------------------------------------------------------------
template <class Src>
__global__ __launch_bounds__(BLOCK) void ke_run_flags(Src s, uint32_t n, uint32_t* __restrict__ flag) {
  uint32_t i = gtid();
  if (i > n) return;
  flag[i] = i < n && (s.seg_start(i) || !s.equal(i, i - 1)) ? 1u : 0u;
}

// run_first[r] = first value of run r; run_first[R] = n
__global__ __launch_bounds__(BLOCK) void ke_run_first(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ run_ex, uint32_t n,
                                                      uint32_t* __restrict__ run_first) {
  uint32_t i = gtid();
  if (i > n) return;
  if (i == n) run_first[run_ex[n]] = n;
  else if (flag[i]) run_first[run_ex[i]] = i;
}

// a record starts at every run that is not a lone value, and at the first of a stretch of lone values
template <class Src>
__global__ __launch_bounds__(BLOCK) void ke_group_flags(Src s, uint32_t n, const uint32_t* __restrict__ run_ex, const uint32_t* __restrict__ run_first,
                                                        uint32_t* __restrict__ grp_flag) {
  uint32_t r = gtid();
  if (r > n) return;
  uint32_t R = run_ex[n];
  uint32_t f = 0;
  if (r < R) {
    uint32_t a = run_first[r];
    bool lone = run_first[r + 1] - a == 1 && !s.is_null(a);
    bool prev_lone = false;
    if (r > 0 && !s.seg_start(a)) { uint32_t b = run_first[r - 1]; prev_lone = a - b == 1 && !s.is_null(b); }
    f = (!lone || !prev_lone) ? 1u : 0u;
  }
  grp_flag[r] = f;
}

__global__ __launch_bounds__(BLOCK) void ke_group_first(const uint32_t* __restrict__ grp_flag, const uint32_t* __restrict__ grp_ex, const uint32_t* __restrict__ run_ex,
                                                        uint32_t n, uint32_t* __restrict__ grp_first) {
  uint32_t r = gtid();
  if (r > n) return;
  uint32_t R = run_ex[n];
  if (r == R) grp_first[grp_ex[R]] = R;
  else if (r < R && grp_flag[r]) grp_first[grp_ex[r]] = r;
}

template <class Src>
__global__ __launch_bounds__(BLOCK) void ke_run_sizes(Src s, uint32_t n, const uint32_t* __restrict__ run_ex, const uint32_t* __restrict__ run_first,
                                                      const uint32_t* __restrict__ grp_flag, const uint32_t* __restrict__ grp_ex,
                                                      const uint32_t* __restrict__ grp_first, uint32_t* __restrict__ size) {
  uint32_t r = gtid();
  if (r > n) return;
  uint32_t R = run_ex[n];
  uint32_t sz = 0;
  if (r < R) {
    uint32_t a = run_first[r], len = run_first[r + 1] - a;
    // a column (segment) of nothing but nulls is empty
    const bool whole = s.seg ? (s.seg_start(a) && (a + len == n || s.seg_start(a + len))) : R == 1;
    if (s.is_null(a)) sz = whole ? 0 : 1 + uleb_size(len);
    else if (len >= 2) sz = sleb_size((int64_t)len) + s.vsize(a);
    else {
      sz = s.vsize(a);
      if (grp_flag[r]) { uint32_t g = grp_ex[r]; sz += sleb_size(-(int64_t)(grp_first[g + 1] - grp_first[g])); }
    }
  }
  size[r] = sz;
}

template <class Src>
__global__ __launch_bounds__(BLOCK) void ke_run_write(Src s, uint32_t n, const uint32_t* __restrict__ run_ex, const uint32_t* __restrict__ run_first,
                                                      const uint32_t* __restrict__ grp_flag, const uint32_t* __restrict__ grp_ex,
                                                      const uint32_t* __restrict__ grp_first, const uint32_t* __restrict__ off_ex, uint8_t* __restrict__ out,
                                                      uint32_t cap) {
  uint32_t r = gtid();
  if (r >= n) return;
  uint32_t R = run_ex[n];
  if (r >= R) return;
  if (off_ex[r + 1] > cap) return;  // never past the caller's carve-out: the caller sees the total (d_len) and sizes again or refuses
  uint32_t a = run_first[r], len = run_first[r + 1] - a;
  uint8_t* p = out + off_ex[r];
  if (s.is_null(a)) {
    const bool whole = s.seg ? (s.seg_start(a) && (a + len == n || s.seg_start(a + len))) : R == 1;
    if (whole) return;
    *p++ = 0;
    put_uleb(p, len);
  } else if (len >= 2) {
    p = put_sleb(p, (int64_t)len);
    s.write(p, a);
  } else {
    if (grp_flag[r]) { uint32_t g = grp_ex[r]; p = put_sleb(p, -(int64_t)(grp_first[g + 1] - grp_first[g])); }
    s.write(p, a);
  }
}

template <class Src>
static void enc_rle(Src s, uint32_t n, EncWork& w, uint8_t* out, uint32_t* d_len, hipStream_t st, uint32_t cap) {
  if (!n) { (void)hipMemsetAsync(d_len, 0, 4, st); return; }
  AM355_LAUNCH_INDEPENDENT(ke_run_flags<Src>, grid_for(n + 1), dim3(BLOCK), st, s, n, w.flag);
  exclusive_scan_u32(w.flag, w.run_ex, n + 1, nullptr, w.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(ke_run_first, grid_for(n + 1), dim3(BLOCK), st, (const uint32_t*)w.flag, (const uint32_t*)w.run_ex, n, w.run_first);
  AM355_LAUNCH_INDEPENDENT(ke_group_flags<Src>, grid_for(n + 1), dim3(BLOCK), st, s, n, (const uint32_t*)w.run_ex, (const uint32_t*)w.run_first, w.grp_flag);
  exclusive_scan_u32(w.grp_flag, w.grp_ex, n + 1, nullptr, w.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(ke_group_first, grid_for(n + 1), dim3(BLOCK), st, (const uint32_t*)w.grp_flag, (const uint32_t*)w.grp_ex, (const uint32_t*)w.run_ex, n,
                           w.grp_first);
  AM355_LAUNCH_INDEPENDENT(ke_run_sizes<Src>, grid_for(n + 1), dim3(BLOCK), st, s, n, (const uint32_t*)w.run_ex, (const uint32_t*)w.run_first,
                           (const uint32_t*)w.grp_flag, (const uint32_t*)w.grp_ex, (const uint32_t*)w.grp_first, w.size);
  exclusive_scan_u32(w.size, w.off_ex, n + 1, d_len, w.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(ke_run_write<Src>, grid_for(n), dim3(BLOCK), st, s, n, (const uint32_t*)w.run_ex, (const uint32_t*)w.run_first,
                           (const uint32_t*)w.grp_flag, (const uint32_t*)w.grp_ex, (const uint32_t*)w.grp_first, (const uint32_t*)w.off_ex, out, cap);
}

void enc_rle_numbers(const uint32_t* vals, const uint8_t* nullmask, uint32_t n, bool is_signed, EncWork& w, uint8_t* out, uint32_t* d_len, hipStream_t st,
                     const uint32_t* seg, uint32_t cap) {
  enc_rle(NumSrc{vals, nullmask, is_signed, seg}, n, w, out, d_len, st, cap);
}
void enc_rle_strings(const uint8_t* arena, const uint32_t* off, const uint32_t* len, uint32_t n, EncWork& w, uint8_t* out, uint32_t* d_len, hipStream_t st,
                     const uint32_t* seg, uint32_t cap) {
  enc_rle(StrSrc{arena, off, len, seg}, n, w, out, d_len, st, cap);
}

// byte offset at which every segment's bytes begin in the output of the LAST run-length encode done with `w` (seg_base[k] = index of
// segment k's first value, seg_base[n_seg] = n): seg_off[k], k = 0 .. n_seg
__global__ __launch_bounds__(BLOCK) void ke_seg_offsets(const uint32_t* __restrict__ seg_base, uint32_t n_seg, const uint32_t* __restrict__ run_ex,
                                                        const uint32_t* __restrict__ off_ex, uint32_t* __restrict__ seg_off) {
  uint32_t k = gtid();
  if (k > n_seg) return;
  seg_off[k] = off_ex[run_ex[seg_base[k]]];
}
void enc_segment_offsets(const uint32_t* seg_base, uint32_t n_seg, const EncWork& w, uint32_t* seg_off, hipStream_t st) {
  AM355_LAUNCH_INDEPENDENT(ke_seg_offsets, grid_for(n_seg + 1), dim3(BLOCK), st, seg_base, n_seg, (const uint32_t*)w.run_ex, (const uint32_t*)w.off_ex, seg_off);
}
__global__ __launch_bounds__(BLOCK) void ke_seg_offsets_raw(const uint32_t* __restrict__ seg_base, uint32_t n_seg, const uint32_t* __restrict__ off_ex,
                                                            uint32_t* __restrict__ seg_off) {
  uint32_t k = gtid();
  if (k > n_seg) return;
  seg_off[k] = off_ex[seg_base[k]];
}
void enc_segment_offsets_raw(const uint32_t* seg_base, uint32_t n_seg, const EncWork& w, uint32_t* seg_off, hipStream_t st) {
  AM355_LAUNCH_INDEPENDENT(ke_seg_offsets_raw, grid_for(n_seg + 1), dim3(BLOCK), st, seg_base, n_seg, (const uint32_t*)w.off_ex, seg_off);
}

// delta ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void ke_nonnull_flags(const uint32_t* __restrict__ vals, uint32_t n, uint32_t* __restrict__ flag) {
  uint32_t i = gtid();
  if (i > n) return;
  flag[i] = i < n && vals[i] != NONE32 ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void ke_compact(const uint32_t* __restrict__ vals, uint32_t n, const uint32_t* __restrict__ ex, uint32_t* __restrict__ packed) {
  uint32_t i = gtid();
  if (i < n && vals[i] != NONE32) packed[ex[i]] = vals[i];
}
__global__ __launch_bounds__(BLOCK) void ke_deltas(const uint32_t* __restrict__ vals, uint32_t n, const uint32_t* __restrict__ ex, const uint32_t* __restrict__ packed,
                                                   uint32_t* __restrict__ deltas, uint8_t* __restrict__ nullmask, const uint32_t* __restrict__ seg) {
  uint32_t i = gtid();
  if (i >= n) return;
  uint32_t v = vals[i];
  if (v == NONE32) { deltas[i] = 0; nullmask[i] = 1; return; }
  uint32_t k = ex[i];
  const uint32_t k0 = seg ? ex[seg[i]] : 0u;   // non-null values in front of the segment: the running value starts at 0 in every segment
  deltas[i] = v - (k > k0 ? packed[k - 1] : 0u);  // modulo 2^32: the counters are below 2^31, so the difference is an exact int32
  nullmask[i] = 0;
}
void enc_delta_prepare(const uint32_t* vals, uint32_t n, uint32_t* deltas, uint8_t* nullmask, EncWork& w, hipStream_t st, const uint32_t* seg) {
  if (!n) return;
  AM355_LAUNCH_INDEPENDENT(ke_nonnull_flags, grid_for(n + 1), dim3(BLOCK), st, vals, n, w.flag);
  exclusive_scan_u32(w.flag, w.run_ex, n + 1, nullptr, w.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(ke_compact, grid_for(n), dim3(BLOCK), st, vals, n, (const uint32_t*)w.run_ex, w.run_first);
  AM355_LAUNCH_INDEPENDENT(ke_deltas, grid_for(n), dim3(BLOCK), st, vals, n, (const uint32_t*)w.run_ex, (const uint32_t*)w.run_first, deltas, nullmask, seg);
}

// boolean ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void ke_bool_flags(const uint8_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ flag, const uint32_t* __restrict__ seg) {
  uint32_t i = gtid();
  if (i > n) return;
  flag[i] = i < n && (i == 0 || (seg && seg[i] == i) || (v[i] != 0) != (v[i - 1] != 0)) ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void ke_bool_sizes(const uint8_t* __restrict__ v, uint32_t n, const uint32_t* __restrict__ run_ex,
                                                       const uint32_t* __restrict__ run_first, uint32_t* __restrict__ size, const uint32_t* __restrict__ seg) {
  uint32_t r = gtid();
  if (r > n) return;
  uint32_t R = run_ex[n], sz = 0;
  if (r < R) {
    const uint32_t a = run_first[r];
    sz = uleb_size(run_first[r + 1] - a);
    if ((a == 0 || (seg && seg[a] == a)) && v[a]) sz += 1;  // the first run counts `false`: empty when the column (segment) starts with `true`
  }
  size[r] = sz;
}
__global__ __launch_bounds__(BLOCK) void ke_bool_write(const uint8_t* __restrict__ v, uint32_t n, const uint32_t* __restrict__ run_ex,
                                                       const uint32_t* __restrict__ run_first, const uint32_t* __restrict__ off_ex, uint8_t* __restrict__ out,
                                                       const uint32_t* __restrict__ seg) {
  uint32_t r = gtid();
  if (r >= n || r >= run_ex[n]) return;
  uint8_t* p = out + off_ex[r];
  const uint32_t a = run_first[r];
  if ((a == 0 || (seg && seg[a] == a)) && v[a]) *p++ = 0;
  put_uleb(p, run_first[r + 1] - a);
}
void enc_boolean(const uint8_t* vals, uint32_t n, EncWork& w, uint8_t* out, uint32_t* d_len, hipStream_t st, const uint32_t* seg) {
  if (!n) { (void)hipMemsetAsync(d_len, 0, 4, st); return; }
  AM355_LAUNCH_INDEPENDENT(ke_bool_flags, grid_for(n + 1), dim3(BLOCK), st, vals, n, w.flag, seg);
  exclusive_scan_u32(w.flag, w.run_ex, n + 1, nullptr, w.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(ke_run_first, grid_for(n + 1), dim3(BLOCK), st, (const uint32_t*)w.flag, (const uint32_t*)w.run_ex, n, w.run_first);
  AM355_LAUNCH_INDEPENDENT(ke_bool_sizes, grid_for(n + 1), dim3(BLOCK), st, vals, n, (const uint32_t*)w.run_ex, (const uint32_t*)w.run_first, w.size, seg);
  exclusive_scan_u32(w.size, w.off_ex, n + 1, d_len, w.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(ke_bool_write, grid_for(n), dim3(BLOCK), st, vals, n, (const uint32_t*)w.run_ex, (const uint32_t*)w.run_first, (const uint32_t*)w.off_ex, out, seg);
}

// raw values -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void ke_raw_sizes(const uint32_t* __restrict__ val_tl, uint32_t n, uint32_t* __restrict__ size) {
  uint32_t i = gtid();
  if (i > n) return;
  size[i] = i < n ? val_tl[i] >> 4 : 0;
}
__global__ __launch_bounds__(BLOCK) void ke_raw_copy(const uint8_t* __restrict__ arena, const uint32_t* __restrict__ val_off, const uint32_t* __restrict__ val_tl,
                                                     uint32_t n, const uint32_t* __restrict__ off_ex, uint8_t* __restrict__ out) {
  uint32_t i = gtid();
  if (i >= n) return;
  uint32_t len = val_tl[i] >> 4;
  const uint8_t* a = arena + val_off[i];
  uint8_t* p = out + off_ex[i];
  for (uint32_t k = 0; k < len; k++) p[k] = a[k];
}
void enc_raw_values(const uint8_t* arena, const uint32_t* val_off, const uint32_t* val_tl, uint32_t n, EncWork& w, uint8_t* out, uint32_t* d_len, hipStream_t st) {
  if (!n) { (void)hipMemsetAsync(d_len, 0, 4, st); return; }
  AM355_LAUNCH_INDEPENDENT(ke_raw_sizes, grid_for(n + 1), dim3(BLOCK), st, val_tl, n, w.size);
  exclusive_scan_u32(w.size, w.off_ex, n + 1, d_len, w.scan_ws, st);
  AM355_LAUNCH_INDEPENDENT(ke_raw_copy, grid_for(n), dim3(BLOCK), st, arena, val_off, val_tl, n, (const uint32_t*)w.off_ex, out);
}

}  // namespace am355
