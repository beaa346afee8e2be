// Device-side column encoders (RLE / delta / boolean / raw values) for Backend.save: see am355_encode.hip.
#pragma once
#include "am355_device.h"
#include <stddef.h>

namespace am355 {

struct EncWork {  // scratch of one column encode, each array [n + 2]
  uint32_t *flag, *run_ex, *run_first, *grp_flag, *grp_ex, *grp_first, *size, *off_ex;
  void* scan_ws;
};
size_t enc_work_bytes(uint32_t n);
void enc_carve(EncWork& w, void* base, uint32_t n);
// upper bound of the encoded size of an n-value numeric column (10 bytes per value covers a header and a 5-byte value each)
inline size_t enc_numbers_bound(uint32_t n) { return 10 * (size_t)n + 16; }

// RLE of nullable numbers (encoding.js:558-783). nullmask == nullptr: NONE32 is null. is_signed: values are int32 written
// as signed LEB128 (the delta columns), else unsigned LEB128. *d_len (device) receives the byte count.
// `seg` (every encoder; null = the column is one segment): seg[i] = index of the first value of the segment of value i. The columns
// of ALL changes of a document are encoded in one go for the history reconstruction (am355_hist.hip): a segment is one change,
// runs / literal stretches / null runs / delta chains end where a segment ends, and an all-null segment is empty.
// `cap`: bytes `out` holds -- a run that would end beyond it is not written (*d_len still reports the full size: the caller compares).
void enc_rle_numbers(const uint32_t* vals, const uint8_t* nullmask, uint32_t n, bool is_signed, EncWork& w, uint8_t* out, uint32_t* d_len, hipStream_t st,
                     const uint32_t* seg = nullptr, uint32_t cap = 0xffffffffu);
// successive differences of the non-null values (encoding.js:932-948); deltas / nullmask feed enc_rle_numbers(is_signed)
void enc_delta_prepare(const uint32_t* vals, uint32_t n, uint32_t* deltas, uint8_t* nullmask, EncWork& w, hipStream_t st, const uint32_t* seg = nullptr);
// RLE of nullable UTF-8 strings given as arena ranges (len NONE32 = null)
void enc_rle_strings(const uint8_t* arena, const uint32_t* off, const uint32_t* len, uint32_t n, EncWork& w, uint8_t* out, uint32_t* d_len, hipStream_t st,
                     const uint32_t* seg = nullptr, uint32_t cap = 0xffffffffu);
// alternating run lengths, the first run counts `false` (encoding.js:1061-1135)
void enc_boolean(const uint8_t* vals, uint32_t n, EncWork& w, uint8_t* out, uint32_t* d_len, hipStream_t st, const uint32_t* seg = nullptr);
// concatenation of the value bytes (val_tl >> 4 bytes at val_off) of every row
void enc_raw_values(const uint8_t* arena, const uint32_t* val_off, const uint32_t* val_tl, uint32_t n, EncWork& w, uint8_t* out, uint32_t* d_len, hipStream_t st);
// Byte offsets of the segments in the output of the LAST encode done with `w` (seg_base[k] = index of segment k's first value,
// seg_base[n_seg] = n): seg_off[0 .. n_seg]. _raw: after enc_raw_values (offsets by value, not by run).
void enc_segment_offsets(const uint32_t* seg_base, uint32_t n_seg, const EncWork& w, uint32_t* seg_off, hipStream_t st);
void enc_segment_offsets_raw(const uint32_t* seg_base, uint32_t n_seg, const EncWork& w, uint32_t* seg_off, hipStream_t st);

}  // namespace am355
