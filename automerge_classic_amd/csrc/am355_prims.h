// Device-wide primitives (see am355_prims.hip).
#pragma once
#include "am355_device.h"
#include "am355_canary.h"
#include <stddef.h>

namespace am355 {
// Several word ranges filled with a value each in ONE launch (a memset per range costs a launch gap each). Ranges of <= 64 words
// are written by one thread in the order given (they may overlap: a later range wins); longer ones must not overlap.
struct FillRanges {
  uint32_t* p[8];
  uint32_t n_words[8];
  uint32_t value[8];
  uint32_t n = 0;
  void add(void* q, size_t bytes, uint32_t v) { p[n] = (uint32_t*)q; n_words[n] = (uint32_t)((bytes + 3) / 4); value[n] = v; n++; }
};
void launch_fill_ranges(const FillRanges& f, hipStream_t st);
// Small host -> device copies as ONE launch: the kernel reads the (pinned, device-visible) sources over the link itself. A copy command
// costs the host ~5-8 us whatever its size; a call of Backend.applyChanges with a one-change batch has five of a few hundred bytes each.
struct CopyRanges {
  void* dst[8];
  const void* src[8];
  uint32_t bytes[8];
  uint32_t n = 0;
  void add(void* d, const void* s, size_t b) { dst[n] = d; src[n] = s; bytes[n] = (uint32_t)b; n++; }
};
void launch_copy_ranges(const CopyRanges& r, hipStream_t st);
// Result words of a phase into pinned host memory, then the sequence number (signal_host, am355_device.h), as a one-thread launch of its
// own behind the phase: for phases that end in a library scan or in one of several kernels. Two source ranges, a then b (n_b may be 0).
void launch_signal_words(const uint32_t* src_a, uint32_t n_a, const uint32_t* src_b, uint32_t n_b, uint32_t* host_words, volatile uint32_t* host_seq, uint32_t seq,
                         hipStream_t st);
size_t scan_workspace_bytes(uint32_t n);
// out[i] = sum(in[0..i)); in == out allowed. *d_total (device, optional) receives the grand total.
void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* d_total, void* ws, hipStream_t st);
// out[i] = number of bytes in [0, i) with bit 7 clear (LEB128 terminators), i = 0 .. L: the scan of a flag array that is never stored
void exclusive_scan_terminators(const uint8_t* bytes, uint32_t L, uint32_t* out, uint32_t* d_total, void* ws, hipStream_t st);
// two scans over the same range in one pass (same workspace size); in == out allowed, totals optional
void exclusive_scan2_u32(const uint32_t* in_a, uint32_t* out_a, uint32_t* d_total_a, const uint32_t* in_b, uint32_t* out_b, uint32_t* d_total_b, uint32_t n,
                         void* ws, hipStream_t st);
// *d_out = max(*d_out, max(v[0..n)))
void max_u32(const uint32_t* v, uint32_t n, uint32_t* d_out, hipStream_t st);
size_t sort_workspace_bytes(uint32_t n);
// first_hist_done: the caller's own kernel (the one that produced keys_a) has already left the histogram of the FIRST digit (bits
// [begin_bit, begin_bit + 8)) in sort_first_table(ws), laid out [digit x sort_tiles(n) + tile] over tiles of SORT_TILE_ELEMS elements --
// only for sorts with sort_is_fused(n)
int radix_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, uint32_t n, int begin_bit, int end_bit,
                     void* ws, hipStream_t st, bool first_hist_done = false);
constexpr uint32_t SORT_TILE_ELEMS = 2048;
uint32_t sort_tiles(uint32_t n);
bool sort_is_fused(uint32_t n);
uint32_t* sort_first_table(void* ws);
// Forward chain marking: next[i] > i, or >= n (NONE32) at the end of a chain. mark[] holds the start nodes on entry and
// is nonzero on every node reachable from a start on return (values already nonzero are kept). `work` needs
// chain_work_bytes(n) bytes.
size_t chain_work_bytes(uint32_t n);
void chain_mark(const uint32_t* next, uint32_t n, uint32_t* mark, void* work, hipStream_t st);
}  // namespace am355
