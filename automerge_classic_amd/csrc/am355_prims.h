// Device-wide primitives (see am355_prims.hip).
#pragma once
#include "am355_device.h"
#include <stddef.h>

namespace am355 {
size_t scan_workspace_bytes(uint32_t n);
// out[i] = sum(in[0..i)); in == out allowed. *d_total (device, optional) receives the grand total.
void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* d_total, void* ws, hipStream_t st);
// two scans over the same range in one pass (same workspace size); in == out allowed, totals optional
void exclusive_scan2_u32(const uint32_t* in_a, uint32_t* out_a, uint32_t* d_total_a, const uint32_t* in_b, uint32_t* out_b, uint32_t* d_total_b, uint32_t n,
                         void* ws, hipStream_t st);
// *d_out = max(*d_out, max(v[0..n)))
void max_u32(const uint32_t* v, uint32_t n, uint32_t* d_out, hipStream_t st);
size_t sort_workspace_bytes(uint32_t n);
int radix_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, uint32_t n, int begin_bit, int end_bit,
                     void* ws, hipStream_t st);
// Forward chain marking: next[i] > i, or >= n (NONE32) at the end of a chain. mark[] holds the start nodes on entry and
// is nonzero on every node reachable from a start on return (values already nonzero are kept). `work` needs
// chain_work_bytes(n) bytes.
size_t chain_work_bytes(uint32_t n);
void chain_mark(const uint32_t* next, uint32_t n, uint32_t* mark, void* work, hipStream_t st);
}  // namespace am355
