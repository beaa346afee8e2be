// Host-side byte utilities around the device work (no device code): SHA-256 of one buffer (document chunk checksum), raw
// DEFLATE decode of one stream (compressed changes, document columns).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace am355 {

// FIPS 180-4 SHA-256 of p[0..len) (columnar.js:693-705: the chunk checksum is its first four bytes). Uses the x86 SHA
// extensions when the CPU has them (~2 GB/s instead of ~0.25 GB/s: the checksum of a 44 MB document was the longest
// single item of Backend.load's host side).
void sha256_digest(const uint8_t* p, size_t len, uint8_t out[32]);
// The same in pieces (the checksum of a document runs beside the copy of its bytes): h = sha256_initial(), whole 64-byte blocks through
// sha256_blocks as they become available, then the last `rem` < 64 bytes and the total length through sha256_finish.
void sha256_initial(uint32_t h[8]);
void sha256_blocks(uint32_t h[8], const uint8_t* p, size_t nblocks);
void sha256_finish(uint32_t h[8], const uint8_t* tail, size_t rem, uint64_t total_len, uint8_t out[32]);

// Raw DEFLATE of one stream (columnar.js:813-823, 1062-1067; the reference calls pako.inflateRaw). Returns 0 on success, 1 on
// malformed / truncated data, 2 when the inflated size would pass `cap`, 3 on allocation failure. A truncated stream ends in
// 1 (the reference throws a catchable error), never in an endless reallocation.
int inflate_raw(const uint8_t* in, size_t in_len, std::vector<uint8_t>& out, size_t cap);

}  // namespace am355
