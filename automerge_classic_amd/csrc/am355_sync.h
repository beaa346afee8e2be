// Bloom filters of the sync protocol (backend/sync.js:38-128) over the change hashes resident in HBM (see am355_sync.hip).
#pragma once
#include "am355_device.h"

namespace am355 {
// bits[] (n_bits / 8 bytes, zeroed by the callee) gets the probes of hashes[32 * idx[k]] for k < n set
void launch_bloom_build(const uint8_t* hashes, const uint32_t* idx, uint32_t n, uint32_t* bit_words, uint32_t n_bits, uint32_t num_probes, hipStream_t st);
// contains[k] = 1 iff every probe of hashes[32 * idx[k]] is set in bits[]
void launch_bloom_probe(const uint8_t* hashes, const uint32_t* idx, uint32_t n, const uint8_t* bits, uint32_t n_bits, uint32_t num_probes, uint8_t* contains,
                        hipStream_t st);
}  // namespace am355
