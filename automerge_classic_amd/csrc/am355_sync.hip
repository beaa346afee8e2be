// Sync protocol, bulk side: the Bloom filters a peer sends and receives are built and probed from the 32-byte change
// hashes that the replay left in HBM (k_hash_changes), one lane per hash.
//
// Reference: backend/sync.js:38-128 class BloomFilter -- the first 12 bytes of a SHA-256 hash are three little-endian 32-bit words
// x, y, z, taken modulo the number of bits; probe 0 = x, then (x, y) <- ((x + y) mod m, (y + z) mod m) for every further probe (triple
// hashing, Dillinger & Manolios 2004); BITS_PER_ENTRY = 10, NUM_PROBES = 7 on the sending side, whatever the filter says on the
// receiving side (sync.js:28-31).  The arithmetic is JavaScript's: the sums stay below 2^33 and are exact.
#include "am355_sync.h"

namespace am355 {

__device__ __forceinline__ uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

__global__ __launch_bounds__(BLOCK) void k_bloom_build(const uint8_t* __restrict__ hashes, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ words,
                                                       uint32_t n_bits, uint32_t num_probes) {
  uint32_t k = gtid();
  if (k >= n || n_bits == 0) return;
  const uint8_t* h = hashes + 32 * (size_t)idx[k];
  unsigned long long x = le32(h) % n_bits, y = le32(h + 4) % n_bits, z = le32(h + 8) % n_bits;
  for (uint32_t i = 0; i < num_probes; i++) {
    if (i) { x = (x + y) % n_bits; y = (y + z) % n_bits; }
    atomicOr(&words[x >> 5], 1u << (x & 31));  // (bit x of the byte array = bit (x & 7) of byte x >> 3: the same bit of the little-endian word)
  }
}

__global__ __launch_bounds__(BLOCK) void k_bloom_probe(const uint8_t* __restrict__ hashes, const uint32_t* __restrict__ idx, uint32_t n, const uint8_t* __restrict__ bits,
                                                       uint32_t n_bits, uint32_t num_probes, uint8_t* __restrict__ contains) {
  uint32_t k = gtid();
  if (k >= n) return;
  bool all = n_bits != 0;
  if (all) {
    const uint8_t* h = hashes + 32 * (size_t)idx[k];
    unsigned long long x = le32(h) % n_bits, y = le32(h + 4) % n_bits, z = le32(h + 8) % n_bits;
    for (uint32_t i = 0; i < num_probes && all; i++) {
      if (i) { x = (x + y) % n_bits; y = (y + z) % n_bits; }
      all = (bits[x >> 3] >> (x & 7)) & 1;
    }
  }
  contains[k] = all ? 1 : 0;
}

void launch_bloom_build(const uint8_t* hashes, const uint32_t* idx, uint32_t n, uint32_t* bit_words, uint32_t n_bits, uint32_t num_probes, hipStream_t st) {
  (void)hipMemsetAsync(bit_words, 0, 4 * (((size_t)n_bits + 31) / 32 + 1), st);
  if (n) AM355_LAUNCH_INDEPENDENT(k_bloom_build, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), st, hashes, idx, n, bit_words, n_bits, num_probes);
}

void launch_bloom_probe(const uint8_t* hashes, const uint32_t* idx, uint32_t n, const uint8_t* bits, uint32_t n_bits, uint32_t num_probes, uint8_t* contains,
                        hipStream_t st) {
  if (n) AM355_LAUNCH_INDEPENDENT(k_bloom_probe, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), st, hashes, idx, n, bits, n_bits, num_probes, contains);
}

}  // namespace am355
