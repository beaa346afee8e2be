// Parallel raw-DEFLATE decode of ONE long stream on the host pool (document columns: columnar.js:1062-1067 inflateColumn; the
// reference calls pako.inflateRaw on every column whose id carries the DEFLATE bit, one stream per column).
//
// A DEFLATE stream cannot be entered in the middle for two reasons: block boundaries are bit positions nobody recorded, and
// a back-reference may reach 32 KiB behind its block. Both are worked around (the two-stage scheme of pugz / rapidgzip):
//   search   -- per chunk of compressed bytes, the first bit position that reads as the header of a dynamic-Huffman block
//               (complete pre-code, complete literal/length code with an end-of-block symbol, valid distance code);
//   decode   -- from that position into 16-BIT symbols: 0..255 = a byte, 256 + w = "byte w of the 32 KiB in front of this
//               chunk", which is what a back-reference beyond the chunk's start copies (the unknown window is a prefix of
//               markers in front of the chunk's output, so copies need no special case). A chunk stops at the block boundary
//               that IS the next chunk's start; a found start that the chunk in front never lands on was a false positive and
//               is skipped (the chunk in front simply goes on);
//   link     -- serial per stream, O(chunks x 32 KiB): the chain of chunks from bit 0 to the final block, every chunk's
//               window from the one before it, output offsets;
//   resolve  -- parallel: symbols -> bytes through the chunk's window, straight into the caller's buffer (the pinned arena).
// Anything unexpected (no chain, a marker in the first chunk, output beyond the cap, a truncated stream) makes the job FAIL:
// the caller then runs the ordinary single-stream inflate (am355_host.cpp inflate_raw), whose verdict is the one reported.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <memory>
#include <vector>

namespace am355 {

constexpr uint32_t PINFLATE_WINDOW = 32768;

struct PInflateChunk {
  size_t nominal_bit = 0;                 // the search starts here ...
  size_t limit_bit = 0;                   // ... and gives up here (nominal bit of the next chunk)
  std::atomic<uint64_t> start_bit{0};     // published by the search: bit position, NOT_FOUND, or PENDING
  uint16_t* sym = nullptr;                // PINFLATE_WINDOW marker prefix + the chunk's symbols (buffer kept between jobs, never zeroed)
  size_t sym_cap = 0;
  size_t n_out = 0;                       // symbols produced (without the prefix)
  uint32_t next = 0;                      // chunk whose start this one reached; n_chunks = it decoded the final block
  int status = 0;                         // 0 not decoded, 1 ok, 2 failed
  size_t out_off = 0;                     // link: offset of the chunk's bytes in the stream's output
  std::vector<uint8_t> lut;               // link: symbol -> byte: the identity for 0..255, then the 32 KiB in front of the chunk (one branch-free gather resolves a symbol)
  ~PInflateChunk() { free(sym); }
};

struct PInflateJob {
  static constexpr uint64_t PENDING = ~0ull, NOT_FOUND = ~0ull - 1;
  static constexpr size_t PIECE = (size_t)1 << 20;  // symbols per resolve task
  const uint8_t* in = nullptr;
  size_t in_len = 0, cap = 0;
  unsigned n_chunks = 0;
  std::vector<std::unique_ptr<PInflateChunk>> chunks;  // (kept between jobs for their buffers; only the first n_chunks are used)
  // link results
  bool ok = false;
  size_t out_len = 0;
  uint8_t last_byte = 0;
  std::vector<unsigned> chain;
  std::atomic<int> resolve_failed{0};
  // symbols all chunks of the job have made so far (each chunk reports when it grows its buffer): the cap bounds the JOB, not each
  // chunk -- a stream of zeros lets every chunk of N grow towards `cap` on its own otherwise
  std::atomic<size_t> made_syms{0};

  // chunk_bytes: compressed bytes per chunk
  void prepare(const uint8_t* in_, size_t in_len_, size_t cap_, size_t chunk_bytes);
  void search(unsigned k);    // independent of everything; publishes chunks[k]->start_bit
  void decode(unsigned k);    // waits (spinning) for the searches of the chunks behind k: run it in a pool job whose search
                              // tasks have LOWER task indexes, so that they have all been drawn when a decode task runs
  bool link();                // serial, after all decode tasks
  unsigned n_pieces(unsigned ci) const { return (unsigned)((chunks[chain[ci]]->n_out + PIECE - 1) / PIECE); }
  void resolve(unsigned ci, unsigned r, uint8_t* dst);  // piece r of chunk chain[ci] -> dst[out_off + ...]
  // a column whose bytes nobody needs (an unknown op column): only the verdict zlib would give -- is some back-reference "too far back"?
  bool has_too_far_marker() const;
  // The symbol buffers are kept between jobs (a load of the same shape then faults no pages); trim frees every buffer above
  // `keep_chunk_bytes` and whatever exceeds `keep_job_bytes` in total, so that one large or hostile column does not stay resident.
  void trim(size_t keep_chunk_bytes, size_t keep_job_bytes);
};

// One-call form (tests, and streams inflated outside a larger pool job): runs the stages on `n_threads` std::threads.
// Returns 0 and fills `out`, or 1 when the parallel path gave up (the caller falls back to inflate_raw).
int inflate_raw_parallel(const uint8_t* in, size_t in_len, std::vector<uint8_t>& out, size_t cap, size_t chunk_bytes, unsigned n_threads);

}  // namespace am355
