// History reconstruction after Backend.load: the per-change binaries and hashes of a saved document.
//
// Reference: backend/new.js:1887-1912 computeHashGraph -> backend/columnar.js:1040-1047 decodeDocument, :876-943 groupChangeOps
// (ops -> changes by (actor, maxOp); `del` ops rebuilt from succ entries; preds = inverse of succ), :945-981 decodeDocumentChanges
// (deps by index -> hashes, each change re-encoded to learn its hash), :710-739 encodeChange, :370-444 encodeOps, :122-170 parseAllOpIds
// (change-local actor table: author first, the others sorted).
//
// Division of work: the op columns of the document are decoded on the GPU (am355_bigcol.hip, the Backend.load path); this module
// takes those rows on the host, regroups them into changes and re-encodes every change -- independent per change, on the engine's
// host threads -- then chains the hashes in document order (a change's header holds the hashes of its dependencies).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

namespace am355 {

struct HistoryInput {
  // op rows of the document in canonical order (actor fields: rank in `actors`; NONE32 where absent), succ lists flattened
  uint32_t n_rows = 0, n_succ = 0;
  const uint32_t *obj_actor = nullptr, *obj_ctr = nullptr, *key_actor = nullptr, *key_ctr = nullptr, *key_off = nullptr, *key_len = nullptr;
  const uint32_t *action = nullptr, *val_tl = nullptr, *val_off = nullptr, *succ_first = nullptr, *succ_num = nullptr, *id_ctr = nullptr, *id_actor = nullptr;
  const uint8_t* insert = nullptr;
  const uint32_t *succ_actor = nullptr, *succ_ctr = nullptr;
  const uint8_t* arena = nullptr;  // values and keys are byte ranges of it
  size_t arena_len = 0;
  const std::vector<std::string>* actors = nullptr;  // raw actor ids by rank (lexicographic)
  // change metadata columns of the document, inflated: (column id, bytes); actor indexes in them are DOCUMENT actor indexes
  const std::vector<std::pair<uint32_t, std::vector<uint8_t>>>* change_columns = nullptr;
  const std::vector<uint32_t>* doc_actor_rank = nullptr;  // document actor index -> rank
  const uint8_t* key_column = nullptr;  // the keyStr column as stored (its value count must equal n_rows, or be zero)
  size_t key_column_len = 0;
  size_t val_raw_len = 0;               // length of the valRaw column (the values of the rows must cover it exactly)
  const uint8_t* heads = nullptr;  // the document's heads, 32 bytes each, sorted
  uint32_t n_heads = 0;
};

struct HistoryOutput {
  std::vector<uint8_t> arena;     // the changes back to back, in document order
  std::vector<uint64_t> offsets;  // [n_changes + 1]
  std::vector<uint8_t> hashes;    // 32 bytes per change
};

enum { HISTORY_OK = 0, HISTORY_INVALID = 1, HISTORY_UNSUPPORTED = 2 };

// par(k, fn): runs fn(0..k-1) on the caller's host threads and returns when all are done.
using ParallelFor = std::function<void(unsigned, const std::function<void(unsigned)>&)>;

// deflate: compress changes of >= 256 bytes as encodeChange does (columnar.js:798-811; zlib level 6 raw = pako's defaults).
int reconstruct_history(const HistoryInput& in, bool deflate, const ParallelFor& par, HistoryOutput& out, std::string& err);

}  // namespace am355
