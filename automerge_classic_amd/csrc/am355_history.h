// History reconstruction after Backend.load: the per-change binaries and hashes of a saved document.
//
// Reference: backend/new.js:1887-1912 computeHashGraph -> backend/columnar.js:1040-1047 decodeDocument, :876-943 groupChangeOps
// (ops -> changes by (actor, maxOp); `del` ops rebuilt from succ entries; preds = inverse of succ), :945-981 decodeDocumentChanges
// (deps by index -> hashes, each change re-encoded to learn its hash), :710-739 encodeChange, :370-444 encodeOps, :122-170 parseAllOpIds
// (change-local actor table: author first, the others sorted).
//
// Division of work: the op columns of the document are decoded on the GPU (am355_bigcol.hip, the Backend.load path) and stay there;
// the device regroups the rows into changes -- ids to slots, preds from succ lists, the changes' slot ranges -- and encodes the twelve
// op columns of ALL changes, segmented by change (am355_hist.hip + am355_encode.hip). The host reads the change metadata columns
// (history_metadata: a few thousand values), writes the headers around the column pieces and chains the hashes in document order
// (history_finish: a change's header holds the hashes of its dependencies; SHA-256 is one dependent block after the other, which a
// host core with SHA extensions does two orders of magnitude faster than a lane of the device).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

namespace am355 {

struct HistoryInput {
  // the op rows of the document stay in device memory (n_rows of them, n_succ succ entries)
  uint32_t n_rows = 0, n_succ = 0;
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

// one change of the document, from the change metadata columns (+ n_ops / start_op once the device has counted its ids)
struct HistoryChange {
  uint32_t actor = 0;  // rank
  uint64_t seq = 0, max_op = 0, start_op = 0;
  int64_t time = 0;
  std::string message, extra;
  uint32_t dep_first = 0, dep_num = 0;
  uint32_t op_base = 0, n_ops = 0;
  uint32_t prev_same_actor = 0xffffffffu;
};
struct HistoryMeta {
  std::vector<HistoryChange> chg;
  std::vector<uint32_t> dep_index;              // dependencies by change index, flattened (HistoryChange.dep_first / dep_num)
  uint32_t n_deps = 0;                          // entries of dep_index (known after history_metadata; filled by history_dependencies)
  std::vector<uint32_t> act_max, word_base;     // per actor: last maxOp | first 32-bit word of its stretch of the id bitmaps ([NA + 1])
};
constexpr int HISTORY_NCOL = 12;
// what the device stages left, in host memory
struct HistoryPieces {
  const uint32_t* chg_nops = nullptr;           // [NC] ops of every change
  const uint32_t* abits = nullptr;              // [NC x aw] actors a change mentions (bit = rank)
  uint32_t aw = 0;
  const uint32_t* col_off = nullptr;            // [HISTORY_NCOL][2 x (NC + 1)]: begin / end of change k in column q at [q][2k], [q][2k + 1]
  const uint8_t* col_bytes[HISTORY_NCOL] = {};  // the encoded columns, all changes back to back
};

struct HistoryOutput {
  std::vector<uint8_t> arena;     // the changes back to back, in document order
  std::vector<uint64_t> offsets;  // [n_changes + 1]
  std::vector<uint8_t> hashes;    // 32 bytes per change
};

enum { HISTORY_OK = 0, HISTORY_INVALID = 1, HISTORY_UNSUPPORTED = 2 };

// par(k, fn): runs fn(0..k-1) on the caller's host threads and returns when all are done.
using ParallelFor = std::function<void(unsigned, const std::function<void(unsigned)>&)>;

// change metadata columns -> meta (and the layout of the id bitmaps the device builds)
int history_metadata(const HistoryInput& in, HistoryMeta& meta, std::string& err);
// the dependency indexes of every change (meta.dep_index): needs only what history_metadata left; may run on another thread beside the device stages
int history_dependencies(const HistoryInput& in, HistoryMeta& meta, std::string& err);
// headers + column pieces -> changes, hash chain, containers. deflate: compress changes of >= 256 bytes as encodeChange does
// (columnar.js:798-811; zlib level 6 raw = pako's defaults).
int history_finish(const HistoryInput& in, HistoryMeta& meta, const HistoryPieces& pc, bool deflate, const ParallelFor& par, HistoryOutput& out, std::string& err);

}  // namespace am355
