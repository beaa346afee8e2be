/*
 * am355.h -- C ABI of the MI355X-native bulk change-replay engine for automerge-classic's backend.
 *
 * This is the drop-in boundary for ONE path of the reference: Backend.loadChanges(Backend.init(), changes)
 * followed by Backend.getPatch(state)  (reference: backend/backend.js:116-129, backend/new.js:1797-1879 and
 * 2060-2068; callers: src/automerge.js:52-55 load, :105-118 getHistory, :43-46 clone) -- plus the two calls on
 * either side of it: Backend.load(bytes) (am355_load_document) and Backend.save(state) (am355_save).  The reference is 100 %
 * JavaScript and has no FFI; the host-side binding a maintainer adds is the N-API addon in
 * automerge_classic_amd/js/ (see INTEGRATION.md), which calls exactly these entry points and re-exports the
 * Backend module surface (backend/index.js:1-8) with every other call delegated to the JS backend.
 *
 * Plain pointers and sizes only; no C++ or torch types.  All functions return 0 on success or a negative
 * AM355_E_* code; am355_last_error() gives a message.  A context is bound to one GPU and one HIP stream and
 * is not thread-safe (the reference API is synchronous and single-threaded: backend/columnar.js:8-12); the engine itself
 * uses host threads internally (column DEFLATE in am355_save, patch text in am355_patch_json).
 *
 * The engine has no CPU fallback: without a gfx950 device am355_create() fails.
 */
#ifndef AM355_H
#define AM355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct am355_ctx am355_ctx;

enum {
  AM355_OK = 0,
  AM355_E_DEVICE = -1,      /* no usable GPU / HIP error */
  AM355_E_ARG = -2,         /* bad argument */
  AM355_E_INVALID = -3,     /* the reference would throw on this input (RangeError); see am355_flags() */
  AM355_E_UNSUPPORTED = -4, /* legal input outside the GPU-served subset; the JS host replays it on the JS path */
  AM355_E_STATE = -5,       /* call sequence error */
  AM355_E_NOMEM = -6
};

/* Validity / support flags (bit set), as raised by the device kernels and the host scheduler. */
enum {
  AM355_F_BAD_MAGIC = 1u << 0, AM355_F_BAD_CHECKSUM = 1u << 1, AM355_F_BAD_CHUNK = 1u << 2, AM355_F_BAD_COLUMNS = 1u << 3,
  AM355_F_BAD_LEB = 1u << 4, AM355_F_BAD_RLE = 1u << 5, AM355_F_BAD_ROW = 1u << 6, AM355_F_UNKNOWN_OBJECT = 1u << 7,
  AM355_F_BAD_ELEM = 1u << 8, AM355_F_BAD_PRED = 1u << 9, AM355_F_DUP_OPID = 1u << 10, AM355_F_BAD_COUNTER = 1u << 11,
  AM355_F_UNSUPPORTED = 1u << 12, AM355_F_OVERFLOW = 1u << 13,
  AM355_F_BAD_SEQ = 1u << 16,        /* new.js:1571-1578 sequence number reuse / gap */
  AM355_F_UNKNOWN_ACTOR = 1u << 17,  /* new.js:1442-1449 */
  AM355_F_BAD_DEFLATE = 1u << 18     /* columnar.js:813-823 inflate failure */
};

am355_ctx *am355_create(int device_ordinal);
void am355_destroy(am355_ctx *ctx);
const char *am355_last_error(const am355_ctx *ctx);
uint32_t am355_flags(const am355_ctx *ctx);

/*
 * Stage a batch of binary changes: `arena` holds n_changes change containers back to back, change i occupying
 * arena[offsets[i] .. offsets[i+1]).  DEFLATEd changes (chunk type 2, columnar.js:798-823) are inflated on the
 * host (libdeflate when the system has it, else zlib) into an internal pinned "raw arena", which is copied to HBM.  Replaces
 * the per-buffer decodeChangeColumns() preamble of BackendDoc.applyChanges (new.js:1806-1810).  The caller's buffers are not
 * retained: when the call returns every byte is in the engine's arena -- the copies to HBM may still be running; am355_replay
 * is ordered behind them on the context's stream (the next load waits for them before it rewrites the arena).
 */
int am355_load_changes(am355_ctx *ctx, const uint8_t *arena, const uint64_t *offsets, uint32_t n_changes);

/*
 * Stage one saved document (chunk type 0, as written by Backend.save / Automerge.save): container + header parse, chunk
 * checksum, zlib inflate of DEFLATEd columns and the change-metadata scan (clock) on the host, op columns to HBM.
 * Replaces decodeDocumentHeader + readDocumentChanges of `new BackendDoc(buffer)` (columnar.js:1006-1038, new.js:1645-1675,
 * 1695-1750).  am355_replay then decodes the op columns and builds the whole-document patch on the device, so
 * am355_load_document + am355_replay + am355_patch_json == JSON.stringify(Backend.getPatch(Backend.load(bytes))).
 */
int am355_load_document(am355_ctx *ctx, const uint8_t *doc, size_t len);

/*
 * Backend.load(bytes) in ONE call (backend/backend.js:104-107 -> new BackendDoc(buffer), new.js:1695-1750) = am355_load_document +
 * am355_replay, with two things the two-call form cannot do: the chunk checksum (columnar.js:699-705, one SHA-256 over the whole
 * document: a dependent chain of ~20 ms for a 44 MB document, longer than everything else of the load) runs on a thread of its own
 * beside the inflate, the copies to HBM AND the device stages, and its verdict is asked for at the end (a mismatch outranks any other
 * finding, as in the reference, which verifies it first); and the copy of the patch IR to host memory is enqueued behind the device
 * stages right away, so that am355_fetch_ir / am355_patch_json find it under way. Same results, errors and flags as the two calls.
 */
int am355_backend_load(am355_ctx *ctx, const uint8_t *doc, size_t len);

/*
 * The hot path, device-resident in and out: container parse + SHA-256 + column decode, causal scheduling
 * (host, between two device phases), op-set merge, RGA ordering, whole-document patch IR.  Equivalent to
 * Backend.loadChanges(Backend.init(), changes) + the work of Backend.getPatch().  Blocking.
 * May be called repeatedly on the same staged batch (each call recomputes everything).
 */
int am355_replay(am355_ctx *ctx);

/* Copy the patch IR to the host (pinned buffers owned by ctx) and render JSON.stringify(Backend.getPatch(state))
 * byte-for-byte.  `*json` is NUL-terminated and valid until the next call on ctx. */
int am355_patch_json(am355_ctx *ctx, const char **json, size_t *len);

/* ---- results / introspection ---- */
typedef struct {
  uint32_t n_changes, n_applied, n_pending;
  uint32_t n_actors, n_objects, n_heads;
  uint64_t n_ops;        /* op rows in applied changes (dels included): the unit of the ops/s metric */
  uint64_t max_op;
  uint64_t raw_bytes;    /* bytes of the staged (uncompressed) changes */
  uint64_t n_map_values, n_list_elems, n_edits;
  uint64_t ir_bytes;     /* bytes of patch IR produced in HBM by the last replay */
  /* timing of the last am355_replay (milliseconds; device figures from HIP events on the engine's stream). ms_parse / ms_decode /
     ms_merge / ms_order are measured only by a context created with AM355_PHASE_EVENTS=1 in the environment (else 0): an event
     record between two kernels is a packet of its own in front of the next dispatch, i.e. microseconds of the replay itself. */
  float ms_total, ms_parse, ms_host_schedule, ms_decode, ms_merge, ms_order;
  float ms_hash_stream;  /* SHA-256 + dependency resolution on the second stream (overlaps decode/merge) */
  uint32_t fast_path;    /* 1: in-order fast path (device-verified); 2: general path, scheduled on the device (csrc/am355_sched.hip);
                            0: general path, scheduled by the host (thousands of actors, pathological dependency chains) */
} am355_stats;
int am355_get_stats(const am355_ctx *ctx, am355_stats *out);
/* Measurement switch: on != 0 makes the following replays record HIP events between their phases (ms_parse / ms_decode / ms_merge /
 * ms_order of am355_stats); 0 (the default, unless AM355_PHASE_EVENTS=1 is in the environment at am355_create) leaves them out of the
 * stream and reports those four as 0. No counterpart in the reference. */
int am355_set_phase_events(am355_ctx *ctx, int on);

/* 32-byte SHA-256 change hashes in input order (columnar.js:693-705). `out` holds 32 * n_changes bytes. */
int am355_get_hashes(const am355_ctx *ctx, uint8_t *out);

/* Input indexes of the applied changes, in application order (what BackendDoc.changes holds, backend/new.js:1847, and
 * Backend.getAllChanges returns, new.js:1924-1927): duplicates and queued changes do not appear. Valid after am355_replay
 * of changes. out may be NULL to query the count. */
int am355_get_applied(const am355_ctx *ctx, uint32_t *out, uint32_t *n_applied);

/* Backend.save(state) (reference: backend/new.js:2033-2055 BackendDoc.save, backend/columnar.js:983-1004
 * encodeDocumentHeader): the document as one binary chunk -- actor table in order of first appearance, heads, change
 * metadata columns, all non-`del` op rows in canonical order with their succ lists (columns encoded on the GPU),
 * columns of >= 256 bytes DEFLATEd. Valid after am355_replay. A state produced by am355_load_document returns the
 * bytes it was loaded from, as the reference does (new.js:2034); flags bit 0 forces a re-encode of its op columns
 * (diagnostic for the column encoders). AM355_E_UNSUPPORTED while changes are queued (the JS path saves those).
 * *bytes is owned by ctx and valid until the next am355_save / am355_destroy. */
int am355_save(am355_ctx *ctx, uint32_t flags, const uint8_t **bytes, size_t *len);

/* History of a loaded document (reference: backend/new.js:1887-1912 BackendDoc.computeHashGraph ->
 * backend/columnar.js:1040-1047 decodeDocument, :876-943 groupChangeOps, :945-981 decodeDocumentChanges, :710-739 encodeChange):
 * the binary changes the document was made of, in document order -- what Backend.getAllChanges(Backend.load(bytes)) returns --
 * and their 32-byte hashes.  The op columns come from the GPU decode of am355_replay; regrouping rows into changes
 * (`del` ops rebuilt from succ lists, preds = inverse of succ), re-encoding each change and chaining the hashes runs on the
 * engine's host threads.  flags bit 0: DEFLATE changes of >= 256 bytes as encodeChange does (columnar.js:738).
 * Valid after am355_load_document + am355_replay.  AM355_E_INVALID when the document's heads do not match the rebuilt hash graph
 * or its rows / change metadata contradict each other (the reference throws); AM355_E_UNSUPPORTED for documents holding what this
 * path does not rebuild byte for byte (child / link columns, byte-array or unknown-typed values, non-minimal numbers, columns
 * outside the modelled set): the JS path serves those.  Pointers are owned by ctx and valid until the next load / destroy. */
int am355_doc_changes(am355_ctx *ctx, uint32_t flags, const uint8_t **arena, const uint64_t **offsets, uint32_t *n_changes,
                      const uint8_t **hashes);

/* Raw (uncompressed) arena as staged by am355_load_changes: pointers valid until the next load. */
int am355_get_raw(const am355_ctx *ctx, const uint8_t **arena, const uint64_t **offsets, uint32_t *n_changes);

/*
 * Patch IR on the host (valid after am355_fetch_ir or am355_patch_json; owned by ctx, pinned memory).  This is the OUTPUT of the
 * hot path: three record tables written by the device in exactly this layout and copied to the host as they are (three copies).
 * The N-API addon hands them to JavaScript as external ArrayBuffers and automerge_classic_amd/js/materialize.js builds the patch
 * object from them; am355_patch_json renders the same tables as JSON text.  Values and map keys are byte ranges of the raw
 * arena; op ids are (counter, actor rank) with the actor table in the envelope.
 *
 *   objects   index 0 is _root, then the make ops in application order.  A map/table object owns the map records
 *             [map_begin, map_end), a list/text object the edit records [edit_begin, edit_end).
 *   map       one record per visible value of a map key (conflicts: several records with the same key), sorted by (object, key in
 *             UTF-16 code unit order, op id)  -- new.js:1035-1039 `props[key][opId] = value`.
 *   edits     the edits of the reference's whole-document patch, in document order: insert, update, or multi-insert (new.js:747-782
 *             appendEdit: consecutive op ids of one actor, elemId == opId, same value class).  A record carries
 *             count = edits[k+1].first - first values (the table ends with a sentinel record whose `first` is n_values), all with
 *             the SAME type/length word, their bytes back to back in the arena from val_off on -- consecutive ops of one change
 *             have consecutive values in its valRaw column, so a typed run is one record and needs no per-value table.  A
 *             multi-insert whose values change length (mixed-width UTF-8) continues in the next record (AM355_EDIT_CONT): the
 *             host appends that record's values to the same edit.  count >= 2 or a following CONT record = multi-insert.
 *             (type/length word as in the valLen column: len << 4 | type, columnar.js:300-329.)
 */
typedef struct {
  uint32_t id_ctr, id_actor;      /* objectId = id_ctr@actor (ignored for _root) */
  uint32_t type;                  /* action of the make op: 0 makeMap, 2 makeList, 4 makeText, 6 makeTable */
  uint32_t map_begin, map_end;    /* map / table: range of map records */
  uint32_t edit_begin, edit_end;  /* list / text: range of edit records */
  uint32_t make_row;              /* (engine-internal: op row of the make op) */
} am355_ir_object;
enum { AM355_MAP_COUNTER = 1u, AM355_MAP_CHILD = 2u,
       AM355_MAP_EMPTY = 4u   /* incremental patches only: the key is left without a value (`props[key] = {}`, new.js:1037) */ };
typedef struct {
  uint32_t id_ctr, id_actor;      /* opId under which the value is listed */
  uint32_t key_off, key_len;      /* key bytes (UTF-8) in the arena */
  uint32_t val_tl, val_off;       /* value; AM355_MAP_CHILD: val_off = object index */
  uint32_t flags, pad;
  int64_t counter;                /* AM355_MAP_COUNTER: the counter's total (new.js:937-967) */
} am355_ir_map;
enum { AM355_EDIT_UPDATE = 1u, AM355_EDIT_CONT = 2u, AM355_EDIT_CHILD = 4u,
       AM355_EDIT_REMOVE = 8u, /* `remove` edit of count = next record's first - first elements (new.js:775-777, 1029): incremental patches, and
                                  whole-document patches of lists that hold a visible row without a value (an increment of a deleted counter) */
       AM355_EDIT_MULTI = 16u, /* incremental patches only: a `multi-insert` edit even with one value left (appendUpdate took the last
                                  value of a two-value multi-insert away, new.js:812-814; the reference keeps the edit's action) */
       AM355_EDIT_COUNTER = 32u /* the value is the total of a counter inside a list (new.js:937-965): one value per record,
                                  (int64_t)((uint64_t)pad << 32 | val_off); val_tl keeps the type/length word of the counter's `set` */ };
typedef struct {
  uint32_t flags;                 /* AM355_EDIT_UPDATE: `update` edit (else insert / multi-insert); AM355_EDIT_CHILD: the value is an object;
                                     AM355_EDIT_CONT: more values of the previous record's multi-insert */
  uint32_t index;                 /* list index (of the record's first value) */
  uint32_t id_ctr, id_actor;      /* opId (of the record's first value) */
  uint32_t elem_ctr, elem_actor;  /* elemId */
  uint32_t first;                 /* ordinal of the record's first value among all list values; count = next record's first - first */
  uint32_t val_tl;                /* type/length word of every value of the record */
  uint32_t val_off;               /* arena offset of the first value (value i at val_off + i * (val_tl >> 4)); AM355_EDIT_CHILD: object index */
  uint32_t pad;                   /* AM355_EDIT_COUNTER: high word of the total */
} am355_ir_edit;

typedef struct {
  uint32_t n_objects, n_map, n_edits, n_values;   /* n_values: visible list values (sum of the records' counts) */
  const am355_ir_object *objects; /* [n_objects] */
  const am355_ir_map *map;        /* [n_map] */
  const am355_ir_edit *edits;     /* [n_edits + 1] (sentinel) */
  /* envelope */
  uint64_t max_op;
  uint32_t n_actors;            /* actors by rank (lexicographic order of raw ids) */
  const uint32_t *actor_off;    /* [n_actors+1] into actor_bytes */
  const uint8_t *actor_bytes;
  uint32_t n_clock;
  const uint32_t *clock_actor;  /* actor rank, in first-applied order (JS property order of `clock`) */
  const uint64_t *clock_seq;
  uint32_t n_heads;
  const uint8_t *heads;         /* 32 bytes each, sorted */
  uint32_t pending;
  const uint8_t *arena;         /* raw arena (values and keys are ranges of it) */
  uint64_t arena_len;
} am355_patch_ir;
int am355_fetch_ir(am355_ctx *ctx, am355_patch_ir *out);

/*
 * Backend.applyChanges(state, changes) with the INCREMENTAL patch it returns (reference: backend/backend.js:27-31,
 * backend/new.js:1797-1879 BackendDoc.applyChanges, :1052-1290 mergeDocChangeOps, :884-1040 updatePatchProperty in its incremental
 * mode, :747-782 appendEdit, :1461-1528 setupPatches).  `state` is what the context holds: the changes of an earlier
 * am355_load_changes + am355_replay or of earlier am355_apply_changes calls (applied and queued ones), or nothing -- then the call
 * is Backend.applyChanges(Backend.init(), changes).  The engine replays the earlier changes and the batch together (the queue
 * semantics of new.js:1822-1841 included) and derives the patch of the batch from the merged state and from which op rows are new
 * (automerge_classic_amd/csrc/am355_delta.hip): list edits with the index they had when the op was applied, the visible values of every map key
 * the batch touched, and the links from the touched objects up to _root.  Afterwards am355_apply_patch_json / am355_fetch_apply_ir
 * give that patch, and am355_patch_json / am355_fetch_ir / am355_save / am355_get_applied ... describe the new state as after
 * am355_replay.
 * AM355_E_INVALID: the reference throws on this batch.  AM355_E_UNSUPPORTED: legal, but outside the subset served here -- the error
 * text names the reason (DR_* in csrc/am355_delta.h): a list element that holds a `link` op, or counter rows on which the reference's
 * count of visible elements and the values it lists part, or more value rows than the stage walks, a deletion whose place in the
 * merge loop's work list is ambiguous, a property history
 * the device cannot replay (too long, or dependent on call boundaries the host did not keep), a sharded context.  A context made by
 * am355_load_document + am355_replay IS served: Backend.applyChanges(Backend.load(doc), changes) -- the engine rebuilds the document's
 * changes (am355_doc_changes: what computeHashGraph does in the reference), replays them as the state the batch goes onto, schedules
 * the batch as a BackendDoc without hash graph does (am355_hash_graph_known) and takes objectMeta from the document's rows; documents
 * am355_doc_changes refuses are refused here with its reason.  Assignments to list elements, conflicting ones included (several values per element: one edit record per
 * visible value), are served.  The host serves a refused call on the JS path.  After either error the context no longer holds a
 * state (every refusal path drops it: load again).
 */
int am355_apply_changes(am355_ctx *ctx, const uint8_t *arena, const uint64_t *offsets, uint32_t n_changes);
/* Forget the state the context holds: the next am355_apply_changes is Backend.applyChanges(Backend.init(), changes). */
int am355_reset(am355_ctx *ctx);
/* The staged changes were NOT applied by the one am355_load_changes + am355_replay that built the state: the host replayed the
 * retained changes of a state that several Backend.applyChanges calls had built (automerge_classic_amd/js/index.js does so when the
 * context of a state has moved on). Where those calls ended is then not known to the engine -- the reference's merge calls never
 * cross a call (new.js:1052-1290 runs per applyChanges), and what its objectMeta.children lists depends on them (new.js:916-931,
 * 1125-1149) --: am355_apply_changes then refuses the few patches that depend on it instead of assuming one call.
 * from_document = n > 0: the first n staged changes are the REBUILT history of a document that the reference loaded (Backend.load):
 * its objectMeta came from one pass over the document's rows (new.js:1604-1635, 1695-1750), not from that history -- the delta stage
 * replays that pass for the properties it must know (am355_delta.hip kh_simulate) --, and whether the reference has rebuilt the
 * document's hash graph by now is unknown unless am355_hash_graph_known says: a batch whose schedule depends on it is refused.
 * Holds until am355_reset / the next am355_load_changes. */
int am355_forget_call_history(am355_ctx *ctx, int from_document);
/* A BackendDoc made by Backend.load knows the hashes of the document's heads only; it rebuilds the hashes of all the document's changes
 * (computeHashGraph, new.js:1887-1912) when an applyChanges call gets stuck on a missing dependency (new.js:1833-1840) or when
 * getChanges / getChangeByHash / getMissingDeps / getChangesAdded / clone / applyLocalChange ask (new.js:1774, 1922, 1980, 2000, 2015,
 * backend.js:38). Until then applyChanges schedules against what it knows, and the call that rebuilds the graph forgets the hashes
 * of the changes it had applied so far -- both show in the patch (`pendingChanges`, `clock`, `deps`) and in the state. am355_apply_changes
 * reproduces that for a context made by am355_load_document (and for the calls that follow); the host, which serves those queries
 * from am355_doc_changes, says here that the reference would have the graph by now.
 * set: 1 = the graph has been rebuilt, 0 = not yet (after am355_forget_call_history(ctx, n)), -1 = only ask. *known (may be NULL):
 * the state after the call -- 1 for every lineage that did not begin with a document. */
int am355_hash_graph_known(am355_ctx *ctx, int set, int *known);

/* Input indexes of the changes still queued for a missing dependency (BackendDoc.queue, new.js:1866), in queue order. The engine's
 * own list of changes after am355_apply_changes is: the changes applied before the call in application order, the batch, the
 * changes queued before the call -- am355_get_applied / am355_get_pending / am355_get_hashes index that list. out may be NULL. */
int am355_get_pending(const am355_ctx *ctx, uint32_t *out, uint32_t *n_pending);
/* JSON.stringify of the patch Backend.applyChanges returned -- byte for byte; valid until the next call on ctx */
int am355_apply_patch_json(am355_ctx *ctx, const char **json, size_t *len);
/* The same patch as record tables (layout above; records may carry AM355_EDIT_REMOVE / AM355_MAP_EMPTY, map records of one object
 * come in the order the reference inserted the keys, objects the patch does not reach have empty ranges). */
int am355_fetch_apply_ir(am355_ctx *ctx, am355_patch_ir *out);

/*
 * Sync protocol, bulk side (reference backend/sync.js).  A sync message that carries changes is served by
 * am355_apply_changes; what the protocol computes around it over the hashes of a replayed state:
 *   am355_get_dep_graph     the dependency graph of the context's list of changes by index (change i depends on
 *                           dep_index[dep_first[i] .. dep_first[i + 1]); UINT32_MAX: a change the context does not hold) -- what
 *                           BackendDoc.getChanges / getMissingDeps walk (new.js:1921-1976, 2014-2028), resolved on the device while
 *                           the changes were hashed; the pointers are owned by ctx and valid until the next replay
 *   am355_sync_bloom_build  the Bloom filter over the hashes of the changes idx[0 .. n) (sync.js:38-128 BloomFilter of makeBloomFilter,
 *                           :240-244: 10 bits per entry, 7 probes, triple hashing over the first 12 bytes), built on the device from the
 *                           resident hashes; bits receives ceil(n * 10 / 8) bytes (the `bits` field of the filter)
 *   am355_sync_bloom_probe  for a filter received from a peer: contains[k] = 1 iff the hash of change idx[k] is in it
 *                           (BloomFilter.containsHash of getChangesToSend, sync.js:262-283)
 * Valid after am355_replay / am355_apply_changes of changes.
 */
int am355_get_dep_graph(am355_ctx *ctx, const uint32_t **dep_first, const uint32_t **dep_index, uint32_t *n_changes);
int am355_sync_bloom_build(am355_ctx *ctx, const uint32_t *idx, uint32_t n, uint8_t *bits, size_t capacity);
int am355_sync_bloom_probe(am355_ctx *ctx, const uint32_t *idx, uint32_t n, uint32_t num_entries, uint32_t bits_per_entry, uint32_t num_probes,
                           const uint8_t *bits, size_t n_bytes, uint8_t *contains);

/*
 * objectId sharding over several GPUs (one context per GPU, one process per GPU).  Ordering and pred / succ
 * resolution never cross objects (new.js:1141-1145, 1173-1176), so after am355_set_shard(ctx, rank, world) a replay merges
 * only the objects rank `rank` owns (_root: rank 0; any other object: (counter + actor rank) mod world) -- every rank still
 * stages and decodes the whole batch, which keeps op id -> row arithmetic and the object table identical everywhere; the one
 * cross-object link (make op -> child object, new.js:894-897, 973-976) is the object index.  The OUTPUT is what is exchanged:
 * am355_export_fragment writes this rank's record tables as one contiguous block into caller memory (device memory for an
 * RCCL all_gather over xGMI, or host memory), and am355_import_fragments stitches the blocks of all ranks into the patch IR
 * of the whole document, which am355_patch_json / am355_fetch_ir of that context then return.  world == 1 restores the
 * unsharded engine.  am355_save is not available on a sharded context.
 */
int am355_set_shard(am355_ctx *ctx, uint32_t rank, uint32_t world);
int am355_fragment_size(am355_ctx *ctx, size_t *bytes);
int am355_export_fragment(am355_ctx *ctx, void *dst, size_t capacity, int dst_is_device, size_t *len);
/* fragments of ranks 0..world-1 back to back in host memory, fragment r = frags[offsets[r] .. offsets[r+1]) */
int am355_import_fragments(am355_ctx *ctx, const uint8_t *frags, const uint64_t *offsets, uint32_t world);

/*
 * The sharded replay with its collective INSIDE the library: RCCL over xGMI, one process per GPU (north_star: "Changes shard by
 * objectId across the 8 GPUs of one node with an RCCL ... over xGMI").  The library opens librccl.so.1 when the first communicator
 * is made (AM355_RCCL_LIB overrides the name; nothing of RCCL is loaded by a single-GPU user) and calls ncclGetUniqueId /
 * ncclCommInitRank / ncclAllGather / ncclCommDestroy on the context's stream.
 *   am355_shard_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId); the host carries it to the other ranks' processes
 *                           (the JS host: process.send between the per-GPU workers, js/sharded.js).
 *   am355_shard_init        ncclCommInitRank on the context's GPU; implies am355_set_shard(ctx, rank, world).
 *   am355_sharded_replay    am355_replay of the staged batch (every rank stages the same batch) + one ncclAllGather of
 *                           {failed, fragment bytes} + one ncclAllGather of the patch-IR fragments, HBM to HBM + the stitch
 *                           (am355_import_fragments) on rank 0, or on every rank when stitch_on_all_ranks != 0.  A batch any rank
 *                           rejects is rejected on every rank (no rank waits in a collective).  Afterwards am355_patch_json /
 *                           am355_fetch_ir of a stitching rank return the patch of the WHOLE document.
 *   am355_shard_fragment_bytes  bytes every rank contributed to the last sharded replay (diagnostics, world entries).
 *   am355_shard_finalize    ncclCommDestroy; the context is unsharded again.
 * What the reference has in this place: nothing (a BackendDoc is one single-threaded object, new.js:1695); the partition is legal
 * because preds never cross objects (new.js:1141-1145, 1173-1176) and the one cross-object link is make op -> child (:894-897).
 */
#define AM355_SHARD_ID_BYTES 128
int am355_shard_unique_id(uint8_t id[AM355_SHARD_ID_BYTES]);
int am355_shard_init(am355_ctx *ctx, const uint8_t id[AM355_SHARD_ID_BYTES], uint32_t rank, uint32_t world);
int am355_sharded_replay(am355_ctx *ctx, int stitch_on_all_ranks);
int am355_shard_fragment_bytes(am355_ctx *ctx, uint64_t *bytes, uint32_t capacity);
int am355_shard_finalize(am355_ctx *ctx);

/* diagnostics: how many am355_apply_changes calls on this context merged the batch alone into the resident state (out[0]), how
 * many took the full replay although they asked for it (out[1]: new actor, dependency not applied yet, duplicate, capacity ...), and
 * how many of the first kind also merged their new list elements into the stored document order in place (out[2]) */
int am355_resident_counters(const am355_ctx *ctx, uint64_t out[3]);
/* ... and for how many of the first kind the map half of the merge ran on its own: batches of plain map rows only (`set` / `del` on
 * string keys: every list stayed as it was) and batches of such rows beside list edits (the list rows merged in place first) (*out) */
int am355_resident_maps_only_calls(const am355_ctx *ctx, uint64_t *out);

/* For bindings that mirror per-state tables of the context in their own memory (the N-API addon: BackendDoc.changes, their hashes and
 * the raw arena, backend/new.js:1847, 1855-1879) and must not copy all of them for every one-change Backend.applyChanges:
 *  - am355_arena_epoch: a counter that changes whenever bytes the context handed out as am355_patch_ir.arena may no longer be what
 *    they were (a new am355_load_changes / am355_load_document / am355_reset); while it stays the same the arena has only GROWN at its
 *    end (am355_apply_changes stages a batch behind the kept changes), so a mirror copies [its length, arena_len) only;
 *  - am355_get_hashes_range: change hashes [first, first + count) in input order, 32 bytes each;
 *  - am355_applied_in_input_order: *yes = 1 when every staged change is applied, in the order it was staged, and none is queued (the
 *    applied order is then 0 .. n_changes - 1 and the pending list empty: no need to fetch either). */
int am355_arena_epoch(const am355_ctx *ctx, uint64_t *epoch);
int am355_get_hashes_range(const am355_ctx *ctx, uint32_t first, uint32_t count, uint8_t *out);
int am355_applied_in_input_order(const am355_ctx *ctx, int *yes);

/* ---- diagnostics: device primitives exposed for kernel-level tests ---- */
int am355_test_sort(am355_ctx *ctx, uint64_t *keys, uint32_t *vals, uint32_t n, int key_bits);
int am355_test_scan(am355_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total);
/* decoded op rows of the last replay, copied to caller arrays of length n_ops (any pointer may be NULL) */
int am355_get_rows(am355_ctx *ctx, uint32_t *obj_actor, uint32_t *obj_ctr, uint32_t *key_actor, uint32_t *key_ctr, uint32_t *key_off,
                   uint32_t *key_len, uint32_t *action, uint32_t *val_tl, uint32_t *val_off, uint32_t *pred_num, uint32_t *id_ctr,
                   uint32_t *id_actor, uint8_t *insert, uint32_t *succ_cnt);

#ifdef __cplusplus
}
#endif
#endif
