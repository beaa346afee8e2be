/*
 * am_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the reference's bulk change-replay path
 * (automerge-classic: backend/columnar.js, backend/encoding.js, backend/new.js), used only as the
 * checker for the HIP engine: by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg.
 * The product (automerge_classic_amd/, include/am355.h) never links or calls anything in oracle/.
 *
 * Parity pin: this restatement is checked against patches produced by the unmodified reference JS
 * backend (run under node in the build container by oracle/js/make_golden.js; fixtures committed in
 * tests/golden/), and its SHA-256 against the checksums the reference's own tests pin
 * (test/columnar_test.js:17,57; test/new_backend_test.js:1860; test/backend_test.js:735).
 */
#ifndef AM_ORACLE_H
#define AM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct amo_doc amo_doc;

/* FIPS 180-4 SHA-256 (the reference uses fast-sha256@1.3.0: columnar.js:21,676-679,699-701). */
void amo_sha256(const uint8_t *data, size_t len, uint8_t out[32]);

/*
 * Backend.loadChanges(Backend.init(), changes): `arena` holds n binary changes back to back,
 * change i = arena[offsets[i] .. offsets[i+1]). Returns NULL and fills `err` when the reference would
 * throw (or for the few inputs this restatement refuses, prefixed "unsupported:").
 * Follows new.js:1797-1879 (BackendDoc.applyChanges) and new.js:1550-1597 (applyChanges).
 */
amo_doc *amo_replay(const uint8_t *arena, const uint64_t *offsets, uint32_t n_changes, char *err, size_t errcap);

/* Backend.load(document bytes): columnar.js:1006-1038 + new.js:1645-1675, 1695-1750. The patch of the loaded state is
 * then amo_patch_json(). Returns NULL and fills err when the reference would throw (or "unsupported:" inputs). */
amo_doc *amo_load_document(const uint8_t *doc, size_t len, char *err, size_t errcap);

/* JSON.stringify(Backend.getPatch(state)) -- new.js:2060-2068, 1604-1635, 884-1040. NUL-terminated,
 * owned by the doc. Returns NULL (and fills err) if the reference would throw while building the patch. */
const char *amo_patch_json(amo_doc *doc, size_t *len, char *err, size_t errcap);

void amo_free(amo_doc *doc);

/*
 * Backend.applyChanges(state, changes) with the INCREMENTAL patch it returns (am_oracle_apply.c restates
 * new.js:1797-1879, 1550-1597, 1052-1290 mergeDocChangeOps line by line, 884-1040 updatePatchProperty in its incremental mode,
 * 747-869, 1461-1528 setupPatches).  A session document starts with amo_init() (Backend.init()) or amo_load_document() and is
 * advanced call by call, as the reference was called.  Returns JSON.stringify(patch) (owned by the doc, valid until the next
 * call) or NULL with `err` filled when the reference would throw ("unsupported:" for what this restatement refuses); after an
 * error the document is unusable (the reference leaves it unchanged: rebuild it).  is_local: the call came from
 * applyLocalChange (the patch of a single change then carries `actor` and `seq`, new.js:1874-1877).
 * amo_patch_json(doc) afterwards gives Backend.getPatch of the same state.
 */
amo_doc *amo_init(void);
/* A BackendDoc made by Backend.load knows the hashes of the document's heads only and rebuilds the hash graph when a scheduling
 * round applies nothing (new.js:1833-1840, computeHashGraph :1887-1912) -- into a fresh index that lacks what the running call has
 * applied so far. This restatement does not rebuild changes from a document: the test gives the hashes the reference's
 * getAllChanges(load(doc)) has (32 bytes each, document order; kept by pointer). Without them a call that would need the graph fails
 * with "unsupported:". rebuilt != 0: the reference has rebuilt the graph already (it was asked for changes: new.js:1922). */
void amo_set_document_history(amo_doc *doc, const uint8_t *hashes, uint32_t n, int rebuilt);
const char *amo_apply_changes(amo_doc *doc, const uint8_t *arena, const uint64_t *offsets, uint32_t n_changes, int is_local,
                              size_t *len, char *err, size_t errcap);

/* ---- introspection used by stage-level parity tests ---- */
uint32_t amo_num_changes(const amo_doc *doc);          /* changes given */
uint32_t amo_num_applied(const amo_doc *doc);          /* changes applied (rest are pending) */
const uint8_t *amo_change_hashes(const amo_doc *doc);  /* 32 bytes per given change, input order */
uint64_t amo_num_ops(const amo_doc *doc);              /* op rows in applied changes (dels included) */
uint64_t amo_max_op(const amo_doc *doc);
uint32_t amo_num_actors(const amo_doc *doc);
/* actor i (document order = first-applied order): pointer to raw id bytes, length via *len */
const uint8_t *amo_actor(const amo_doc *doc, uint32_t i, uint32_t *len);

/*
 * Canonical op store (the order of rows in the reference's document op columns).
 * Fills caller arrays of length amo_num_rows(): id (ctr, actor), obj (ctr, actor; ctr 0 = _root),
 * insert flag, action, succ count. Actor numbers are document actor indexes.
 */
uint64_t amo_num_rows(const amo_doc *doc);
void amo_rows(const amo_doc *doc, uint64_t *id_ctr, uint32_t *id_actor, uint64_t *obj_ctr, uint32_t *obj_actor,
              uint8_t *insert, uint32_t *action, uint32_t *succ_num);

/* Decode ONE change (columnar.js:741-765 + new.js:570-610,678-724) into fixed-width arrays, for checking
 * the decode kernels in isolation. Actor numbers are change-local indexes (0 = author). Null is reported as
 * ctr = UINT64_MAX / actor = UINT32_MAX / key_len = UINT32_MAX. Offsets are relative to `raw`, the
 * uncompressed (chunk type 1) form of the change. Returns NULL and fills err on malformed input. */
typedef struct {
  uint64_t *obj_ctr, *key_ctr, *val_tag_len, *val_off;
  uint32_t *obj_actor, *key_actor, *action, *pred_num;
  uint32_t *key_off, *key_len;
  uint8_t *insert;
  uint64_t *pred_ctr; /* flattened pred lists */
  uint32_t *pred_actor;
  uint64_t n_ops, n_preds;
  uint64_t seq, start_op;
  uint32_t n_deps, n_actors;
  uint8_t hash[32];
  uint64_t raw_len;
  uint8_t *raw;
} amo_decoded_t;
amo_decoded_t *amo_decode_change(const uint8_t *change, size_t len, char *err, size_t errcap);
void amo_decoded_free(amo_decoded_t *o);

#ifdef __cplusplus
}
#endif
#endif
