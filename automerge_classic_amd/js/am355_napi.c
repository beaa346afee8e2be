/*
 * am355_napi.c -- thin N-API (v8, Node >= 12) binding of the C ABI in include/am355.h.
 *
 * The reference has no native addon; this is the binding a maintainer adds so that the unchanged JavaScript
 * frontend can call the MI355X engine through `Automerge.setDefaultBackend(require('mi355x-backend'))`
 * (reference plug-in point: src/automerge.js:147-149; harness: test/wasm.js:12-25).  One JS function per C entry
 * point, no logic of its own.
 *
 *   const am = require('./am355_napi.node')
 *   const ctx = am.create(0)                       // throws if no MI355X is usable (no CPU fallback)
 *   am.loadChanges(ctx, [Uint8Array, ...])         // stage + inflate + copy to HBM
 *   am.loadDocument(ctx, Uint8Array)               // stage one saved document (Backend.save bytes)
 *   am.backendLoad(ctx, Uint8Array)                // Backend.load in one call: loadDocument + replay, checksum beside the device stages
 *   am.replay(ctx)                                 // the hot path (blocking, like every Backend call)
 *   am.patchJSON(ctx) -> string                    // JSON.stringify(getPatch) text, built from the device IR
 *   am.fetchIR(ctx[, reuse]) -> {objects, map, edits, arena, ...}  // the record tables; materialize.js builds the patch object.
 *                                                  // reuse = true: into the context's own ArrayBuffers (valid until its next fetchIR)
 *   am.stats(ctx) -> {nOps, nChanges, msTotal, ...};  am.hashes(ctx) -> Uint8Array(32 * n);  am.destroy(ctx)
 */
#include <node_api.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/am355.h"

#define NAPI_CALL(env, call)                                                   \
  do {                                                                         \
    if ((call) != napi_ok) {                                                   \
      napi_throw_error((env), NULL, "N-API call failed: " #call);              \
      return NULL;                                                             \
    }                                                                          \
  } while (0)

/* the external wraps a small box so that an explicit destroy() and the finalizer cannot both release the context */
/* It also keeps what the binding reuses from call to call on a context: the gather buffer of loadChanges / applyChanges (a fresh
 * 13 MB malloc per call cost more in page faults than the copy itself) and, for fetchIR(ctx, true), the JS ArrayBuffers the record
 * tables are copied into. */
enum { IR_OBJECTS = 0, IR_MAP, IR_EDITS, IR_ARENA, IR_SLOTS };
typedef struct {
  am355_ctx *ctx;
  uint8_t *gather; size_t gather_cap;
  uint64_t *offsets; size_t offsets_cap;
  napi_ref ir_ref[IR_SLOTS]; size_t ir_cap[IR_SLOTS];
  uint64_t arena_epoch; size_t arena_copied;   /* what the reusable arena buffer mirrors: bytes [0, arena_copied) of the arena of that epoch */
} ctx_box;

static void finalize_ctx(napi_env env, void *data, void *hint) {
  (void)hint;
  ctx_box *box = (ctx_box *)data;
  if (!box) return;
  if (box->ctx) am355_destroy(box->ctx);
  for (int k = 0; k < IR_SLOTS; k++)
    if (box->ir_ref[k]) napi_delete_reference(env, box->ir_ref[k]);
  free(box->gather);
  free(box->offsets);
  free(box);
}

static ctx_box *get_box(napi_env env, napi_value v) {
  void *p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((ctx_box *)p)->ctx) {
    napi_throw_type_error(env, NULL, "expected a live am355 context");
    return NULL;
  }
  return (ctx_box *)p;
}

static am355_ctx *get_ctx(napi_env env, napi_value v) {
  void *p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((ctx_box *)p)->ctx) {
    napi_throw_type_error(env, NULL, "expected a live am355 context");
    return NULL;
  }
  return ((ctx_box *)p)->ctx;
}

static napi_value throw_engine(napi_env env, am355_ctx *ctx, int rc) {
  /* error object carries the code and the validity flags so the JS wrapper can decide to replay on the JS path */
  napi_value msg, err, code, flags;
  napi_create_string_utf8(env, am355_last_error(ctx), NAPI_AUTO_LENGTH, &msg);
  napi_create_error(env, NULL, msg, &err);
  napi_create_int32(env, rc, &code);
  napi_create_uint32(env, am355_flags(ctx), &flags);
  napi_set_named_property(env, err, "am355Code", code);
  napi_set_named_property(env, err, "am355Flags", flags);
  napi_throw(env, err);
  return NULL;
}

static napi_value js_create(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int32_t device = 0;
  if (argc >= 1) napi_get_value_int32(env, argv[0], &device);
  am355_ctx *ctx = am355_create(device);
  if (!ctx) {
    napi_throw_error(env, NULL, "am355_create failed: no usable MI355X / HIP device (the engine has no CPU fallback)");
    return NULL;
  }
  ctx_box *box = (ctx_box *)calloc(1, sizeof(ctx_box));
  if (!box) { am355_destroy(ctx); napi_throw_error(env, NULL, "out of memory"); return NULL; }
  box->ctx = ctx;
  napi_value ext;
  NAPI_CALL(env, napi_create_external(env, box, finalize_ctx, NULL, &ext));
  return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info) {
  /* releases the engine context now; the finalizer of the external then finds an empty box */
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void *p = NULL;
  if (argc >= 1 && napi_get_value_external(env, argv[0], &p) == napi_ok && p && ((ctx_box *)p)->ctx) {
    am355_destroy(((ctx_box *)p)->ctx);
    ((ctx_box *)p)->ctx = NULL;
  }
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

/* loadChanges(ctx, changes) = am355_load_changes; applyChanges(ctx, changes) = am355_apply_changes (Backend.applyChanges onto the
 * state the context holds: the incremental patch is then read with fetchApplyIR / applyPatchJSON) */
static napi_value stage_changes(napi_env env, napi_callback_info info, int apply) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ctx_box *box = get_box(env, argv[0]);
  if (!box) return NULL;
  am355_ctx *ctx = box->ctx;
  bool is_array = false;
  napi_is_array(env, argv[1], &is_array);
  if (!is_array) { napi_throw_type_error(env, NULL, "applyChanges takes an array of Uint8Arrays"); return NULL; }
  uint32_t n = 0;
  napi_get_array_length(env, argv[1], &n);
  if (box->offsets_cap < (size_t)n + 1) {
    size_t cap = ((size_t)n + 1) * 2;
    uint64_t *q = (uint64_t *)realloc(box->offsets, sizeof(uint64_t) * cap);
    if (!q) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    box->offsets = q; box->offsets_cap = cap;
  }
  uint64_t *offsets = box->offsets;
  /* ONE pass over the array: every change is copied behind the one in front as its typed-array info arrives (two N-API calls per
   * change; the first version made five and went over the array twice) */
  size_t total = 0;
  offsets[0] = 0;
  for (uint32_t i = 0; i < n; i++) {
    napi_value el;
    napi_typedarray_type t; size_t len; void *data; napi_value ab; size_t off;
    if (napi_get_element(env, argv[1], i, &el) != napi_ok || napi_get_typedarray_info(env, el, &t, &len, &data, &ab, &off) != napi_ok || t != napi_uint8_array) {
      napi_throw_type_error(env, NULL, "change is not a Uint8Array");
      return NULL;
    }
    if (total + len > box->gather_cap) {
      size_t cap = (total + len) * 2 + 4096;
      uint8_t *q = (uint8_t *)realloc(box->gather, cap);
      if (!q) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
      box->gather = q; box->gather_cap = cap;
    }
    if (len) memcpy(box->gather + total, data, len);
    total += len;
    offsets[i + 1] = total;
  }
  static const uint8_t none = 0;
  int rc = apply ? am355_apply_changes(ctx, total ? box->gather : &none, offsets, n) : am355_load_changes(ctx, total ? box->gather : &none, offsets, n);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}
static napi_value js_load_changes(napi_env env, napi_callback_info info) { return stage_changes(env, info, 0); }
static napi_value js_apply_changes(napi_env env, napi_callback_info info) { return stage_changes(env, info, 1); }

static napi_value js_load_document(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  bool is_ta = false;
  napi_is_typedarray(env, argv[1], &is_ta);
  if (!is_ta) { napi_throw_type_error(env, NULL, "document is not a Uint8Array"); return NULL; }
  napi_typedarray_type t; size_t len; void *data; napi_value ab; size_t off;
  napi_get_typedarray_info(env, argv[1], &t, &len, &data, &ab, &off);
  if (t != napi_uint8_array) { napi_throw_type_error(env, NULL, "document is not a Uint8Array"); return NULL; }
  int rc = am355_load_document(ctx, (const uint8_t *)data, len);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

static napi_value js_backend_load(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  bool is_ta = false;
  napi_is_typedarray(env, argv[1], &is_ta);
  if (!is_ta) { napi_throw_type_error(env, NULL, "document is not a Uint8Array"); return NULL; }
  napi_typedarray_type t; size_t len; void *data; napi_value ab; size_t off;
  napi_get_typedarray_info(env, argv[1], &t, &len, &data, &ab, &off);
  if (t != napi_uint8_array) { napi_throw_type_error(env, NULL, "document is not a Uint8Array"); return NULL; }
  int rc = am355_backend_load(ctx, (const uint8_t *)data, len);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

static napi_value js_replay(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  int rc = am355_replay(ctx);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

static napi_value js_patch_json(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  const char *json = NULL;
  size_t len = 0;
  int rc = am355_patch_json(ctx, &json, &len);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value s;
  NAPI_CALL(env, napi_create_string_utf8(env, json, len, &s));
  return s;
}

/* save(ctx, flags) -> Uint8Array: am355_save (Backend.save, new.js:2033-2055); the bytes are copied into a JS-owned buffer */
static napi_value js_save(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t flags = 0;
  if (argc > 1) napi_get_value_uint32(env, argv[1], &flags);
  const uint8_t *bytes = NULL;
  size_t len = 0;
  int rc = am355_save(ctx, flags, &bytes, &len);
  if (rc) return throw_engine(env, ctx, rc);
  void *data = NULL;
  napi_value ab, ta;
  NAPI_CALL(env, napi_create_arraybuffer(env, len, &data, &ab));
  if (len) memcpy(data, bytes, len);
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, len, ab, 0, &ta));
  return ta;
}

/* appliedOrder(ctx) -> Uint32Array: input indexes of the applied changes in application order (am355_get_applied) */
static napi_value js_applied_order(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t n = 0;
  int rc = am355_get_applied(ctx, NULL, &n);
  if (rc) return throw_engine(env, ctx, rc);
  void *data = NULL;
  napi_value ab, ta;
  NAPI_CALL(env, napi_create_arraybuffer(env, 4 * (size_t)n, &data, &ab));
  rc = am355_get_applied(ctx, (uint32_t *)data, &n);
  if (rc) return throw_engine(env, ctx, rc);
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint32_array, n, ab, 0, &ta));
  return ta;
}

/* appliedInInputOrder(ctx) -> boolean: every staged change applied, in staged order, none queued (am355_applied_in_input_order) */
static napi_value js_applied_in_input_order(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out;
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  int yes = 0;
  int rc = am355_applied_in_input_order(ctx, &yes);
  if (rc) return throw_engine(env, ctx, rc);
  NAPI_CALL(env, napi_get_boolean(env, yes != 0, &out));
  return out;
}

/* forgetCallHistory(ctx, docChanges): am355_forget_call_history (the staged changes were replayed in one go, not by the calls that built the state) */
static napi_value js_forget_call_history(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t doc_changes = 0;  /* how many of the leading staged changes are the rebuilt history of a loaded document (0: none) */
  if (argc > 1) NAPI_CALL(env, napi_get_value_uint32(env, argv[1], &doc_changes));
  int rc = am355_forget_call_history(ctx, (int)doc_changes);
  if (rc) return throw_engine(env, ctx, rc);
  return NULL;
}

/* hashGraphKnown(ctx, set) -> boolean: am355_hash_graph_known (set: 1 / 0 / -1 = only ask) */
static napi_value js_hash_graph_known(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  int32_t set = -1;
  if (argc > 1) NAPI_CALL(env, napi_get_value_int32(env, argv[1], &set));
  int known = 1;
  int rc = am355_hash_graph_known(ctx, (int)set, &known);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value out;
  NAPI_CALL(env, napi_get_boolean(env, known != 0, &out));
  return out;
}

/* pendingOrder(ctx) -> Uint32Array: am355_get_pending (the changes still queued, as indexes into the engine's list of changes) */
static napi_value js_pending_order(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t n = 0;
  int rc = am355_get_pending(ctx, NULL, &n);
  if (rc) return throw_engine(env, ctx, rc);
  void *data = NULL;
  napi_value ab, ta;
  NAPI_CALL(env, napi_create_arraybuffer(env, (size_t)n * 4, &data, &ab));
  if (n) am355_get_pending(ctx, (uint32_t *)data, &n);
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint32_array, n, ab, 0, &ta));
  return ta;
}

static napi_value copy_to_arraybuffer(napi_env env, const void *src, size_t len);

static int get_u32_array(napi_env env, napi_value v, uint32_t **data, size_t *n) {
  bool is_ta = false;
  napi_is_typedarray(env, v, &is_ta);
  if (!is_ta) return 0;
  napi_typedarray_type t; size_t len; void *d; napi_value ab; size_t off;
  napi_get_typedarray_info(env, v, &t, &len, &d, &ab, &off);
  if (t != napi_uint32_array) return 0;
  *data = (uint32_t *)d;
  *n = len;
  return 1;
}

/* depGraph(ctx) -> { depFirst: Uint32Array[n + 1], depIndex: Uint32Array }: am355_get_dep_graph (copies) */
static napi_value js_dep_graph(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  const uint32_t *first = NULL, *index = NULL;
  uint32_t n = 0;
  int rc = am355_get_dep_graph(ctx, &first, &index, &n);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value o, ab, ta;
  NAPI_CALL(env, napi_create_object(env, &o));
  ab = copy_to_arraybuffer(env, first, 4 * ((size_t)n + 1));
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint32_array, (size_t)n + 1, ab, 0, &ta));
  napi_set_named_property(env, o, "depFirst", ta);
  ab = copy_to_arraybuffer(env, index, 4 * (size_t)first[n]);
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint32_array, first[n], ab, 0, &ta));
  napi_set_named_property(env, o, "depIndex", ta);
  return o;
}

/* bloomBuild(ctx, Uint32Array idx) -> Uint8Array: am355_sync_bloom_build (the `bits` of the filter over those changes' hashes) */
static napi_value js_bloom_build(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t *idx = NULL;
  size_t n = 0;
  if (!get_u32_array(env, argv[1], &idx, &n)) { napi_throw_type_error(env, NULL, "bloomBuild takes a Uint32Array of change indexes"); return NULL; }
  size_t bytes = (n * 10 + 7) / 8;
  void *data = NULL;
  napi_value ab, ta;
  NAPI_CALL(env, napi_create_arraybuffer(env, bytes, &data, &ab));
  int rc = am355_sync_bloom_build(ctx, idx, (uint32_t)n, (uint8_t *)data, bytes);
  if (rc) return throw_engine(env, ctx, rc);
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &ta));
  return ta;
}

/* bloomProbe(ctx, Uint32Array idx, numEntries, numBitsPerEntry, numProbes, Uint8Array bits) -> Uint8Array flags: am355_sync_bloom_probe */
static napi_value js_bloom_probe(napi_env env, napi_callback_info info) {
  size_t argc = 6;
  napi_value argv[6];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t *idx = NULL;
  size_t n = 0;
  if (!get_u32_array(env, argv[1], &idx, &n)) { napi_throw_type_error(env, NULL, "bloomProbe takes a Uint32Array of change indexes"); return NULL; }
  uint32_t ne = 0, nb = 0, np = 0;
  napi_get_value_uint32(env, argv[2], &ne); napi_get_value_uint32(env, argv[3], &nb); napi_get_value_uint32(env, argv[4], &np);
  bool is_ta = false;
  napi_is_typedarray(env, argv[5], &is_ta);
  if (!is_ta) { napi_throw_type_error(env, NULL, "filter bits must be a Uint8Array"); return NULL; }
  napi_typedarray_type t; size_t blen; void *bdata; napi_value bab; size_t boff;
  napi_get_typedarray_info(env, argv[5], &t, &blen, &bdata, &bab, &boff);
  if (t != napi_uint8_array) { napi_throw_type_error(env, NULL, "filter bits must be a Uint8Array"); return NULL; }
  void *data = NULL;
  napi_value ab, ta;
  NAPI_CALL(env, napi_create_arraybuffer(env, n ? n : 1, &data, &ab));
  int rc = am355_sync_bloom_probe(ctx, idx, (uint32_t)n, ne, nb, np, (const uint8_t *)bdata, blen, (uint8_t *)data);
  if (rc) return throw_engine(env, ctx, rc);
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta));
  return ta;
}

static napi_value js_reset(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  int rc = am355_reset(ctx);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

/* hashes(ctx[, first]) -> Uint8Array: the hashes of changes [first, n_changes), 32 bytes each (first = 0: all of them) */
static napi_value js_hashes(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t first = 0;
  if (argc > 1) (void)napi_get_value_uint32(env, argv[1], &first);
  am355_stats st;
  am355_get_stats(ctx, &st);
  if (first > st.n_changes) { napi_throw_range_error(env, NULL, "hashes: first beyond the staged changes"); return NULL; }
  void *data = NULL;
  napi_value ab, ta;
  size_t bytes = 32 * (size_t)(st.n_changes - first);
  NAPI_CALL(env, napi_create_arraybuffer(env, bytes, &data, &ab));
  int rc = am355_get_hashes_range(ctx, first, st.n_changes - first, (uint8_t *)data);
  if (rc) return throw_engine(env, ctx, rc);
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &ta));
  return ta;
}

static void set_num(napi_env env, napi_value obj, const char *name, double v) {
  napi_value n;
  napi_create_double(env, v, &n);
  napi_set_named_property(env, obj, name, n);
}


/* copy `len` bytes into a JS-owned ArrayBuffer (the engine's pinned buffers are reused by the next call on the context) */
static napi_value copy_to_arraybuffer(napi_env env, const void *src, size_t len) {
  void *data = NULL;
  napi_value ab;
  if (napi_create_arraybuffer(env, len, &data, &ab) != napi_ok) return NULL;
  if (len && src) memcpy(data, src, len);
  return ab;
}

/* fetchIR(ctx) -> {objects, map, edits, arena: ArrayBuffer, nObjects, nMap, nEdits, nValues, maxOp, pending,
 *                  actorOff: ArrayBuffer(u32), actorBytes: ArrayBuffer, clockActor: ArrayBuffer(u32), clockSeq: ArrayBuffer(f64), heads: ArrayBuffer}
 * am355_fetch_ir: the record tables of include/am355.h as the device wrote them; materialize.js builds the patch object. */
/* the table in a JS ArrayBuffer: a fresh one, or -- fetchIR(ctx, true) -- the context's own buffer for that table, grown when needed
 * (its content then changes with the next fetchIR on the context: for callers that materialise the patch at once, as index.js does) */
static napi_value ir_table(napi_env env, ctx_box *box, int slot, bool reuse, const void *src, size_t len) {
  if (!reuse) return copy_to_arraybuffer(env, src, len);
  napi_value ab = NULL;
  void *data = NULL;
  size_t have = 0;
  if (box->ir_ref[slot] && box->ir_cap[slot] >= len && napi_get_reference_value(env, box->ir_ref[slot], &ab) == napi_ok && ab &&
      napi_get_arraybuffer_info(env, ab, &data, &have) == napi_ok && have >= len) {
    if (len && src) memcpy(data, src, len);
    return ab;
  }
  if (box->ir_ref[slot]) { napi_delete_reference(env, box->ir_ref[slot]); box->ir_ref[slot] = NULL; box->ir_cap[slot] = 0; }
  size_t cap = (len + len / 4 + 4096 + 7) & ~(size_t)7;   /* (a multiple of 8: the JS side lays Uint32Array / Float64Array views over the whole buffer) */
  if (napi_create_arraybuffer(env, cap, &data, &ab) != napi_ok) return NULL;
  if (len && src) memcpy(data, src, len);
  if (napi_create_reference(env, ab, 1, &box->ir_ref[slot]) == napi_ok) box->ir_cap[slot] = cap;
  return ab;
}

/* The raw arena in the context's reusable buffer. Backend.applyChanges call after call appends a batch to the arena and keeps what is
 * there (am355_arena_epoch unchanged): only the new tail is copied -- the whole arena of a 1 M-op document is 13 MB, 0.3 ms of memcpy
 * per call for a batch of 3 KB. */
static napi_value arena_table(napi_env env, ctx_box *box, const void *src, size_t len) {
  uint64_t epoch = 0;
  napi_value ab = NULL;
  void *data = NULL;
  size_t have = 0;
  if (am355_arena_epoch(box->ctx, &epoch) == AM355_OK && epoch == box->arena_epoch && box->ir_ref[IR_ARENA] && box->ir_cap[IR_ARENA] >= len &&
      box->arena_copied <= len && napi_get_reference_value(env, box->ir_ref[IR_ARENA], &ab) == napi_ok && ab &&
      napi_get_arraybuffer_info(env, ab, &data, &have) == napi_ok && have >= len) {
    if (len > box->arena_copied && src) memcpy((uint8_t *)data + box->arena_copied, (const uint8_t *)src + box->arena_copied, len - box->arena_copied);
    box->arena_copied = len;
    return ab;
  }
  ab = ir_table(env, box, IR_ARENA, true, src, len);   /* (a new buffer comes with a quarter of room to grow into) */
  box->arena_epoch = epoch;
  box->arena_copied = ab ? len : 0;
  return ab;
}

static napi_value fetch_ir_common(napi_env env, napi_callback_info info, int apply) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ctx_box *box = get_box(env, argv[0]);
  if (!box) return NULL;
  am355_ctx *ctx = box->ctx;
  bool reuse = false;
  if (argc > 1) (void)napi_get_value_bool(env, argv[1], &reuse);
  am355_patch_ir ir;
  int rc = apply ? am355_fetch_apply_ir(ctx, &ir) : am355_fetch_ir(ctx, &ir);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
#define PUT_TAB(name, slot, ptr, bytes)                                                 \
  do {                                                                                  \
    napi_value ab_ = ir_table(env, box, (slot), reuse, (ptr), (bytes));                 \
    if (!ab_) { napi_throw_error(env, NULL, "out of memory (patch IR)"); return NULL; } \
    napi_set_named_property(env, o, name, ab_);                                         \
  } while (0)
#define PUT_AB(name, ptr, bytes)                                                        \
  do {                                                                                  \
    napi_value ab_ = copy_to_arraybuffer(env, (ptr), (bytes));                          \
    if (!ab_) { napi_throw_error(env, NULL, "out of memory (patch IR)"); return NULL; } \
    napi_set_named_property(env, o, name, ab_);                                         \
  } while (0)
  PUT_TAB("objects", IR_OBJECTS, ir.objects, (size_t)ir.n_objects * sizeof(am355_ir_object));
  PUT_TAB("map", IR_MAP, ir.map, (size_t)ir.n_map * sizeof(am355_ir_map));
  PUT_TAB("edits", IR_EDITS, ir.edits, ((size_t)ir.n_edits + 1) * sizeof(am355_ir_edit));
  if (reuse) {
    napi_value ab_ = arena_table(env, box, ir.arena, (size_t)ir.arena_len);
    if (!ab_) { napi_throw_error(env, NULL, "out of memory (patch IR)"); return NULL; }
    napi_set_named_property(env, o, "arena", ab_);
  } else PUT_TAB("arena", IR_ARENA, ir.arena, (size_t)ir.arena_len);
  PUT_AB("actorOff", ir.actor_off, ((size_t)ir.n_actors + 1) * sizeof(uint32_t));
  PUT_AB("actorBytes", ir.actor_bytes, ir.n_actors ? (size_t)ir.actor_off[ir.n_actors] : 0);
  PUT_AB("clockActor", ir.clock_actor, (size_t)ir.n_clock * sizeof(uint32_t));
  PUT_AB("heads", ir.heads, (size_t)ir.n_heads * 32);
  {
    void *data = NULL;
    napi_value ab;
    NAPI_CALL(env, napi_create_arraybuffer(env, (size_t)ir.n_clock * sizeof(double), &data, &ab));
    for (uint32_t i = 0; i < ir.n_clock; i++) ((double *)data)[i] = (double)ir.clock_seq[i];
    napi_set_named_property(env, o, "clockSeq", ab);
  }
#undef PUT_AB
#undef PUT_TAB
  set_num(env, o, "nObjects", ir.n_objects); set_num(env, o, "nMap", ir.n_map); set_num(env, o, "nEdits", ir.n_edits);
  set_num(env, o, "nValues", ir.n_values); set_num(env, o, "nActors", ir.n_actors); set_num(env, o, "maxOp", (double)ir.max_op);
  set_num(env, o, "pending", ir.pending);
  return o;
}
static napi_value js_fetch_ir(napi_env env, napi_callback_info info) { return fetch_ir_common(env, info, 0); }
/* fetchApplyIR(ctx): the patch of the last applyChanges as record tables (am355_fetch_apply_ir) */
static napi_value js_fetch_apply_ir(napi_env env, napi_callback_info info) { return fetch_ir_common(env, info, 1); }

/* docChanges(ctx, flags) -> { changes: Uint8Array[], hashes: Uint8Array }: am355_doc_changes -- the history of a loaded document
 * (Backend.getAllChanges(Backend.load(bytes)), new.js:1887-1927). The changes are views into ONE JS-owned ArrayBuffer (the
 * reference's change buffers are views too: encoding.js Encoder.buffer), hashes holds 32 bytes per change. */
static napi_value js_doc_changes(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t flags = 1;
  if (argc > 1) napi_get_value_uint32(env, argv[1], &flags);
  const uint8_t *arena = NULL, *hashes = NULL;
  const uint64_t *offsets = NULL;
  uint32_t n = 0;
  int rc = am355_doc_changes(ctx, flags, &arena, &offsets, &n, &hashes);
  if (rc) return throw_engine(env, ctx, rc);
  size_t total = (size_t)offsets[n];
  void *data = NULL;
  napi_value ab, list, out, hab, hta;
  NAPI_CALL(env, napi_create_arraybuffer(env, total, &data, &ab));
  if (total) memcpy(data, arena, total);
  NAPI_CALL(env, napi_create_array_with_length(env, n, &list));
  for (uint32_t i = 0; i < n; i++) {
    napi_value ta;
    NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, (size_t)(offsets[i + 1] - offsets[i]), ab, (size_t)offsets[i], &ta));
    NAPI_CALL(env, napi_set_element(env, list, i, ta));
  }
  hab = copy_to_arraybuffer(env, hashes, 32 * (size_t)n);
  if (!hab) return NULL;
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, 32 * (size_t)n, hab, 0, &hta));
  NAPI_CALL(env, napi_create_object(env, &out));
  NAPI_CALL(env, napi_set_named_property(env, out, "changes", list));
  NAPI_CALL(env, napi_set_named_property(env, out, "hashes", hta));
  return out;
}

static napi_value js_stats(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  am355_stats st;
  am355_get_stats(ctx, &st);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
  set_num(env, o, "nChanges", st.n_changes); set_num(env, o, "nApplied", st.n_applied); set_num(env, o, "nPending", st.n_pending);
  set_num(env, o, "nActors", st.n_actors); set_num(env, o, "nObjects", st.n_objects); set_num(env, o, "nOps", (double)st.n_ops);
  set_num(env, o, "maxOp", (double)st.max_op); set_num(env, o, "rawBytes", (double)st.raw_bytes); set_num(env, o, "nEdits", (double)st.n_edits);
  set_num(env, o, "msTotal", st.ms_total); set_num(env, o, "msParse", st.ms_parse); set_num(env, o, "msHostSchedule", st.ms_host_schedule);
  set_num(env, o, "msDecode", st.ms_decode); set_num(env, o, "msMerge", st.ms_merge); set_num(env, o, "msOrder", st.ms_order);
  set_num(env, o, "msHashStream", st.ms_hash_stream); set_num(env, o, "fastPath", st.fast_path);
  return o;
}

/* ---- objectId sharding with the collective inside the library (am355_shard_init / am355_sharded_replay: RCCL over xGMI) ----
 * shardUniqueId() -> Uint8Array(128)      rank 0's worker makes it; js/sharded.js hands it to the other workers (process.send)
 * shardInit(ctx, id, rank, world)         ncclCommInitRank on the context's GPU
 * shardedReplay(ctx, stitchOnAllRanks)    replay + ncclAllGather of the fragments + stitch; then patchJSON / fetchIR as usual
 * shardFragmentBytes(ctx, world) -> [..]  bytes each rank contributed;  shardFinalize(ctx)                                         */
static napi_value js_shard_unique_id(napi_env env, napi_callback_info info) {
  (void)info;
  void *data = NULL;
  napi_value ab, ta;
  NAPI_CALL(env, napi_create_arraybuffer(env, AM355_SHARD_ID_BYTES, &data, &ab));
  if (am355_shard_unique_id((uint8_t *)data) != AM355_OK) {
    napi_throw_error(env, NULL, "RCCL is not available (librccl.so.1, or AM355_RCCL_LIB)");
    return NULL;
  }
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, AM355_SHARD_ID_BYTES, ab, 0, &ta));
  return ta;
}

static napi_value js_shard_init(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  napi_typedarray_type ty;
  size_t len = 0, off = 0;
  void *data = NULL;
  napi_value ab;
  uint32_t rank = 0, world = 0;
  if (argc < 4 || napi_get_typedarray_info(env, argv[1], &ty, &len, &data, &ab, &off) != napi_ok || ty != napi_uint8_array || len != AM355_SHARD_ID_BYTES ||
      napi_get_value_uint32(env, argv[2], &rank) != napi_ok || napi_get_value_uint32(env, argv[3], &world) != napi_ok) {
    napi_throw_type_error(env, NULL, "shardInit(ctx, Uint8Array(128), rank, world)");
    return NULL;
  }
  int rc = am355_shard_init(ctx, (const uint8_t *)data, rank, world);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

static napi_value js_sharded_replay(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  bool all = false;
  if (argc > 1) (void)napi_get_value_bool(env, argv[1], &all);
  int rc = am355_sharded_replay(ctx, all ? 1 : 0);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

static napi_value js_shard_fragment_bytes(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t world = 0;
  if (argc < 2 || napi_get_value_uint32(env, argv[1], &world) != napi_ok || world == 0 || world > 4096) {
    napi_throw_type_error(env, NULL, "shardFragmentBytes(ctx, world)");
    return NULL;
  }
  uint64_t *bytes = (uint64_t *)calloc(world, sizeof(uint64_t));
  if (!bytes) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  int rc = am355_shard_fragment_bytes(ctx, bytes, world);
  if (rc) { free(bytes); return throw_engine(env, ctx, rc); }
  napi_value arr;
  napi_create_array_with_length(env, world, &arr);
  for (uint32_t r = 0; r < world; r++) {
    napi_value n;
    napi_create_double(env, (double)bytes[r], &n);
    napi_set_element(env, arr, r, n);
  }
  free(bytes);
  return arr;
}

static napi_value js_shard_finalize(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  am355_ctx *ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  int rc = am355_shard_finalize(ctx);
  if (rc) return throw_engine(env, ctx, rc);
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

static napi_value init(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
      {"create", NULL, js_create, NULL, NULL, NULL, napi_enumerable, NULL},
      {"destroy", NULL, js_destroy, NULL, NULL, NULL, napi_enumerable, NULL},
      {"loadChanges", NULL, js_load_changes, NULL, NULL, NULL, napi_enumerable, NULL},
      {"loadDocument", NULL, js_load_document, NULL, NULL, NULL, napi_enumerable, NULL},
      {"backendLoad", NULL, js_backend_load, NULL, NULL, NULL, napi_enumerable, NULL},
      {"replay", NULL, js_replay, NULL, NULL, NULL, napi_enumerable, NULL},
      {"patchJSON", NULL, js_patch_json, NULL, NULL, NULL, napi_enumerable, NULL},
      {"save", NULL, js_save, NULL, NULL, NULL, napi_enumerable, NULL},
      {"appliedOrder", NULL, js_applied_order, NULL, NULL, NULL, napi_enumerable, NULL},
      {"appliedInInputOrder", NULL, js_applied_in_input_order, NULL, NULL, NULL, napi_enumerable, NULL},
      {"hashes", NULL, js_hashes, NULL, NULL, NULL, napi_enumerable, NULL},
      {"stats", NULL, js_stats, NULL, NULL, NULL, napi_enumerable, NULL},
      {"docChanges", NULL, js_doc_changes, NULL, NULL, NULL, napi_enumerable, NULL},
      {"fetchIR", NULL, js_fetch_ir, NULL, NULL, NULL, napi_enumerable, NULL},
      {"applyChanges", NULL, js_apply_changes, NULL, NULL, NULL, napi_enumerable, NULL},
      {"pendingOrder", NULL, js_pending_order, NULL, NULL, NULL, napi_enumerable, NULL},
      {"forgetCallHistory", NULL, js_forget_call_history, NULL, NULL, NULL, napi_enumerable, NULL},
      {"hashGraphKnown", NULL, js_hash_graph_known, NULL, NULL, NULL, napi_enumerable, NULL},
      {"reset", NULL, js_reset, NULL, NULL, NULL, napi_enumerable, NULL},
      {"depGraph", NULL, js_dep_graph, NULL, NULL, NULL, napi_enumerable, NULL},
      {"bloomBuild", NULL, js_bloom_build, NULL, NULL, NULL, napi_enumerable, NULL},
      {"bloomProbe", NULL, js_bloom_probe, NULL, NULL, NULL, napi_enumerable, NULL},
      {"fetchApplyIR", NULL, js_fetch_apply_ir, NULL, NULL, NULL, napi_enumerable, NULL},
      {"shardUniqueId", NULL, js_shard_unique_id, NULL, NULL, NULL, napi_enumerable, NULL},
      {"shardInit", NULL, js_shard_init, NULL, NULL, NULL, napi_enumerable, NULL},
      {"shardedReplay", NULL, js_sharded_replay, NULL, NULL, NULL, napi_enumerable, NULL},
      {"shardFragmentBytes", NULL, js_shard_fragment_bytes, NULL, NULL, NULL, napi_enumerable, NULL},
      {"shardFinalize", NULL, js_shard_finalize, NULL, NULL, NULL, napi_enumerable, NULL},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
