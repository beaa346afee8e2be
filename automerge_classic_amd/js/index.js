// mi355x-backend: drop-in replacement for automerge-classic's `backend` module whose bulk-replay path runs on an
// AMD MI355X through the N-API addon (am355_napi.node -> include/am355.h -> HIP kernels).
//
//   const Automerge = require('automerge')
//   Automerge.setDefaultBackend(require('mi355x-backend'))        // reference plug-in point: src/automerge.js:147-149
//
// Module surface = reference backend/index.js:1-8.  GPU-served calls (SURVEY.md "Scope decisions"):
//   loadChanges(init(), changes)  and  getPatch(state)             (backend/backend.js:116-129, new.js:1797-1879, 2060-2068)
//   applyChanges(state, changes) with its incremental patch        (backend/backend.js:27-31; SURVEY.md 8f-2) when `state` is empty
//                                                                  or was built by the engine
//   load / save / getAllChanges after load                         (SURVEY.md a21, 8f-1, 8f-3)
// Everything else (applyLocalChange, clone, hash-graph queries with dependencies) is delegated to the reference JS backend, onto
// which a GPU-built state is hydrated lazily -- by replaying the retained change buffers -- the first time such a
// call is made.  Inputs the engine rejects (AM355_E_INVALID: the reference would throw; AM355_E_UNSUPPORTED: legal
// but outside the GPU-served subset) are re-run on the JS path so the caller sees the reference's exact exception
// or result.  If the addon or the GPU is missing, requiring this module throws: there is no silent CPU mode
// (set MI355X_BACKEND_JS_ONLY=1 to run the pure pass-through plumbing configuration explicitly).
'use strict'
const path = require('path')

const JS_ONLY = process.env.MI355X_BACKEND_JS_ONLY === '1'
const HYDRATE_FROM_DOC = process.env.MI355X_HYDRATE === 'doc'
const PATCH_VIA_JSON = process.env.MI355X_PATCH_VIA_JSON === '1'   // A/B: JSON text rendered by the engine + JSON.parse
const { materialize } = require('./materialize.js')
let addon = null, ctx = null
// A few engine contexts, used least-recently-used first: a GPU-built state remembers which replay (generation) made it, and as
// long as that replay still sits in one of the contexts, Backend.save of the state is served from it instead of replaying the
// retained changes again (MI355X_CONTEXTS, default 4; every context owns its device buffers).
const MAX_CONTEXTS = Math.max(1, parseInt(process.env.MI355X_CONTEXTS || '4'))
const contexts = []
let tick = 0
if (!JS_ONLY) {
  addon = require(path.join(__dirname, 'am355_napi.node'))
  ctx = addon.create(parseInt(process.env.MI355X_DEVICE || '0'))   // throws without an MI355X
  contexts.push({ ctx, generation: 0, used: 0 })
}

// The unmodified reference backend (the package a user already has installed)
let refBackend = null
function ref() {
  if (!refBackend) {
    const spec = process.env.AUTOMERGE_BACKEND_PATH || 'automerge/backend'
    refBackend = require(spec)
  }
  return refBackend
}

const AM355_E_INVALID = -3, AM355_E_UNSUPPORTED = -4

// A state built on the GPU. `changes` are retained (by reference, like BackendDoc.changes new.js:1847) so that a
// JS BackendDoc can be hydrated later; `patch` is the whole-document patch computed by the engine.
class GpuState {
  constructor(changes, patch, heads, doc) {
    this.changes = changes   // retained change buffers (loadChanges), or null
    this.doc = doc || null   // retained document bytes (load), or null
    this.patch = patch
    this.heads = heads
    this.js = null           // hydrated reference backend handle
    this.generation = 0      // engine replay this state was built by (it may still sit in one of the contexts)
    this.applied = null      // input indexes of the applied changes, application order (loadChanges states)
    this.hashes = null       // 32 bytes per input change
    this.pending = 0
    this.byHash = null       // lazily: hex hash -> input index, applied changes only
    this.pendingIdx = null   // input indexes of the changes still queued (loadChanges / applyChanges states)
    this.inOrder = false     // every change applied in the order of `changes`, none queued: applied = 0 .. n - 1 (gpuApplyChanges)
    this.hashStore = null    // the growing buffer `hashes` is a view of, shared along a line of applyChanges calls
    this.calls = 1           // backend calls that built the state (1: one loadChanges / applyChanges onto an empty document)
    this.fromDocument = false // the lineage began with load(bytes): the reference's objectMeta came from the document, not from changes
    // How the reference would have come to this state, call by call: the document the lineage began with (or null) and the batch of
    // every loadChanges / applyChanges call since. hydrate() replays exactly these calls: what the reference's objectMeta.children
    // holds depends on where its calls ended (new.js:916-931), and a call the engine refused is re-run on a state that must be the
    // reference's own (ADVICE r3).
    this.baseDoc = null
    this.batches = null
  }
}
let generation = 0           // bumped by every GPU replay

// the context the next replay runs in (the least recently used one), stamped with a new generation
function acquireContext() {
  let e
  if (contexts.length < MAX_CONTEXTS && contexts.every(x => x.generation !== 0)) {
    e = { ctx: addon.create(parseInt(process.env.MI355X_DEVICE || '0')), generation: 0, used: 0 }
    contexts.push(e)
  } else {
    e = contexts.reduce((a, b) => (b.used < a.used ? b : a))
  }
  e.generation = ++generation
  e.used = ++tick
  ctx = e.ctx
  return e
}
// the context that still holds the replay of `gen`, if any
function contextOf(gen) {
  const e = contexts.find(x => x.generation === gen && gen !== 0)
  if (e) { e.used = ++tick; ctx = e.ctx }
  return e || null
}
const counters = { gpuLoadChanges: 0, gpuLoad: 0, gpuSave: 0, gpuHistory: 0, gpuApplyChanges: 0, saveReplays: 0, fallbackToJs: 0, hydrations: 0 }   // (diagnostics: which path served the calls)

function isFrozenCheck(backend) {
  // reference util.js:1-10
  if (backend.frozen) {
    throw new Error(
      'Attempting to use an outdated Automerge document that has already been updated. ' +
      'Please use the latest document state, or call Automerge.clone() if you really ' +
      'need to use this old document state.')
  }
}

function isEmptyRefState(backend) {
  const s = backend.state
  return s && !(s instanceof GpuState) && Array.isArray(s.changes) && s.changes.length === 0 &&
    (!s.queue || s.queue.length === 0) && !s.binaryDoc
}

// The JS (reference) handle equivalent to `backend`: a GPU-built state is hydrated once and the handle is cached on it.
// Read-only reference calls (clone, save, getAllChanges, getChanges, getChangesAdded, getChangeByHash, getMissingDeps,
// generateSyncMessage) use it WITHOUT freezing anything, as the reference never freezes on those (backend.js:12-14, 93-98,
// 139-196): `Automerge.load(bytes)` -> `getAllChanges(doc)` -> `Automerge.change(doc)` must keep working.
function hydrate(backend) {
  isFrozenCheck(backend)
  if (!(backend.state instanceof GpuState)) return backend
  const g = backend.state
  if (!g.js) {
    counters.hydrations++
    if (g.doc) { g.js = ref().load(g.doc); if (g.graphKnown && !g.js.state.haveHashGraph) g.js.state.computeHashGraph() }
    else if (HYDRATE_FROM_DOC && !JS_ONLY) {
      // hydrate the JS BackendDoc from the engine's save() bytes: Backend.load of a document is several times cheaper in JS than
      // replaying every change (opt-in: MI355X_HYDRATE=doc; the default replays the retained changes, exact by construction)
      let bytes = null
      try {
        if (!contextOf(g.generation)) replayForReading(g)
        bytes = addon.save(ctx, 0)
      } catch (e) {
        if (e.am355Code !== AM355_E_INVALID && e.am355Code !== AM355_E_UNSUPPORTED) throw e
      }
      g.js = bytes ? ref().load(bytes) : ref().loadChanges(ref().init(), g.changes)
    } else if (g.batches) {
      let handle = g.baseDoc ? ref().load(g.baseDoc) : ref().init()
      for (const batch of g.batches) {
        if (batch === GRAPH_QUERY) { if (!handle.state.haveHashGraph) handle.state.computeHashGraph() } else handle = ref().loadChanges(handle, batch)
      }
      g.js = handle
    } else g.js = ref().loadChanges(ref().init(), g.changes)
  }
  return g.js
}

// Mutating reference calls (applyChanges, applyLocalChange, loadChanges, receiveSyncMessage) TAKE the JS handle: the
// reference freezes the handle it is given and returns a new one (backend.js:9-31, util.js:1-10), so the wrapper handle is
// frozen with it.
function toJs(backend) {
  const handle = hydrate(backend)
  if (backend.state instanceof GpuState) {
    backend.state.js = null
    backend.frozen = true
  }
  return handle
}

// the patch object of the last replay: built from the record tables the device wrote (materialize.js) -- real JS values, no
// JSON text in between
function gpuPatch() {
  return PATCH_VIA_JSON ? JSON.parse(addon.patchJSON(ctx)) : materialize(addon.fetchIR(ctx, true))
}

function gpuReplay(changes) {
  acquireContext()
  addon.loadChanges(ctx, changes)
  addon.replay(ctx)
  return gpuPatch()
}

// The retained changes of `g` replayed into a fresh context, in one go. When that is not how the state came to be -- several calls
// built it, or its lineage began with a loaded document -- the engine is told (am355_forget_call_history: how many of the leading changes
// are the document's): the few patches of later applyChanges calls that depend on where the reference's calls ended, or on whether it
// had rebuilt the document's hash graph by then, are refused instead of guessed
function replayRetained(g) {
  gpuReplay(g.changes)
  g.generation = generation
  g.fromChanges = true
  if (g.calls > 1 || g.fromDocument || g.doc) addon.forgetCallHistory(ctx, g.doc ? g.changes.length : (g.docChanges || 0))
  if (g.fromDocument || g.doc) addon.hashGraphKnown(ctx, g.graphKnown ? 1 : 0)
}

// A lineage that began with a loaded document can hold queued changes whose dependencies ARE in the document: the call in which the
// reference rebuilt its hash graph forgot the hashes of what it had applied before (new.js:1837-1840), and what depended on those waits
// for the next call. A replay of all retained changes would apply them too early: such a state is replayed from its APPLIED changes
// only, and the next applyChanges hands the queue over behind its batch (the reference's queue is batch ++ queue, new.js:1822).
const heldBack = g => !!(g.fromDocument && g.pendingIdx && g.pendingIdx.length > 0)
const appliedOnly = g => Array.from(g.applied, i => g.changes[i])
function replayForReading(g) {   // whole-document patch / save of a state whose context has moved on
  if (heldBack(g)) return gpuReplay(appliedOnly(g))   // (the context then holds nobody's list: g.generation stays as it is)
  const patch = gpuReplay(g.changes)
  g.generation = generation
  return patch
}

function init() {
  return ref().init()
}

function loadChanges(backend, changes) {
  isFrozenCheck(backend)
  if (!JS_ONLY && isEmptyRefState(backend) && Array.isArray(changes) && changes.length > 0) {
    try {
      const patch = gpuReplay(changes)
      counters.gpuLoadChanges++
      backend.frozen = true
      const state = new GpuState(changes.slice(), patch, patch.deps)
      state.batches = [state.changes]
      state.generation = generation
      state.applied = addon.appliedOrder(ctx)
      state.pendingIdx = addon.pendingOrder(ctx)
      state.hashes = addon.hashes(ctx)
      state.pending = patch.pendingChanges
      return { state, heads: patch.deps }
    } catch (e) {
      if (e.am355Code !== AM355_E_INVALID && e.am355Code !== AM355_E_UNSUPPORTED && !(e instanceof TypeError)) throw e
      counters.fallbackToJs++
      // fall through: the reference path raises the exact exception (or serves the unsupported case)
    }
  }
  return ref().loadChanges(toJs(backend), changes)
}

function getPatch(backend) {
  isFrozenCheck(backend)
  if (backend.state instanceof GpuState) {
    const g = backend.state
    if (!g.patch) {
      // a state made by applyChanges: the whole-document patch is built when somebody asks for it (the engine context that
      // replayed the state still holds its record tables, else the retained changes are replayed)
      if (contextOf(g.generation)) g.patch = gpuPatch()
      else g.patch = replayForReading(g)
    }
    return g.patch
  }
  return ref().getPatch(backend)
}

function getHeads(backend) {
  return backend.heads   // reference backend.js:135-137
}

function load(data) {
  // Backend.load(bytes) (backend.js:104-107, new.js:1695-1750): header / checksum / inflate on the host, op-column decode and
  // whole-document patch on the GPU
  if (!JS_ONLY && data instanceof Uint8Array) {
    try {
      acquireContext()
      addon.backendLoad(ctx, data)
      const patch = gpuPatch()
      counters.gpuLoad++
      const state = new GpuState(null, patch, patch.deps, data)
      state.generation = generation
      return { state, heads: patch.deps }
    } catch (e) {
      if (e.am355Code !== AM355_E_INVALID && e.am355Code !== AM355_E_UNSUPPORTED && !(e instanceof TypeError)) throw e
      counters.fallbackToJs++
    }
  }
  return ref().load(data)
}

// Backend.save(state) (backend.js:93-95, new.js:2033-2055). A GPU-built state is saved by the engine: canonical row order and
// column encoding on the GPU, document assembly / DEFLATE / checksum on the host. If the engine context has moved on to
// another document since, the retained changes are replayed first (still far cheaper than the JS path).
function save(backend) {
  isFrozenCheck(backend)
  const g = backend.state
  if (!JS_ONLY && g instanceof GpuState) {
    if (g.doc) return g.doc   // unchanged loaded document: the bytes it was loaded from (new.js:2034)
    try {
      if (!contextOf(g.generation)) { counters.saveReplays++; replayForReading(g) }
      const bytes = addon.save(ctx, 0)
      counters.gpuSave++
      return bytes
    } catch (e) {
      if (e.am355Code !== AM355_E_INVALID && e.am355Code !== AM355_E_UNSUPPORTED) throw e
    }
  }
  return ref().save(hydrate(backend))
}

// History queries on a state built by loadChanges: the engine knows which changes were applied and in what order, and the
// change buffers are retained, so these need no JS BackendDoc. (Loaded documents and anything involving queued changes
// go to the reference path.)
function gpuHistory(backend, withQueue) {
  const g = backend.state
  if (JS_ONLY || !(g instanceof GpuState) || g.js) return null
  if (g.doc && !g.changes && !g.noHistory) loadedHistory(g)
  if (heldBack(g)) return null   // (its context list is not g.changes: the reference path, on the handle hydrate() makes call by call)
  return g.changes && g.applied && (g.pending === 0 || withQueue) ? g : null
}
// A BackendDoc made by load() rebuilds its hash graph IN PLACE when it is asked for changes, a change by hash, missing dependencies or
// a clone (new.js:1774, 1922, 1980, 2000, 2015) -- and schedules later applyChanges calls differently from then on (new.js:1826-1840).
// The wrapper serves those queries without a BackendDoc: it remembers that the reference would have the graph (am355_hash_graph_known).
const GRAPH_QUERY = Object.freeze([])   // (in GpuState.batches: here the reference rebuilt the hash graph because it was asked, see hydrate())
function noteGraphQuery(backend) {
  const g = backend && backend.state
  if (g instanceof GpuState && (g.doc || g.fromDocument) && !g.graphKnown) {
    g.graphKnown = true
    if (g.batches) g.batches = g.batches.concat([GRAPH_QUERY])
    if (g.js && !g.js.state.haveHashGraph) g.js.state.computeHashGraph()
  }
}
// History of a LOADED document (new.js:1887-1912 computeHashGraph, columnar.js:876-981): the engine rebuilds the binary changes
// and their hashes from the rows it decoded (am355_doc_changes) -- once, on the first history query, as the reference defers it.
// Documents it does not rebuild (or whose heads do not match) go to the reference path, which serves them or throws its error.
function loadedHistory(g) {
  try {
    if (!contextOf(g.generation)) {
      acquireContext()
      addon.loadDocument(ctx, g.doc)
      addon.replay(ctx)
      g.generation = generation
    }
    const h = addon.docChanges(ctx, 1)
    g.changes = h.changes
    g.hashes = h.hashes
    g.applied = Uint32Array.from(h.changes, (_, i) => i)
    g.pendingIdx = new Uint32Array(0)
    g.pending = 0
    counters.gpuHistory++
  } catch (e) {
    if (e.am355Code !== AM355_E_INVALID && e.am355Code !== AM355_E_UNSUPPORTED) throw e
    g.noHistory = true
    counters.fallbackToJs++
  }
}
function hashIndex(g) {
  if (!g.byHash) {
    g.byHash = new Map()
    for (const i of g.applied) g.byHash.set(Buffer.from(g.hashes.buffer, g.hashes.byteOffset + 32 * i, 32).toString('hex'), i)
  }
  return g.byHash
}
function getAllChanges(backend) {   // new.js:1924-1927: BackendDoc.changes in application order (queued changes are not among them)
  isFrozenCheck(backend)
  noteGraphQuery(backend)
  const g = gpuHistory(backend, true)
  if (g) return Array.from(g.applied, i => g.changes[i])
  return ref().getAllChanges(hydrate(backend))
}
function getChanges(backend, haveDeps) {   // backend.js:152-157, new.js:1921-1976
  isFrozenCheck(backend)
  noteGraphQuery(backend)
  if (!Array.isArray(haveDeps)) throw new TypeError('Pass an array of hashes to Backend.getChanges()')
  const g = gpuHistory(backend, true)
  if (g) return changesSince(g, haveDeps).map(i => g.changes[i])
  return ref().getChanges(hydrate(backend), haveDeps)
}
function getChangeByHash(backend, hash) {   // new.js:1999-2002 (applied changes only: queued ones have no index)
  isFrozenCheck(backend)
  noteGraphQuery(backend)
  const g = gpuHistory(backend, true)
  if (g) { const i = hashIndex(g).get(hash); return i === undefined ? undefined : g.changes[i] }
  return ref().getChangeByHash(hydrate(backend), hash)
}
function getMissingDeps(backend, heads = []) {   // new.js:2014-2028
  isFrozenCheck(backend)
  noteGraphQuery(backend)
  const g = gpuHistory(backend, true)
  if (g && g.pendingIdx) return missingDeps(g, heads)
  return ref().getMissingDeps(hydrate(backend), heads)
}

// Backend.applyChanges(backend, changes) -> [backend', patch] (backend.js:27-31, new.js:1797-1879) on the engine: the state is empty
// (applyChanges(init(), changes): everything arrives in one batch) or was built by the engine, whose context then replays the
// earlier changes and the batch together and derives the incremental patch on the device (am355_apply_changes). What the engine
// refuses (an assignment to a list element, an edit inside an object the document no longer reaches ...) or rejects is served by
// the reference path, which returns the reference's patch or throws its exception.
// 0 .. n - 1 as a view of one shared array (application order of a state whose changes all applied in the order they were given)
const EMPTY_U32 = new Uint32Array(0)
let IDENTITY = new Uint32Array(0)
function identity(n) {
  if (n > IDENTITY.length) {
    IDENTITY = new Uint32Array(Math.max(2 * n, 1024))
    for (let i = 0; i < IDENTITY.length; i++) IDENTITY[i] = i
  }
  return IDENTITY.subarray(0, n)
}
// The hashes of the n changes of `state` (all applied in input order). States along one line of calls share a growing store: the
// state at its tip appends the batch's hashes (fetched alone), a state that is not the tip -- a fork -- starts a store of its own.
function hashesExtended(g, state, n) {
  const have = g && g.inOrder && g.hashes && g.hashStore && g.hashStore.used === g.hashes.length && g.hashes.length <= 32 * n ? g.hashes.length / 32 : -1
  let store = have >= 0 ? g.hashStore : null
  if (store && 32 * n <= store.buf.length) {
    if (n > have) store.buf.set(addon.hashes(ctx, have), 32 * have)
  } else {
    const all = addon.hashes(ctx)
    store = { buf: new Uint8Array(Math.max(64 * n, 4096)), used: 0 }
    store.buf.set(all, 0)
  }
  store.used = 32 * n
  state.hashStore = store
  return store.buf.subarray(0, 32 * n)
}

// (AM355_JS_PROFILE=1: where a served applyChanges call spends its time -- before the engine call, in it, fetch + materialise, the new state)
const PROFILE = !!process.env.AM355_JS_PROFILE
const now = () => Number(process.hrtime.bigint()) / 1e6
const profile = { calls: 0, before_ms: 0, engine_ms: 0, patch_ms: 0, state_ms: 0 }
function gpuApplyChanges(backend, changes) {
  const tpIn = PROFILE ? now() : 0
  const g = backend.state instanceof GpuState ? backend.state : null
  if (g && g.doc && !g.changes) {
    loadedHistory(g)   // a loaded document: its changes, rebuilt by the engine (am355_doc_changes)
    if (!g.changes) return null
  }
  if (g && !g.changes) return null
  let entry, handed = null   // handed: queued changes that are not in the context (see heldBack), given behind the batch
  if (g && g.doc) {
    // an unchanged loaded document: the engine applies the batch onto the DOCUMENT in its context (am355_apply_changes rebuilds the
    // hash graph as the reference's first applyChanges after load does, and knows that objectMeta came from the document's rows)
    if (g.fromChanges || !(entry = contextOf(g.generation))) {
      entry = acquireContext()
      addon.loadDocument(ctx, g.doc)
      addon.replay(ctx)
      g.generation = generation
      g.fromChanges = false
    }
  } else if (g) {
    if (!(entry = contextOf(g.generation))) {
      if (heldBack(g)) {
        gpuReplay(appliedOnly(g))
        addon.forgetCallHistory(ctx, g.docChanges || 0)
        addon.hashGraphKnown(ctx, g.graphKnown ? 1 : 0)
        handed = Array.from(g.pendingIdx, i => g.changes[i])
      } else replayRetained(g)
      entry = contextOf(generation)
    }
    if (!g.applied || !g.pendingIdx) { g.applied = addon.appliedOrder(ctx); g.pendingIdx = addon.pendingOrder(ctx) }
  } else {
    entry = acquireContext()
    addon.reset(ctx)
  }
  const docLineage = !!(g && (g.doc || g.fromDocument))
  if (docLineage && g.graphKnown) addon.hashGraphKnown(ctx, 1)
  const tp0 = PROFILE ? now() : 0
  try {
    addon.applyChanges(ctx, handed ? changes.concat(handed) : changes)
  } catch (e) {
    entry.generation = 0   // (whatever the context holds now is nobody's state)
    throw e
  }
  const tp1 = PROFILE ? now() : 0
  const patch = materialize(addon.fetchApplyIR(ctx, true))
  const tp2 = PROFILE ? now() : 0
  // the engine's list of changes: those applied so far in application order, the batch, those that were queued
  const list = !g ? changes.slice() : g.inOrder ? g.changes.concat(changes)
    : Array.from(g.applied, i => g.changes[i]).concat(changes, Array.from(g.pendingIdx, i => g.changes[i]))
  const state = new GpuState(list, null, patch.deps)
  entry.generation = ++generation
  state.generation = generation
  // The usual call -- everything applied in the order it was given, nothing queued -- needs none of the per-change tables copied out of
  // the context: the application order is 0 .. n - 1, the queue is empty, and the hashes of the earlier changes are the previous
  // state's (a call on a 4 k-change document spent 0.27 ms on these copies, more than the engine on the batch).
  state.inOrder = !handed && addon.appliedInInputOrder(ctx)
  if (state.inOrder) {
    state.applied = identity(list.length)
    state.pendingIdx = EMPTY_U32
    state.hashes = hashesExtended(g, state, list.length)
  } else {
    state.applied = addon.appliedOrder(ctx)
    state.pendingIdx = addon.pendingOrder(ctx)
    state.hashes = addon.hashes(ctx)
  }
  state.pending = patch.pendingChanges
  state.calls = g ? g.calls + 1 : 1
  state.fromDocument = !!(g && (g.fromDocument || g.doc))
  state.graphKnown = docLineage ? addon.hashGraphKnown(ctx, -1) : undefined   // (has this call made the reference rebuild the document's hash graph?)
  state.docChanges = g ? (g.doc ? g.changes.length : (g.docChanges || 0)) : 0   // the leading changes that are a loaded document's
  state.baseDoc = g ? (g.baseDoc || g.doc || null) : null
  state.batches = (g && g.batches ? g.batches : (g && g.doc && g.graphKnown ? [GRAPH_QUERY] : [])).concat([changes.slice()])
  counters.gpuApplyChanges++
  if (PROFILE) { const tp3 = now(); profile.calls++; profile.before_ms += tp0 - tpIn; profile.engine_ms += tp1 - tp0; profile.patch_ms += tp2 - tp1; profile.state_ms += tp3 - tp2 }
  return [{ state, heads: patch.deps }, patch]
}

function applyChanges(backend, changes) {
  isFrozenCheck(backend)
  const served = !JS_ONLY && Array.isArray(changes) && changes.length > 0 && changes.every(c => c instanceof Uint8Array) &&
    ((backend.state instanceof GpuState && !backend.state.js) || isEmptyRefState(backend))
  if (served) {
    try {
      const result = gpuApplyChanges(backend, changes)
      if (result) { backend.frozen = true; return result }
    } catch (e) {
      if (e.am355Code !== AM355_E_INVALID && e.am355Code !== AM355_E_UNSUPPORTED && !(e instanceof TypeError)) throw e
      counters.fallbackToJs++
    }
  }
  return ref().applyChanges(toJs(backend), changes)
}

function free(backend) {
  if (backend.state instanceof GpuState) { backend.state = null; backend.frozen = true } else ref().free(backend)
}

// ---------------------------------------------------------------------------------------------------------------------------
// Sync protocol (SURVEY.md 8f-4; reference backend/sync.js).  A message that carries changes is applied by applyChanges above --
// the bulk receive runs on the engine --, and what generateSyncMessage computes over the hash graph of an engine-built state is
// served from the engine too: the dependency graph by index (resolved on the device while the changes were hashed), the Bloom
// filter over the hashes of the changes since the last sync built on the device, filters received from the peer probed there.
// The protocol logic itself (what to send when, the bookkeeping about the peer, the wire encoding) is the reference's own sync.js,
// bound to this module by boundSync() below.
// ---------------------------------------------------------------------------------------------------------------------------
function hexOfHash(g, i) { return Buffer.from(g.hashes.buffer, g.hashes.byteOffset + 32 * i, 32).toString('hex') }

// the engine context that holds the replay of `g` (replayed again from the retained changes when it has moved on)
function ensureContext(g) {
  // (a loaded document sits in its context as a document: the hash graph needs the replay of its rebuilt changes)
  if ((g.doc && !g.fromChanges) || !contextOf(g.generation)) { replayRetained(g); contextOf(generation) }
}

// hash graph of an engine-built state, by index into g.changes: dependencies as resolved on the device, dependents in the order the
// reference pushes them (new.js:1853-1856: in application order, every change onto the lists of its dependencies)
function graphOf(g) {
  if (!g.graph) {
    ensureContext(g)
    const { depFirst, depIndex } = addon.depGraph(ctx)
    const applied = new Uint8Array(g.changes.length)
    for (const i of g.applied) applied[i] = 1
    const dependents = new Map()
    for (const i of g.applied) {
      if (!dependents.has(i)) dependents.set(i, [])
      for (let k = depFirst[i]; k < depFirst[i + 1]; k++) {
        const d = depIndex[k]
        if (!dependents.has(d)) dependents.set(d, [])
        dependents.get(d).push(i)
      }
    }
    g.graph = { depFirst, depIndex, applied, dependents }
  }
  return g.graph
}

// BackendDoc.getChanges(haveDeps) (new.js:1921-1976) as a list of indexes, in the order the reference returns the changes
function changesSince(g, haveDeps) {
  if (haveDeps.length === 0) return Array.from(g.applied)
  const idx = hashIndex(g), { depFirst, depIndex, dependents } = graphOf(g)
  let stack = [], seen = new Set(), toReturn = []
  for (const hash of haveDeps) {
    const i = idx.get(hash)
    if (i === undefined) throw new RangeError(`hash not found: ${hash}`)
    seen.add(i)
    stack.push(...dependents.get(i))
  }
  const depsSeen = i => { for (let k = depFirst[i]; k < depFirst[i + 1]; k++) if (!seen.has(depIndex[k])) return false; return true }
  while (stack.length > 0) {
    const i = stack.pop()
    seen.add(i)
    toReturn.push(i)
    if (!depsSeen(i)) break
    stack.push(...dependents.get(i))
  }
  const heads = g.heads.map(h => idx.get(h))
  if (stack.length === 0 && heads.every(h => seen.has(h))) return toReturn
  stack = haveDeps.map(h => idx.get(h))
  seen = new Set()
  while (stack.length > 0) {
    const i = stack.pop()
    if (!seen.has(i)) {
      for (let k = depFirst[i]; k < depFirst[i + 1]; k++) stack.push(depIndex[k])
      seen.add(i)
    }
  }
  return Array.from(g.applied).filter(i => !seen.has(i))
}

let refColumnar = null
function columnar() {
  if (!refColumnar) refColumnar = require(path.join(process.env.AUTOMERGE_BACKEND_PATH || 'automerge/backend', 'columnar'))
  return refColumnar
}

// BackendDoc.getMissingDeps(heads) (new.js:2014-2028) with a queue: the dependencies of the queued changes and the given heads that
// are neither applied nor queued
function missingDeps(g, heads) {
  const idx = hashIndex(g), allDeps = new Set(heads), inQueue = new Set()
  for (const i of g.pendingIdx) {
    inQueue.add(hexOfHash(g, i))
    for (const dep of columnar().decodeChangeMeta(g.changes[i], false).deps) allDeps.add(dep)
  }
  const missing = []
  for (const hash of allDeps) if (!idx.has(hash) && !inQueue.has(hash)) missing.push(hash)
  return missing.sort()
}

function leb32(out, v) { do { let b = v & 0x7f; v >>>= 7; if (v) b |= 0x80; out.push(b) } while (v) }
function readLeb32(bytes, pos) {
  let v = 0, shift = 0
  for (;;) {
    if (pos.i >= bytes.length) throw new RangeError('buffer ended with incomplete number')
    const b = bytes[pos.i++]
    v += (b & 0x7f) * Math.pow(2, shift)
    shift += 7
    if (!(b & 0x80)) break
    if (shift > 35) throw new RangeError('number out of range')
  }
  if (v > 0xffffffff) throw new RangeError('number out of range')
  return v
}

// makeBloomFilter (sync.js:234-238): the filter over the changes applied since `lastSync`, its bits set on the device
function makeBloomFilterGpu(g, lastSync) {
  const list = changesSince(g, lastSync)
  if (list.length === 0) return { lastSync, bloom: new Uint8Array(0) }
  ensureContext(g)
  const bits = addon.bloomBuild(ctx, Uint32Array.from(list))
  const head = []
  leb32(head, list.length); leb32(head, 10); leb32(head, 7)   // numEntries, BITS_PER_ENTRY, NUM_PROBES (sync.js:31, 66-74)
  const bloom = new Uint8Array(head.length + bits.length)
  bloom.set(head)
  bloom.set(bits, head.length)
  return { lastSync, bloom }
}

// getChangesToSend (sync.js:246-306) over indexes; the peer's filters are probed on the device
function changesToSendGpu(backend, g, have, need) {
  if (have.length === 0) return need.map(hash => getChangeByHash(backend, hash)).filter(change => change !== undefined)
  const lastSyncHashes = {}, filters = []
  for (const h of have) {
    for (const hash of h.lastSync) lastSyncHashes[hash] = true
    if (!(h.bloom instanceof Uint8Array)) throw new TypeError('invalid argument')
    if (h.bloom.byteLength === 0) filters.push({ numEntries: 0, numBitsPerEntry: 0, numProbes: 0, bits: h.bloom })
    else {
      const pos = { i: 0 }
      const numEntries = readLeb32(h.bloom, pos), numBitsPerEntry = readLeb32(h.bloom, pos), numProbes = readLeb32(h.bloom, pos)
      const n = Math.ceil(numEntries * numBitsPerEntry / 8)
      if (pos.i + n > h.bloom.length) throw new RangeError('subarray exceeds buffer size')
      filters.push({ numEntries, numBitsPerEntry, numProbes, bits: h.bloom.subarray(pos.i, pos.i + n) })
    }
  }
  const list = changesSince(g, Object.keys(lastSyncHashes)), { depFirst, depIndex } = graphOf(g)
  ensureContext(g)
  const listArr = Uint32Array.from(list)
  const inSome = new Uint8Array(list.length)
  for (const f of filters) {
    if (f.numEntries === 0) continue
    const flags = addon.bloomProbe(ctx, listArr, f.numEntries, f.numBitsPerEntry, f.numProbes, f.bits)
    for (let k = 0; k < list.length; k++) inSome[k] |= flags[k]
  }
  const inList = new Set(list), dependents = new Map(), toSend = new Set()
  list.forEach((i, k) => {
    for (let q = depFirst[i]; q < depFirst[i + 1]; q++) {
      const d = depIndex[q]
      if (!dependents.has(d)) dependents.set(d, [])
      dependents.get(d).push(i)
    }
    if (!inSome[k]) toSend.add(i)
  })
  const stack = Array.from(toSend)
  while (stack.length > 0) {
    const i = stack.pop()
    for (const d of dependents.get(i) || []) if (!toSend.has(d)) { toSend.add(d); stack.push(d) }
  }
  const idx = hashIndex(g), out = []
  for (const hash of need) {
    const i = idx.get(hash)
    if (i !== undefined) toSend.add(i)
    if (i === undefined || !inList.has(i)) { const change = getChangeByHash(backend, hash); if (change) out.push(change) }
  }
  for (const i of list) if (toSend.has(i)) { engineHashOf.set(g.changes[i], hexOfHash(g, i)); out.push(g.changes[i]) }
  return out
}

// The protocol logic is the REFERENCE'S OWN backend/sync.js, not a restatement of it: a private instance of that file is compiled
// with THIS module as its './backend' (so its Backend.getHeads / getMissingDeps / getChangeByHash / getChanges / applyChanges calls
// land on the functions above -- the bulk receive runs am355_apply_changes) and with two of its module-local functions rebound:
// makeBloomFilter (sync.js:234-238) and getChangesToSend (sync.js:246-306) go to the device paths above when the state is the
// engine's. Its './columnar' is the reference's own with one shortcut: decodeChangeMeta(change, true) of a change the engine
// hashed takes the hash from the engine instead of running SHA-256 in JS again (the reference's own TODO, sync.js:375-377).
// Nothing is written to disk and the instance does not enter require.cache: the installed package keeps its own sync.js.
let boundSyncModule = null
const engineHashOf = new WeakMap()   // change buffer (as retained by a GpuState) -> hex hash the device computed
function boundSync() {
  if (boundSyncModule) return boundSyncModule
  const Module = require('module'), fs = require('fs')
  const dir = path.dirname(require.resolve(process.env.AUTOMERGE_BACKEND_PATH || 'automerge/backend'))
  const file = path.join(dir, 'sync.js')
  const rebind = '\n;module.exports.__rebind = h => { makeBloomFilter = h.makeBloomFilter(makeBloomFilter); getChangesToSend = h.getChangesToSend(getChangesToSend) }\n'
  const realColumnar = columnar()
  const columnarForSync = Object.assign(Object.create(realColumnar), {
    decodeChangeMeta(change, computeHash) {
      const known = computeHash ? engineHashOf.get(change) : undefined
      if (known === undefined) return realColumnar.decodeChangeMeta(change, computeHash)
      const meta = realColumnar.decodeChangeMeta(change, false)
      meta.hash = known
      return meta
    }
  })
  const m = new Module(file, module)
  m.filename = file
  m.paths = Module._nodeModulePaths(dir)
  m.require = function (id) {
    if (id === './backend') return module.exports
    if (id === './columnar') return columnarForSync
    return Module.prototype.require.call(this, id)
  }
  m._compile(fs.readFileSync(file, 'utf8') + rebind, file)
  m.loaded = true
  const engineState = backend => { const g = gpuHistory(backend, true); return g && g.pendingIdx ? g : null }
  m.exports.__rebind({
    makeBloomFilter: original => (backend, lastSync) => {
      const g = engineState(backend)
      return g ? makeBloomFilterGpu(g, lastSync) : original(backend, lastSync)
    },
    getChangesToSend: original => (backend, have, need) => {
      const g = engineState(backend)
      return g ? changesToSendGpu(backend, g, have, need) : original(backend, have, need)
    }
  })
  boundSyncModule = m.exports
  return boundSyncModule
}
const syncModule = () => (JS_ONLY ? ref() : boundSync())

const delegate1 = name => (backend, ...args) => ref()[name](toJs(backend), ...args)

module.exports = {
  init, load, loadChanges, getPatch, getHeads, free, save, getAllChanges, getChanges, getChangeByHash, getMissingDeps,
  clone: backend => { noteGraphQuery(backend); return ref().clone(hydrate(backend)) },
  applyChanges,
  applyLocalChange: delegate1('applyLocalChange'),
  getChangesAdded: (b1, b2) => { noteGraphQuery(b1); return ref().getChangesAdded(hydrate(b1), hydrate(b2)) },
  // sync protocol: the reference's own backend/sync.js bound to this module (boundSync above)
  generateSyncMessage: (...a) => syncModule().generateSyncMessage(...a),
  receiveSyncMessage: (...a) => syncModule().receiveSyncMessage(...a),
  encodeSyncMessage: (...a) => syncModule().encodeSyncMessage(...a),
  decodeSyncMessage: (...a) => syncModule().decodeSyncMessage(...a),
  encodeSyncState: (...a) => syncModule().encodeSyncState(...a),
  decodeSyncState: (...a) => syncModule().decodeSyncState(...a),
  initSyncState: (...a) => syncModule().initSyncState(...a),
  // engine statistics of the last GPU replay (not part of the reference surface)
  _engineStats: () => (addon ? addon.stats(ctx) : null),
  _counters: counters,
  _applyProfile: profile,
  _hydrate: hydrate   // (tests: the reference handle of an engine state, made the way the state came to be)
}
