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
    if (g.doc) g.js = ref().load(g.doc)
    else if (HYDRATE_FROM_DOC && !JS_ONLY) {
      // hydrate the JS BackendDoc from the engine's save() bytes: Backend.load of a document is several times cheaper in JS than
      // replaying every change (opt-in: MI355X_HYDRATE=doc; the default replays the retained changes, exact by construction)
      let bytes = null
      try {
        if (!contextOf(g.generation)) { gpuReplay(g.changes); g.generation = generation }
        bytes = addon.save(ctx, 0)
      } catch (e) {
        if (e.am355Code !== AM355_E_INVALID && e.am355Code !== AM355_E_UNSUPPORTED) throw e
      }
      g.js = bytes ? ref().load(bytes) : ref().loadChanges(ref().init(), g.changes)
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
  return PATCH_VIA_JSON ? JSON.parse(addon.patchJSON(ctx)) : materialize(addon.fetchIR(ctx))
}

function gpuReplay(changes) {
  acquireContext()
  addon.loadChanges(ctx, changes)
  addon.replay(ctx)
  return gpuPatch()
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
      else { g.patch = gpuReplay(g.changes); g.generation = generation }
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
      addon.loadDocument(ctx, data)
      addon.replay(ctx)
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
      if (!contextOf(g.generation)) { counters.saveReplays++; gpuReplay(g.changes); g.generation = generation }
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
function gpuHistory(backend) {
  const g = backend.state
  if (JS_ONLY || !(g instanceof GpuState)) return null
  if (g.doc && !g.changes && !g.noHistory) loadedHistory(g)
  return g.changes && g.applied && g.pending === 0 ? g : null
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
function getAllChanges(backend) {   // new.js:1924-1927: BackendDoc.changes in application order
  isFrozenCheck(backend)
  const g = gpuHistory(backend)
  if (g) return Array.from(g.applied, i => g.changes[i])
  return ref().getAllChanges(hydrate(backend))
}
function getChanges(backend, haveDeps) {
  isFrozenCheck(backend)
  const g = gpuHistory(backend)
  if (g && Array.isArray(haveDeps) && haveDeps.length === 0) return Array.from(g.applied, i => g.changes[i])
  return ref().getChanges(hydrate(backend), haveDeps)
}
function getChangeByHash(backend, hash) {   // new.js:1999-2002
  isFrozenCheck(backend)
  const g = gpuHistory(backend)
  if (g) { const i = hashIndex(g).get(hash); return i === undefined ? undefined : g.changes[i] }
  return ref().getChangeByHash(hydrate(backend), hash)
}
function getMissingDeps(backend, heads = []) {   // new.js:2014-2028 with an empty queue: the given heads we do not have
  isFrozenCheck(backend)
  const g = gpuHistory(backend)
  if (g) { const idx = hashIndex(g); return Array.from(new Set(heads)).filter(h => !idx.has(h)).sort() }
  return ref().getMissingDeps(hydrate(backend), heads)
}

// Backend.applyChanges(backend, changes) -> [backend', patch] (backend.js:27-31, new.js:1797-1879) on the engine: the state is empty
// (applyChanges(init(), changes): everything arrives in one batch) or was built by the engine, whose context then replays the
// earlier changes and the batch together and derives the incremental patch on the device (am355_apply_changes). What the engine
// refuses (an assignment to a list element, an edit inside an object the document no longer reaches ...) or rejects is served by
// the reference path, which returns the reference's patch or throws its exception.
function gpuApplyChanges(backend, changes) {
  const g = backend.state instanceof GpuState ? backend.state : null
  if (g && g.doc && !g.changes) {
    loadedHistory(g)   // a loaded document: its changes, rebuilt by the engine (am355_doc_changes)
    if (!g.changes) return null
  }
  if (g && !g.changes) return null
  let entry
  if (g) {
    if (g.doc || !(entry = contextOf(g.generation))) { gpuReplay(g.changes); g.generation = generation; entry = contextOf(generation) }
    if (!g.applied || !g.pendingIdx) { g.applied = addon.appliedOrder(ctx); g.pendingIdx = addon.pendingOrder(ctx) }
  } else {
    entry = acquireContext()
    addon.reset(ctx)
  }
  try {
    addon.applyChanges(ctx, changes)
  } catch (e) {
    entry.generation = 0   // (whatever the context holds now is nobody's state)
    throw e
  }
  const patch = materialize(addon.fetchApplyIR(ctx))
  // the engine's list of changes: those applied so far in application order, the batch, those that were queued
  const list = g ? Array.from(g.applied, i => g.changes[i]).concat(changes, Array.from(g.pendingIdx, i => g.changes[i])) : changes.slice()
  const state = new GpuState(list, null, patch.deps)
  entry.generation = ++generation
  state.generation = generation
  state.applied = addon.appliedOrder(ctx)
  state.pendingIdx = addon.pendingOrder(ctx)
  state.hashes = addon.hashes(ctx)
  state.pending = patch.pendingChanges
  counters.gpuApplyChanges++
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

// backend/sync.js:420-473: the handle is only replaced (and the old one frozen) when the message carried changes; a message
// without changes returns the handle it was given, which the frontend keeps using (src/automerge.js receiveSyncMessage)
function receiveSyncMessage(backend, syncState, msg) {
  const handle = hydrate(backend)
  const result = ref().receiveSyncMessage(handle, syncState, msg)
  if (backend.state instanceof GpuState) {
    if (result[0] === handle) result[0] = backend
    else { backend.state.js = null; backend.frozen = true }
  }
  return result
}

const delegate1 = name => (backend, ...args) => ref()[name](toJs(backend), ...args)

module.exports = {
  init, load, loadChanges, getPatch, getHeads, free, save, getAllChanges, getChanges, getChangeByHash, getMissingDeps,
  clone: backend => ref().clone(hydrate(backend)),
  applyChanges,
  applyLocalChange: delegate1('applyLocalChange'),
  getChangesAdded: (b1, b2) => ref().getChangesAdded(hydrate(b1), hydrate(b2)),
  // sync protocol: unchanged reference code operating on JS handles (backend/sync.js:20 binds the JS backend)
  generateSyncMessage: (backend, syncState) => ref().generateSyncMessage(hydrate(backend), syncState),
  receiveSyncMessage,
  encodeSyncMessage: (...a) => ref().encodeSyncMessage(...a),
  decodeSyncMessage: (...a) => ref().decodeSyncMessage(...a),
  encodeSyncState: (...a) => ref().encodeSyncState(...a),
  decodeSyncState: (...a) => ref().decodeSyncState(...a),
  initSyncState: (...a) => ref().initSyncState(...a),
  // engine statistics of the last GPU replay (not part of the reference surface)
  _engineStats: () => (addon ? addon.stats(ctx) : null),
  _counters: counters
}
