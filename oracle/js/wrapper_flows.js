// TEST INFRASTRUCTURE (build container: needs the reference tree). The reference FRONTEND over mi355x-backend with the engine
// enabled: flows in which a GPU-built state meets read-only and mutating reference calls (ADVICE r1: read-only calls must not
// freeze the handle -- backend.js:12-14, 93-98 -- mutating ones must).
//   LD_PRELOAD=tests/emu/libam355_emu.so NODE_PATH=oracle/js_shims/node_modules AUTOMERGE_BACKEND_PATH=/root/reference/backend node oracle/js/wrapper_flows.js
const path = require('path')
const assert = require('assert')
const REF = process.env.AUTOMERGE_REF || '/root/reference'
const Automerge = require(path.join(REF, 'src', 'automerge'))
const Backend = require(path.join(__dirname, '..', '..', 'automerge_classic_amd', 'js', 'index.js'))
const RefBackend = require(path.join(REF, 'backend'))

function build(backend) {
  Automerge.setDefaultBackend(backend)
  let a = Automerge.from({ text: new Automerge.Text('hello'), list: [1, 2, 3], n: NaN, inf: Infinity, bytes: new Uint8Array([1, 2, 255]), c: new Automerge.Counter(3) }, 'aaaa')
  let b = Automerge.merge(Automerge.init('bbbb'), a)
  a = Automerge.change(a, d => { d.text.insertAt(5, ' ', 'w'); d.list.push(4); d.c.increment(2) })
  b = Automerge.change(b, d => { d.text.insertAt(0, '>'); d.list.deleteAt(0); d.m = { x: 1 } })
  return [a, b]
}

// documents made with the reference backend, then everything below through mi355x-backend
const [ra, rb] = build(RefBackend)
const bytesA = Automerge.save(ra), bytesB = Automerge.save(rb)
Automerge.setDefaultBackend(Backend)
const before = Object.assign({}, Backend._counters)

// load -> read-only history call -> change (the handle must still be usable)
let f = Automerge.load(bytesA)
assert.strictEqual(Automerge.getAllChanges(f).length, Automerge.getAllChanges(ra).length)
f = Automerge.change(f, d => { d.after = true })
assert.strictEqual(f.after, true)
assert.ok(Number.isNaN(f.n) && f.inf === Infinity, 'non-finite float64 values survive the GPU load path')
assert.deepStrictEqual(Object.assign({}, f.bytes), { 0: 1, 1: 2, 2: 255 })   // (the frontend stores a Uint8Array as a map object)

// history of a loaded document, rebuilt by the engine (am355_doc_changes): the reference's own bytes, change by change
{
  const mine = Automerge.getAllChanges(Automerge.load(bytesB))
  Automerge.setDefaultBackend(RefBackend)
  const theirs = Automerge.getAllChanges(Automerge.load(bytesB))
  Automerge.setDefaultBackend(Backend)
  assert.strictEqual(mine.length, theirs.length)
  for (let i = 0; i < mine.length; i++) assert.ok(Buffer.from(mine[i]).equals(Buffer.from(theirs[i])), `change ${i} of the rebuilt history`)
  assert.ok(Backend._counters.gpuHistory - before.gpuHistory >= 2, 'history of loaded documents must be served by the engine')
}

// merge of two loaded documents (getChangesAdded on both, applyChanges on one)
let g1 = Automerge.load(bytesA), g2 = Automerge.load(bytesB)
const merged = Automerge.merge(g1, g2)
Automerge.setDefaultBackend(RefBackend)
const want = Automerge.merge(Automerge.load(bytesA), Automerge.load(bytesB))
Automerge.setDefaultBackend(Backend)
assert.deepStrictEqual(JSON.parse(JSON.stringify(merged)), JSON.parse(JSON.stringify(want)))
assert.strictEqual(merged.text.toString(), want.text.toString())

// clone keeps the original usable
let h = Automerge.load(bytesB)
const h2 = Automerge.clone(h)
h = Automerge.change(h, d => { d.z = 1 })
assert.strictEqual(h.z, 1)
assert.strictEqual(h2.z, undefined)

// save of a loaded (unchanged) document returns the bytes; save after a change goes through the JS path
assert.deepStrictEqual(Automerge.save(Automerge.load(bytesA)), bytesA)
assert.ok(Automerge.save(h).length > 0)

// sync between a GPU-loaded document and a fresh peer
let n1 = Automerge.load(bytesA), n2 = Automerge.init('cccc')
let s1 = Automerge.initSyncState(), s2 = Automerge.initSyncState()
for (let i = 0; i < 10; i++) {
  let msg
  ;[s1, msg] = Automerge.generateSyncMessage(n1, s1)
  if (msg) [n2, s2] = Automerge.receiveSyncMessage(n2, s2, msg)
  let msg2
  ;[s2, msg2] = Automerge.generateSyncMessage(n2, s2)
  if (msg2) [n1, s1] = Automerge.receiveSyncMessage(n1, s1, msg2)
  if (!msg && !msg2) break
}
assert.strictEqual(n2.text.toString(), n1.text.toString())
n1 = Automerge.change(n1, d => { d.done = true })   // n1's handle survived the sync exchange

// bulk history: applyChanges of all changes onto a fresh document, and getHistory snapshots
const all = Automerge.getAllChanges(ra)
let [fresh] = Automerge.applyChanges(Automerge.init(), all)
assert.strictEqual(fresh.text.toString(), ra.text.toString())
const history = Automerge.getHistory(Automerge.load(bytesA))
assert.strictEqual(history.length, all.length)
assert.strictEqual(history[history.length - 1].snapshot.text.toString(), ra.text.toString())   // snapshot = Backend.loadChanges(init(), changes[0..i])

const served = {}
for (const k of Object.keys(Backend._counters)) served[k] = Backend._counters[k] - before[k]
assert.ok(served.gpuLoad >= 7 && served.gpuLoadChanges >= 1, 'Backend.load must be served by the engine: ' + JSON.stringify(served))
console.log('wrapper flows ok; served by ' + JSON.stringify(served))

// several GPU-built states alive at once: Backend.save of an older one is served from the context that still holds its replay
{
  const c0 = Object.assign({}, Backend._counters)
  const empty = () => ({ state: { changes: [], queue: [] }, heads: [] })
  const s1 = Backend.loadChanges(empty(), Automerge.getAllChanges(ra))
  const s2 = Backend.loadChanges(empty(), Automerge.getAllChanges(rb))
  const b1 = Backend.save(s1), b2 = Backend.save(s2), b1again = Backend.save(s1)
  assert.deepStrictEqual(Buffer.from(b1), Buffer.from(RefBackend.save(RefBackend.loadChanges(RefBackend.init(), Automerge.getAllChanges(ra)))))
  assert.deepStrictEqual(Buffer.from(b2), Buffer.from(RefBackend.save(RefBackend.loadChanges(RefBackend.init(), Automerge.getAllChanges(rb)))))
  assert.deepStrictEqual(Buffer.from(b1again), Buffer.from(b1))
  assert.strictEqual(Backend._counters.gpuSave - c0.gpuSave, 3)
  assert.strictEqual(Backend._counters.saveReplays - c0.saveReplays, 0, 'no silent re-replay while the context pool still holds the state')
  console.log('context pool ok')
}
