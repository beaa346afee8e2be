// The per-state tables index.js keeps for states made by Backend.applyChanges call after call -- the list of changes, their application
// order, their hashes (a view of a store shared along the line of calls, index.js hashesExtended), the mirrored arena of the binding
// (am355_napi.c arena_table) -- against what follows from the change bytes alone:
//   * Backend.getAllChanges(state) is the input list, Backend.getChangeByHash finds every change by the SHA-256 of its chunk
//     (columnar.js:693-705), Backend.getHeads is the hash of the last change of a linear log;
//   * the patch of every call (values are ranges of the mirrored arena) equals the patch the same call gets from a context that was
//     given nothing but this state's changes in ONE earlier call (fresh mirror, full copies).
//   node test_apply_tables.js <log file: u32 n, u64 n_ops, u64 offsets[n+1], arena>
'use strict'
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const Backend = require(path.join(__dirname, 'index.js'))

const buf = fs.readFileSync(process.argv[2])
const n = buf.readUInt32LE(0)
const offs = []
for (let i = 0; i <= n; i++) offs.push(Number(buf.readBigUInt64LE(12 + 8 * i)))
const base = 12 + 8 * (n + 1)
const changes = []
for (let i = 0; i < n; i++) changes.push(new Uint8Array(buf.buffer, buf.byteOffset + base + offs[i], offs[i + 1] - offs[i]))
const hashOf = c => crypto.createHash('sha256').update(c.subarray(8)).digest('hex')
const empty = () => { try { return Backend.init() } catch (e) { return { state: { changes: [], queue: [] }, heads: [] } } }

const head = Math.max(2, n >> 2)
let [state] = Backend.applyChanges(empty(), changes.slice(0, head))
let k = head, size = 1, calls = 0
while (k < n) {
  const batch = changes.slice(k, k + size)
  // the same call from a state built in one go (its context has a fresh mirror of everything)
  const [fresh] = Backend.applyChanges(empty(), changes.slice(0, k))
  const [, want] = Backend.applyChanges(fresh, batch)
  const [next, got] = Backend.applyChanges(state, batch)
  if (JSON.stringify(got) !== JSON.stringify(want)) throw new Error(`call ${calls}: patch differs from the one-go state's`)
  state = next
  k += batch.length
  size = size % 3 + 1
  calls++
  if (calls % 4 === 0 || k >= n) {
    const all = Backend.getAllChanges(state)
    if (all.length !== k) throw new Error(`call ${calls}: getAllChanges has ${all.length} changes, ${k} given`)
    for (let i = 0; i < k; i++) if (all[i] !== changes[i]) throw new Error(`call ${calls}: getAllChanges[${i}] is not input change ${i}`)
    for (let i = 0; i < k; i += Math.max(1, k >> 4)) {
      const c = Backend.getChangeByHash(state, hashOf(changes[i]))
      if (c !== changes[i]) throw new Error(`call ${calls}: getChangeByHash(hash of change ${i}) found ${c === undefined ? 'nothing' : 'another change'}`)
    }
    if (Backend.getChangeByHash(state, hashOf(changes[k - 1])) !== changes[k - 1]) throw new Error(`call ${calls}: the last change is not found by its hash`)
  }
}
const c = Backend._counters
if (c.fallbackToJs) throw new Error('a call was served by the JS fallback: ' + JSON.stringify(c))
console.log(JSON.stringify({ ok: true, changes: n, calls, counters: c }))
