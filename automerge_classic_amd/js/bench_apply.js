// Backend.applyChanges call after call through the JS host (index.js -> N-API addon -> GPU): what an application that receives changes
// one at a time pays per call, with the pieces of a call timed apart (AM355_JS_PROFILE=1 makes index.js record them).
//
//   python -c "from automerge_classic_amd import loggen; loggen.config('c4_text_single', 1.0, False).save('/tmp/c4.bin')"
//   node automerge_classic_amd/js/bench_apply.js /tmp/c4.bin [changes per call = 1] [calls = 40]
//   AM355_JS_ONLY=1 node ... : the reference's own Backend (build container: needs the reference tree)
'use strict'
const fs = require('fs')
const path = require('path')
const Backend = require(path.join(__dirname, 'index.js'))

const buf = fs.readFileSync(process.argv[2])
const per = parseInt(process.argv[3] || '1'), calls = parseInt(process.argv[4] || '40')
const n = buf.readUInt32LE(0)
const offs = []
for (let i = 0; i <= n; i++) offs.push(Number(buf.readBigUInt64LE(12 + 8 * i)))
const base = 12 + 8 * (n + 1)
const changes = []
for (let i = 0; i < n; i++) changes.push(new Uint8Array(buf.buffer, buf.byteOffset + base + offs[i], offs[i + 1] - offs[i]))

const ms = () => Number(process.hrtime.bigint()) / 1e6
const first = n - per * calls
let t0 = ms()
// (the GPU box has no reference package for Backend.init(): an empty state of the shape index.js recognises stands in for it)
let empty
try { empty = Backend.init() } catch (e) { empty = { state: { changes: [], queue: [] }, heads: [] } }
let [state] = Backend.applyChanges(empty, changes.slice(0, first))
const tBase = ms() - t0
const baseProfile = Backend._applyProfile ? Object.assign({}, Backend._applyProfile) : undefined
if (Backend._applyProfile) for (const k of Object.keys(Backend._applyProfile)) Backend._applyProfile[k] = 0   // (the calls below only)
const times = []
let edits = 0
for (let j = 0; j < calls; j++) {
  const batch = changes.slice(first + j * per, first + (j + 1) * per)
  t0 = ms()
  const [next, patch] = Backend.applyChanges(state, batch)
  times.push(ms() - t0)
  state = next
  if (!patch || !patch.diffs) throw new Error('no patch')
  edits += JSON.stringify(patch.diffs).length
}
const later = times.slice(1).sort((a, b) => a - b)
console.log(JSON.stringify({
  n_changes: n, changes_per_call: per, calls, base_ms: tBase, first_call_ms: times[0], median_ms: later[Math.floor(later.length / 2)], min_ms: later[0],
  patch_text_bytes_per_call: Math.round(edits / calls), counters: Backend._counters, profile: Backend._applyProfile, base_profile: baseProfile
}))
