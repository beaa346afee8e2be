// End-to-end timing through the JS host (node -> N-API addon -> GPU): SURVEY.md §8d "T_replay" (change Uint8Arrays in host
// memory -> patch IR in host memory) and "T_e2e" (through the materialised JS patch object, materialize.js). For comparison
// the older route through JSON text (engine renders JSON, JSON.parse builds the object) is timed too. Reported in DESIGN.md §8.
//
//   python -c "from automerge_classic_amd import loggen; loggen.config('c4_text_single', 1.0, False).save('/tmp/c4.bin')"
//   node automerge_classic_amd/js/bench_e2e.js /tmp/c4.bin [reps]
//
// Log file layout (little endian): u32 n_changes, u64 n_ops, u64 offsets[n+1], arena.
'use strict'
const fs = require('fs')
const path = require('path')
const addon = require(path.join(__dirname, 'am355_napi.node'))
const { materialize } = require('./materialize.js')

const buf = fs.readFileSync(process.argv[2])
const reps = parseInt(process.argv[3] || '7')
const n = buf.readUInt32LE(0)
const nOps = Number(buf.readBigUInt64LE(4))
const offs = []
for (let i = 0; i <= n; i++) offs.push(Number(buf.readBigUInt64LE(12 + 8 * i)))
const base = 12 + 8 * (n + 1)
const changes = []
for (let i = 0; i < n; i++) changes.push(new Uint8Array(buf.buffer, buf.byteOffset + base + offs[i], offs[i + 1] - offs[i]))

const ctx = addon.create(parseInt(process.env.MI355X_DEVICE || '0'))
const ms = () => Number(process.hrtime.bigint()) / 1e6
const t = { stage: [], replay: [], fetch: [], materialize: [], json: [], parse: [], save: [] }
let jsonLen = 0, docLen = 0
for (let r = 0; r < reps + 2; r++) {
  const t0 = ms()
  addon.loadChanges(ctx, changes)   // host inflate + copy to HBM
  const t1 = ms()
  addon.replay(ctx)                 // the hot path
  const t2 = ms()
  const ir = addon.fetchIR(ctx, true)   // patch IR (record tables) -> host -> the context's own JS ArrayBuffers (as index.js asks for them)
  const t2a = ms()
  const patchIR = materialize(ir)   // the object the frontend consumes
  const t2b = ms()
  const text = addon.patchJSON(ctx) // (older route) JSON.stringify-identical text rendered by the engine
  const t3 = ms()
  const patch = JSON.parse(text)
  const t4 = ms()
  const doc = addon.save(ctx, 0)
  const t5 = ms()
  if (r === 0 && JSON.stringify(patchIR) !== text) throw new Error('materialised patch != engine JSON text')
  if (r >= 2) { t.stage.push(t1 - t0); t.replay.push(t2 - t1); t.fetch.push(t2a - t2); t.materialize.push(t2b - t2a); t.json.push(t3 - t2b); t.parse.push(t4 - t3); t.save.push(t5 - t4) }
  jsonLen = text.length; docLen = doc.length
  if (!patch.diffs) throw new Error('no patch')
}
const med = a => a.slice().sort((x, y) => x - y)[Math.floor(a.length / 2)]
const m = { stage: med(t.stage), replay: med(t.replay), fetch: med(t.fetch), materialize: med(t.materialize), json: med(t.json), parse: med(t.parse), save: med(t.save) }
const tReplay = m.stage + m.replay + m.fetch, tE2e = tReplay + m.materialize
const tE2eJson = m.stage + m.replay + m.json + m.parse
console.log(JSON.stringify({
  n_ops: nOps, n_changes: n, reps, patch_json_bytes: jsonLen, saved_doc_bytes: docLen,
  ms: m, T_replay_ms: tReplay, T_replay_ops_per_s: nOps / (tReplay / 1e3), T_e2e_ms: tE2e, T_e2e_ops_per_s: nOps / (tE2e / 1e3),
  T_e2e_via_json_text_ms: tE2eJson,
  engine: addon.stats(ctx)
}))
