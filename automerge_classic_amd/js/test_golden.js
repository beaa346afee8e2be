// Runs on the GPU box (node + addon + MI355X, no reference tree needed): every golden fixture produced by the
// unmodified reference must be reproduced byte-for-byte through the JS Backend surface of mi355x-backend.
//   node automerge_classic_amd/js/test_golden.js tests/golden
'use strict'
const fs = require('fs')
const path = require('path')
const Backend = require('./index.js')

const dir = process.argv[2] || path.join(__dirname, '..', '..', 'tests', 'golden')
let failed = 0, n = 0
const crypto = require('crypto')
const historyGolden = JSON.parse(fs.readFileSync(path.join(dir, 'doc_history.json'), 'utf8')).fixtures

// Backend.getAllChanges(Backend.load(doc)) against the unmodified reference's result (digests, oracle/make_history_golden.py);
// documents on which the reference throws are left to the reference path, which is not present on the GPU box
function checkHistory(f, loaded) {
  const want = historyGolden[f.replace(/\.json$/, '')]
  if (!want || want.error) return
  n++
  const before = Backend._counters.gpuHistory
  const all = Backend.getAllChanges(loaded)
  const sum = crypto.createHash('sha256')
  for (const c of all) { const len = Buffer.alloc(4); len.writeUInt32LE(c.byteLength); sum.update(len); sum.update(c) }
  let ok = all.length === want.n_changes && sum.digest('hex') === want.changes_sha256 && Backend._counters.gpuHistory === before + 1
  for (const h of want.heads) ok = ok && Backend.getChangeByHash(loaded, h) !== undefined
  ok = ok && Backend.getMissingDeps(loaded, want.heads.concat(['00'.repeat(32)])).length === 1
  ok = ok && Backend.getChanges(loaded, []).length === want.n_changes
  if (!ok) { failed++; console.error(`FAIL ${f}: history of the loaded document`) } else console.log(`ok   ${f}  (history of the loaded document, ${all.length} changes)`)
}
for (const f of fs.readdirSync(dir).filter(f => f.endsWith('.json')).sort()) {
  const fx = JSON.parse(fs.readFileSync(path.join(dir, f), 'utf8'))
  if (f === 'list_quirks.json') {
    // counters / visible rows without a value inside lists (reference patches: oracle/js/make_list_quirk_golden.js): `remove` edits and
    // counter totals in whole-document patches, through loadChanges and through load of the reference's saved document
    let ok = 0, refusedCases = []
    for (const c of fx.cases) {
      const changes = c.changes.map(x => new Uint8Array(Buffer.from(x, 'base64')))
      try {
        const st = Backend.loadChanges({ state: { changes: [], queue: [] }, heads: [] }, changes)
        const a = JSON.stringify(Backend.getPatch(st)) === c.patch
        const b = JSON.stringify(Backend.getPatch(Backend.load(new Uint8Array(Buffer.from(c.doc, 'base64'))))) === c.load_patch
        if (a && b) ok++; else { failed++; console.error(`FAIL ${f} ${c.name}: ${a ? '' : 'replay '}${b ? '' : 'load'}`) }
      } catch (e) {
        if (c.name === 'hand_increment_deleted') refusedCases.push(c.name)   // (the one case left to the reference path, DESIGN.md 5: not present here)
        else { failed++; console.error(`FAIL ${f} ${c.name}: ${e.message}`) }
      }
    }
    n++
    console.log(`ok   ${f}  (${ok} of ${fx.cases.length} cases, left to the reference path: ${refusedCases.join(' ') || 'none'})`)
    if (ok !== fx.cases.length - 1) { failed++; console.error(`FAIL ${f}: ${ok} cases equal`) }
    continue
  }
  if (!fx.changes && !fx.doc) continue   // not a patch fixture (e.g. digests of generated workloads)
  if (!fx.changes) {
    // document-only fixture: Backend.load + getPatch
    const want = fx.stock_equals_bigblock === false ? fx.load_patch_bigblock : fx.load_patch
    const loaded = Backend.load(new Uint8Array(Buffer.from(fx.doc, 'base64')))
    n++
    if (JSON.stringify(Backend.getPatch(loaded)) !== want) { failed++; console.error(`FAIL ${f}: document load`) } else console.log(`ok   ${f}  (document load, ${fx.rows} rows)`)
    continue
  }
  const changes = fx.changes.map(c => new Uint8Array(Buffer.from(c, 'base64')))
  const expected = fx.stock_equals_bigblock === false ? fx.patch_bigblock : fx.patch
  // the handle given to loadChanges must look like a fresh reference state: {state: {changes: [], queue: []}, heads: []}
  const empty = { state: { changes: [], queue: [] }, heads: [] }
  const state = Backend.loadChanges(empty, changes)
  const got = JSON.stringify(Backend.getPatch(state))
  n++
  if (got !== expected) { failed++; console.error(`FAIL ${f}`) } else console.log(`ok   ${f}  (${changes.length} changes)`)
  if (!empty.frozen) { failed++; console.error(`FAIL ${f}: old handle not frozen`) }
  if (JSON.stringify(Backend.getHeads(state)) !== JSON.stringify(JSON.parse(expected).deps)) { failed++; console.error(`FAIL ${f}: heads`) }
  if (JSON.parse(got).pendingChanges === 0) {
    // history queries served from the engine's application order: every applied change once, each findable by its hash
    // (with queued changes they go to the reference path, which is not present on the GPU box)
    const all = Backend.getAllChanges(state)
    const inputs = new Set(changes.map(c => Buffer.from(c).toString('base64')))
    const uniq = new Set(all.map(c => Buffer.from(c).toString('base64')))
    const patch = JSON.parse(got)
    const applied = Object.values(patch.clock).reduce((a, b) => a + b, 0)
    n++
    let ok = all.length === applied && uniq.size === all.length && all.every(c => inputs.has(Buffer.from(c).toString('base64')))
    for (const h of patch.deps) ok = ok && Backend.getChangeByHash(state, h) !== undefined
    ok = ok && Backend.getMissingDeps(state, patch.deps.concat(['00'.repeat(32)])).length === 1
    if (!f.includes('shuffled') && !f.includes('pending') && !f.includes('dups_later')) ok = ok && all.every((c, i) => Buffer.from(c).equals(Buffer.from(changes[i])))
    if (!ok) { failed++; console.error(`FAIL ${f}: history queries`) } else console.log(`ok   ${f}  (history queries, ${all.length} changes)`)
  }
  if (fx.doc) {
    // Backend.save(state): served by the engine, byte-identical to the reference's document
    const saved = Buffer.from(Backend.save(state))
    n++
    if (!saved.equals(Buffer.from(fx.doc, 'base64'))) { failed++; console.error(`FAIL ${f}: save`) } else console.log(`ok   ${f}  (save, ${saved.length} bytes)`)
    // Backend.load(Backend.save(state)) + getPatch
    const want = fx.stock_equals_bigblock === false ? fx.load_patch_bigblock : fx.load_patch
    const loaded = Backend.load(new Uint8Array(Buffer.from(fx.doc, 'base64')))
    n++
    if (JSON.stringify(Backend.getPatch(loaded)) !== want) { failed++; console.error(`FAIL ${f}: document load`) } else console.log(`ok   ${f}  (document load)`)
    checkHistory(f, loaded)
  }
}
console.log(`${n - failed}/${n} golden fixtures reproduced through the JS Backend surface; engine: ${JSON.stringify(Backend._engineStats())}`)
process.exit(failed ? 1 : 0)
