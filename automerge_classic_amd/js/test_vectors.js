// Every vector captured from the reference's own test suites (tests/golden/ref_suite_vectors.json.gz: 1477 sequences of binary
// changes with the patch the unmodified reference reports, 18 saved documents, 4 rejected batches) through the JS host:
// node -> index.js -> am355_napi.node -> engine -> patch IR -> materialize.js. No reference tree needed (runs on the GPU box).
// A vector passes when JSON.stringify of the materialised patch's `diffs` (key order of every object included), `deps`, `maxOp`
// and `pendingChanges` equal the reference's and `clock` is deep-equal (its key order records application order, which differs
// between one batch and the several calls the reference's tests made).
//   node automerge_classic_amd/js/test_vectors.js [tests/golden/ref_suite_vectors.json.gz] [stride] [offset]
'use strict'
const fs = require('fs')
const path = require('path')
const zlib = require('zlib')
const addon = require(path.join(__dirname, 'am355_napi.node'))
const { materialize } = require('./materialize.js')

const file = process.argv[2] || path.join(__dirname, '..', '..', 'tests', 'golden', 'ref_suite_vectors.json.gz')
const stride = parseInt(process.argv[3] || '1'), offset = parseInt(process.argv[4] || '0')
const d = JSON.parse(zlib.gunzipSync(fs.readFileSync(file)).toString('utf8'))
const pool = d.pool.map(x => new Uint8Array(Buffer.from(x, 'base64')))
const ctx = addon.create(parseInt(process.env.MI355X_DEVICE || '0'))

function samePatch(got, want) {
  if (JSON.stringify(Object.keys(got)) !== JSON.stringify(Object.keys(want))) return false
  for (const k of Object.keys(got)) {
    if (k === 'clock') {
      const a = Object.keys(got.clock).sort(), b = Object.keys(want.clock).sort()
      if (JSON.stringify(a) !== JSON.stringify(b) || a.some(x => got.clock[x] !== want.clock[x])) return false
    } else if (JSON.stringify(got[k]) !== JSON.stringify(want[k])) return false
  }
  return true
}

let equal = 0, rejected = [], unsupported = [], failed = 0, n = 0
d.vectors.forEach((v, i) => {
  if (i % stride !== offset) return
  n++
  const blobs = v.changes.map(k => pool[k])
  let patch
  try {
    if (v.kind === 'doc') addon.loadDocument(ctx, blobs[0]); else addon.loadChanges(ctx, blobs)
    addon.replay(ctx)
    patch = materialize(addon.fetchIR(ctx))
  } catch (e) {
    if (e.am355Code === -3) { rejected.push(i); if (v.kind !== 'reject') { failed++; console.error(`FAIL ${i}: rejected a batch the reference accepts: ${e.message}`) } return }
    if (e.am355Code === -4) { unsupported.push(i); return }
    throw e
  }
  if (v.kind === 'reject') { failed++; console.error(`FAIL ${i}: accepted a batch the reference rejects`); return }
  if (!samePatch(patch, JSON.parse(v.patch))) { failed++; console.error(`FAIL ${i}: patch differs`); return }
  // the engine's own JSON text must say the same (two renderings of one IR)
  if (JSON.stringify(patch) !== addon.patchJSON(ctx)) { failed++; console.error(`FAIL ${i}: materialised patch != engine JSON text`); return }
  equal++
})
addon.destroy(ctx)
console.log(JSON.stringify({ vectors: n, equal, rejected, unsupported, failed }))
process.exit(failed ? 1 : 0)
