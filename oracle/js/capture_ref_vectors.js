// TEST INFRASTRUCTURE (build container only). Runs the reference's own test suites against the UNMODIFIED reference backend
// and records, for every BackendDoc the tests build, the binary changes it was fed and the patch the reference then reports
// (Backend.getPatch), plus the batches the reference rejects. The result pins the oracle and the engine on the scenarios the
// reference's authors wrote (conflicts, counters, nested objects, tables, deletions, reordering, queued changes ...):
// tests/golden/ref_suite_vectors.jsonl, consumed by tests/test_ref_suite_vectors.py.
//
//   NODE_PATH=oracle/js_shims/node_modules node oracle/js/capture_ref_vectors.js out.jsonl new_backend_test.js backend_test.js test.js ...
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const REF = process.env.AUTOMERGE_REF || '/root/reference'
const out = process.argv[2]
const newMod = require(path.join(REF, 'backend', 'new'))
const Orig = newMod.BackendDoc
const seen = new Set()
const lines = []
const b64 = u8 => Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength).toString('base64')
function record(kind, inputs, payload) {
  const h = crypto.createHash('sha256')
  h.update(kind)
  for (const c of inputs) { h.update(String(c.byteLength)); h.update(c) }
  const key = h.digest('hex')
  if (seen.has(key)) return
  seen.add(key)
  lines.push(JSON.stringify(Object.assign({ kind, changes: inputs.map(b64) }, payload)))
}
class Rec extends Orig {
  constructor(buffer) {
    super(buffer)
    if (buffer) {
      this.__inputs = null   // history unknown: only the document vector is recorded
      try { record('doc', [buffer], { patch: JSON.stringify(this.getPatch()) }) } catch (e) { /* not our business */ }
    }
  }
  applyChanges(changes, isLocal) {
    if (this.__inputs === undefined) this.__inputs = (this.changes.length === 0 && this.queue.length === 0) ? [] : null
    const binary = Array.isArray(changes) && changes.every(c => c instanceof Uint8Array)
    let r
    try {
      r = super.applyChanges(changes, isLocal)
    } catch (e) {
      if (this.__inputs && binary) record('reject', this.__inputs.concat(changes), { error: String(e.message).split('\n')[0] })
      throw e
    }
    if (this.__inputs && binary) {
      for (const c of changes) this.__inputs.push(c)
      let patch = null
      try { patch = JSON.stringify(this.getPatch()) } catch (e) { patch = null }
      const extra = { patch }
      if (patch && this.queue.length === 0) {
        // digest of the reference's Backend.save(Backend.loadChanges(Backend.init(), changes)): a fresh document fed the same
        // changes in one batch (the document under test is left untouched)
        try {
          const fresh = new Orig()
          fresh.applyChanges(this.__inputs.slice())
          if (fresh.queue.length !== 0) throw new Error('queued')
          const doc = fresh.save()
          extra.doc_len = doc.byteLength
          extra.doc_sha256 = crypto.createHash('sha256').update(doc).digest('hex')
        } catch (e) { /* leave the digest out */ }
      }
      if (patch) record('changes', this.__inputs, extra)
    } else this.__inputs = null
    return r
  }
  clone() { const c = super.clone(); c.__inputs = this.__inputs ? this.__inputs.slice() : null; return c }
}
newMod.BackendDoc = Rec

// minimal mocha
let stack = [{ name: '', before: [], beforeEach: [], afterEach: [], tests: [], children: [] }]
global.describe = (name, fn) => { const s = { name, before: [], beforeEach: [], afterEach: [], tests: [], children: [] }; stack[stack.length - 1].children.push(s); stack.push(s); fn(); stack.pop() }
global.it = (name, fn) => stack[stack.length - 1].tests.push({ name, fn })
global.it.skip = () => {}
global.describe.skip = () => {}
global.before = fn => stack[stack.length - 1].before.push(fn)
global.beforeEach = fn => stack[stack.length - 1].beforeEach.push(fn)
global.afterEach = fn => stack[stack.length - 1].afterEach.push(fn)
global.after = () => {}
let passed = 0, failed = 0
function run(suite, bes, aes) {
  const ctx = {}
  for (const b of suite.before) b.call(ctx)
  const be = bes.concat(suite.beforeEach), ae = suite.afterEach.concat(aes)
  for (const t of suite.tests) {
    try { for (const b of be) b.call(ctx); t.fn.call(ctx); for (const a of ae) a.call(ctx); passed++ } catch (e) { failed++ }
  }
  for (const c of suite.children) run(c, be, ae)
}
for (const f of process.argv.slice(3)) require(path.join(REF, 'test', f))
run(stack[0], [], [])
fs.writeFileSync(out, lines.join('\n') + '\n')
console.error(`${passed} reference tests passed, ${failed} failed; ${lines.length} vectors -> ${out}`)
