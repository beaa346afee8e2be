// TEST INFRASTRUCTURE (build container only). Runs the reference's own test suites against the UNMODIFIED reference backend and
// records every Backend.applyChanges CALL with the patch that call returned (the incremental patch, new.js:1797-1879 with
// updatePatchProperty in its `newBlock` mode :884-1040 and setupPatches :1461-1528) together with the calls that built the
// document before it -- SURVEY.md 8f-2. A vector is a session: [{changes, local}...] applied one call after the other to an
// empty document (or to `doc`, a saved document), and the patch (or error message) of the LAST call.
//
//   NODE_PATH=oracle/js_shims/node_modules node oracle/js/capture_apply_vectors.js out.jsonl new_backend_test.js backend_test.js ...
// REF_BLOCK_SIZE=<n> runs the block-size-patched reference (oracle/js/ref_loader.js) instead of the stock one.
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const Module = require('module')
const REF = process.env.AUTOMERGE_REF || '/root/reference'
const out = process.argv[2]
if (process.env.REF_BLOCK_SIZE) {
  // the one constant MAX_BLOCK_SIZE replaced in memory (as oracle/js/ref_loader.js does), BEFORE backend.js binds BackendDoc
  const file = path.join(REF, 'backend', 'new.js')
  let src = fs.readFileSync(file, 'utf8')
  const needle = 'const MAX_BLOCK_SIZE = 600'
  if (!src.includes(needle)) throw new Error('MAX_BLOCK_SIZE constant not found in reference new.js')
  src = src.replace(needle, `const MAX_BLOCK_SIZE = ${parseInt(process.env.REF_BLOCK_SIZE)}`)
  const m = new Module(file, module)
  m.filename = file
  m.paths = Module._nodeModulePaths(path.dirname(file))
  m._compile(src, file)
  require.cache[file] = m
  m.loaded = true
}
const newMod = require(path.join(REF, 'backend', 'new'))
const Orig = newMod.BackendDoc
const seen = new Set()
const lines = []
const b64 = u8 => Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength).toString('base64')
function record(session, doc, result) {
  const h = crypto.createHash('sha256')
  if (doc) { h.update('doc'); h.update(doc) }
  for (const call of session) {
    h.update(call.local ? 'L' : 'R')
    for (const c of call.changes) { h.update(String(c.byteLength)); h.update(c) }
  }
  const key = h.digest('hex')
  if (seen.has(key)) return
  seen.add(key)
  const v = { calls: session.map(call => ({ local: call.local, changes: call.changes.map(b64) })) }
  if (doc) v.doc = b64(doc)
  lines.push(JSON.stringify(Object.assign(v, result)))
}
class Rec extends Orig {
  constructor(buffer) {
    super(buffer)
    this.__doc = buffer || null
    this.__session = []
  }
  applyChanges(changes, isLocal) {
    const binary = Array.isArray(changes) && changes.every(c => c instanceof Uint8Array)
    if (!binary || !this.__session) { this.__session = null; return super.applyChanges(changes, isLocal) }
    const session = this.__session.concat([{ local: !!isLocal, changes: changes.slice() }])
    let r
    try {
      r = super.applyChanges(changes, isLocal)
    } catch (e) {
      record(session, this.__doc, { error: String(e.message).split('\n')[0] })
      throw e
    }
    this.__session = session
    record(session, this.__doc, { patch: JSON.stringify(r) })
    return r
  }
  clone() {
    const c = super.clone()
    c.__doc = this.__doc
    c.__session = this.__session ? this.__session.slice() : null
    return c
  }
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
console.error(`${passed} reference tests passed, ${failed} failed; ${lines.length} applyChanges vectors -> ${out}`)
