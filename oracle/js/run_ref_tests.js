// TEST INFRASTRUCTURE. Runs the reference's own mocha suites (from /root/reference/test) with mi355x-backend
// installed as the default backend -- the mechanism of the reference's test/wasm.js:12-25 -- using a minimal
// describe/it runner (mocha is not installed in the build container).
//   NODE_PATH=oracle/js_shims/node_modules MI355X_BACKEND_JS_ONLY=1 AUTOMERGE_BACKEND_PATH=/root/reference/backend \
//     node oracle/js/run_ref_tests.js backend_test.js test.js text_test.js ...
const path = require('path')
const REF = process.env.AUTOMERGE_REF || '/root/reference'
const Automerge = require(path.join(REF, 'src', 'automerge'))
const Backend = require(path.join(__dirname, '..', '..', 'automerge_classic_amd', 'js', 'index.js'))
Automerge.setDefaultBackend(Backend)

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
function run(suite, prefix, bes, aes) {
  const ctx = {}
  for (const b of suite.before) b.call(ctx)
  const be = bes.concat(suite.beforeEach), ae = suite.afterEach.concat(aes)
  for (const t of suite.tests) {
    try {
      for (const b of be) b.call(ctx)
      const r = t.fn.call(ctx)
      if (r && typeof r.then === 'function') throw new Error('async tests unsupported by this runner')
      for (const a of ae) a.call(ctx)
      passed++
    } catch (e) { failed++; console.error(`FAIL ${prefix}${suite.name} > ${t.name}: ${e.message.split('\n')[0]}`) }
  }
  for (const c of suite.children) run(c, prefix + suite.name + ' > ', be, ae)
}
for (const f of process.argv.slice(2)) require(path.join(REF, 'test', f))
run(stack[0], '', [], [])
console.log(`${passed} passed, ${failed} failed (reference suites ${process.argv.slice(2).join(', ')} against mi355x-backend${process.env.MI355X_BACKEND_JS_ONLY === '1' ? ' in JS-only plumbing mode' : ''})`)
if (Backend._counters) console.log('served by: ' + JSON.stringify(Backend._counters))
process.exit(failed ? 1 : 0)
