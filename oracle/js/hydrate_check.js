// TEST INFRASTRUCTURE (build container: needs the reference tree). ADVICE r3: a state that SEVERAL engine-served applyChanges calls
// built must hydrate to the reference's own state -- objectMeta.children depends on where the calls ended -- because every call the
// engine refuses, applyLocalChange and clone run on the hydrated handle. For the sessions of tests/golden/apply_campaign*.json.gz:
// calls 0..i through mi355x-backend (engine), then call i+1 through the REFERENCE on the clone of that state; its patch must be the
// one the reference recorded for call i+1.
//   LD_PRELOAD=tests/emu/libam355_emu.so NODE_PATH=oracle/js_shims/node_modules AUTOMERGE_BACKEND_PATH=/root/reference/backend node oracle/js/hydrate_check.js [fixture...]
const path = require('path'), fs = require('fs'), zlib = require('zlib')
const REF = process.env.AUTOMERGE_REF || '/root/reference'
const Backend = require(path.join(__dirname, '..', '..', 'automerge_classic_amd', 'js', 'index.js'))
const RefBackend = require(path.join(REF, 'backend'))
const golden = path.join(__dirname, '..', '..', 'tests', 'golden')
const fixtures = process.argv.length > 2 ? process.argv.slice(2) : ['apply_campaign.json.gz']
const EVERY = parseInt(process.env.HYDRATE_EVERY || "2")
let checked = 0, different = 0, engineCalls = 0
for (const name of fixtures) {
  const d = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(golden, name))))
  const pool = d.pool.map(b => new Uint8Array(Buffer.from(b, 'base64')))
  for (const s of d.sessions) {
    let backend = Backend.init()
    for (let i = 0; i + 1 < s.calls.length; i++) {
      if (typeof s.patches[i] !== 'string' || typeof s.patches[i + 1] !== 'string') break
      const before = Backend._counters.gpuApplyChanges
      backend = Backend.applyChanges(backend, s.calls[i].map(k => pool[k]))[0]
      engineCalls += Backend._counters.gpuApplyChanges - before
      if (EVERY > 1 && i % EVERY !== EVERY - 1 && i + 2 < s.calls.length) continue      // (every second call and the last: a clone replays the whole lineage)
      const hydrated = Backend.clone(backend)                    // reference handle, made by hydrate()
      backend.state.js = null                                    // (test only: drop the cached handle so that the session stays on the engine)
      const patch = RefBackend.applyChanges(hydrated, s.calls[i + 1].map(k => pool[k]))[1]
      checked++
      if (JSON.stringify(patch) !== s.patches[i + 1]) { different++; console.log(`DIFFERENT ${name} ${s.name} call ${i + 1}`) }
    }
  }
}
console.log(`hydrate check: ${checked} reference calls on hydrated states, ${engineCalls} engine calls before them, DIFFERENT ${different}`)
process.exit(different ? 1 : 0)
