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
// (`clock` may list its keys in another order: the engine keeps no insertion order of a JS object)
function samePatch(a, b) {
  const strip = p => JSON.stringify(Object.assign({}, p, { clock: null }))
  const ck = p => JSON.stringify(Object.keys(p.clock).sort().map(k => [k, p.clock[k]]))
  return strip(a) === strip(b) && ck(a) === ck(b)
}
for (const name of fixtures) {
  const d = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(golden, name))))
  const pool = d.pool.map(b => new Uint8Array(Buffer.from(b, 'base64')))
  const MAX_SESSIONS = parseInt(process.env.MAX_SESSIONS || '1000000')   // (per fixture; the CPU suite takes a slice)
  for (const s of d.sessions.slice(0, MAX_SESSIONS)) {
    // (a session of apply_campaign_loaded.json.gz goes onto a LOADED document: Backend.load on the engine, the calls onto it)
    let backend = s.doc ? Backend.load(new Uint8Array(Buffer.from(s.doc, 'base64'))) : Backend.init()
    if (s.graph) Backend.getAllChanges(backend)   // (the recorded session asked the reference for the document's changes first)
    for (let i = 0; i + 1 < s.calls.length; i++) {
      if (typeof s.patches[i] !== 'string' || typeof s.patches[i + 1] !== 'string') break
      const before = Backend._counters.gpuApplyChanges
      const [b2, own] = Backend.applyChanges(backend, s.calls[i].map(k => pool[k]))
      backend = b2
      engineCalls += Backend._counters.gpuApplyChanges - before
      // STEAL=1 (with MI355X_CONTEXTS=1): another document takes the only engine context after every call, so that every call finds its
      // state's context gone and the wrapper replays the retained changes first (index.js replayRetained / heldBack)
      if (process.env.STEAL) Backend.getPatch(Backend.loadChanges(Backend.init(), [pool[s.calls[0][0]]]))
      if (!samePatch(own, JSON.parse(s.patches[i]))) { different++; console.log(`DIFFERENT (the wrapper's own patch) ${name} ${s.name} call ${i}`) }
      if (EVERY > 1 && i % EVERY !== EVERY - 1 && i + 2 < s.calls.length) continue      // (every second call and the last: a clone replays the whole lineage)
      // reference handle, made by hydrate(). (Not a clone for a loaded lineage: BackendDoc.clone rebuilds the hash graph, new.js:1774,
      // and the recorded session went on without one)
      const hydrated = s.doc ? Backend._hydrate(backend) : Backend.clone(backend)
      backend.state.js = null                                    // (test only: drop the cached handle so that the session stays on the engine)
      const patch = RefBackend.applyChanges(hydrated, s.calls[i + 1].map(k => pool[k]))[1]
      checked++
      if (JSON.stringify(patch) !== s.patches[i + 1]) { different++; console.log(`DIFFERENT ${name} ${s.name} call ${i + 1}`) }
    }
  }
}
console.log(`hydrate check: ${checked} reference calls on hydrated states, ${engineCalls} engine calls before them, DIFFERENT ${different}`)
process.exit(different ? 1 : 0)
