// TEST INFRASTRUCTURE (build container: needs the reference tree). Sessions onto LOADED documents (tests/golden/apply_campaign_loaded.json.gz)
// through the engine-enabled wrapper and through the reference side by side: after every fourth call Backend.save and
// Backend.getAllChanges of both states must be byte-identical (the state behind the patches: actor table, change order, queued changes
// left out of the document, the rebuilt history in front of the applied changes).
//   LD_PRELOAD=tests/emu/libam355_emu.so NODE_PATH=oracle/js_shims/node_modules AUTOMERGE_BACKEND_PATH=/root/reference/backend node oracle/js/loaded_state_check.js
const path = require('path'), fs = require('fs'), zlib = require('zlib')
const Eng = require(path.join(__dirname, '..', '..', 'automerge_classic_amd', 'js', 'index.js'))
const Ref = require(process.env.AUTOMERGE_BACKEND_PATH || '/root/reference/backend')
const d = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(__dirname, '..', '..', 'tests', 'golden', 'apply_campaign_loaded.json.gz'))))
const pool = d.pool.map(b => new Uint8Array(Buffer.from(b, 'base64')))
let same = 0, diff = 0, histSame = 0, histDiff = 0, n = 0
const EVERY_SESSION = parseInt(process.env.EVERY_SESSION || '1')   // (the CPU suite takes every second session)
let si = 0
for (const s of d.sessions) {
  if (s.graph) continue
  if (si++ % EVERY_SESSION) continue
  const doc = new Uint8Array(Buffer.from(s.doc, 'base64'))
  let e = Eng.load(doc), r = Ref.load(doc)
  let ok = true
  for (let i = 0; i < s.calls.length; i++) {
    if (typeof s.patches[i] !== 'string') { ok = false; break }
    const batch = s.calls[i].map(k => pool[k])
    e = Eng.applyChanges(e, batch)[0]; r = Ref.applyChanges(r, batch)[0]
    if (i % 4 === 3 || i === s.calls.length - 1) {
      n++
      const se = Buffer.from(Eng.save(e)), sr = Buffer.from(Ref.save(r))
      if (se.equals(sr)) same++; else { diff++; console.log('SAVE DIFFERS', s.name, i, se.length, sr.length) }
      const he = Eng.getAllChanges(e), hr = Ref.getAllChanges(r)
      if (he.length === hr.length && he.every((c, k) => Buffer.from(c).equals(Buffer.from(hr[k])))) histSame++; else { histDiff++; console.log('HISTORY DIFFERS', s.name, i, he.length, hr.length) }
    }
  }
}
console.log(`checked ${n}: save same ${same} diff ${diff}; getAllChanges same ${histSame} diff ${histDiff}`, JSON.stringify(Eng._counters))
process.exit(diff || histDiff ? 1 : 0)
