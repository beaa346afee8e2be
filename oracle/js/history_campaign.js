// TEST INFRASTRUCTURE (build container: needs the reference tree). Differential campaign for the history reconstruction
// (am355_doc_changes vs the reference's Backend.getAllChanges(Backend.load(doc))) on damaged documents: small documents made by the
// reference from prefixes of the golden fixtures' changes (columns stay below the DEFLATE threshold), one byte mutated, container
// checksum repaired. The engine must never return a history the reference does not return byte for byte; it may refuse.
//   LD_PRELOAD=tests/emu/libam355_emu.so NODE_PATH=oracle/js_shims/node_modules node oracle/js/history_campaign.js [mutations per doc] [seed]
const fs = require('fs'), path = require('path'), crypto = require('crypto')
const { loadBackend } = require('./ref_loader')
const { Backend } = loadBackend()
const addon = process.env.HISTORY_CAMPAIGN_CHILD ? null : require(path.join(__dirname, '..', '..', 'automerge_classic_amd', 'js', 'am355_napi.node'))
const ctx = addon && addon.create(0)
const perDoc = parseInt(process.argv[2] || '300'), seed0 = parseInt(process.argv[3] || '1')
let s = BigInt(seed0)
function rnd(n) { s = (s * 6364136223846793005n + 1442695040888963407n) & ((1n << 64n) - 1n); return Number((s >> 33n) % BigInt(n)) }

// (in a child process: the reference can exhaust the heap on a damaged run length, which no try/catch survives)
const { spawnSync } = require('child_process')
function refHistory(doc) {
  if (process.env.HISTORY_CAMPAIGN_CHILD) {
    try { return Backend.getAllChanges(Backend.load(doc)).map(c => Buffer.from(c).toString('hex')) } catch (e) { return { error: e.message } }
  }
  const r = spawnSync(process.execPath, ['--max-old-space-size=512', __filename], { input: Buffer.from(doc), env: Object.assign({}, process.env, { HISTORY_CAMPAIGN_CHILD: '1', LD_PRELOAD: '' }), maxBuffer: 1 << 28, timeout: 120000 })
  if (r.status !== 0) return { error: 'reference process died' }
  return JSON.parse(r.stdout.toString())
}
if (process.env.HISTORY_CAMPAIGN_CHILD) {
  const chunks = []
  process.stdin.on('data', c => chunks.push(c))
  process.stdin.on('end', () => { process.stdout.write(JSON.stringify(refHistory(new Uint8Array(Buffer.concat(chunks))))); process.exit(0) })
  return
}
function engineHistory(doc) {
  try {
    addon.loadDocument(ctx, doc); addon.replay(ctx)
  } catch (e) { return { error: 'load: ' + e.message } }
  try { return addon.docChanges(ctx, 1).changes.map(c => Buffer.from(c).toString('hex')) } catch (e) { return { error: e.message, code: e.am355Code } }
}

const dir = path.join(__dirname, '..', '..', 'tests', 'golden')
let docs = []
for (const f of ['frontend_mixed_3actors.json', 'frontend_text_4actors.json', 'campaign_mixed_1008.json', 'frontend_mixed_6actors.json', 'hand_key_order.json']) {
  const fx = JSON.parse(fs.readFileSync(path.join(dir, f), 'utf8'))
  const changes = fx.changes.map(c => new Uint8Array(Buffer.from(c, 'base64')))
  for (const k of [3, 6, 10]) {
    if (k > changes.length) continue
    try { docs.push({ name: `${f}[0..${k})`, doc: Backend.save(Backend.loadChanges(Backend.init(), changes.slice(0, k))) }) } catch (e) { /* prefix with missing deps */ }
  }
}
let same = 0, refused = 0, bothFail = 0, bad = 0, total = 0
for (const { name, doc } of docs) {
  const base = engineHistory(doc), want = refHistory(doc)
  if (JSON.stringify(base) !== JSON.stringify(want)) { console.error(`DISAGREE on the undamaged ${name}`); bad++ }
  for (let m = 0; m < perDoc; m++) {
    const d = Uint8Array.from(doc)
    const pos = 8 + rnd(d.length - 8)
    d[pos] = rnd(256)
    crypto.createHash('sha256').update(d.subarray(8)).digest().copy(d, 4, 0, 4)
    const got = engineHistory(d), ref = refHistory(d)
    total++
    if (Array.isArray(got)) {
      if (Array.isArray(ref) && JSON.stringify(got) === JSON.stringify(ref)) same++
      else { bad++; console.error(`DISAGREE ${name} byte ${pos}: engine returned ${got.length} changes, reference ${Array.isArray(ref) ? ref.length + ' different changes' : 'threw ' + ref.error}`) }
    } else if (Array.isArray(ref)) refused++
    else bothFail++
  }
}
console.log(`${docs.length} documents, ${total} mutations: identical history ${same}, engine refused what the reference serves ${refused}, both refuse ${bothFail}, DISAGREE ${bad}`)
process.exit(bad ? 1 : 0)
