// TEST INFRASTRUCTURE (build container: needs the reference tree). Differential campaign over STRUCTURE that oracle/js/make_golden.js
// does not generate: tables (rows added / updated / removed), bulk list operations (splice of several values = one multi-insert op,
// deleteAt(i, n) = one multi-delete op, columnar.js:446-475 expandMultiOps), lists of lists, text in lists, concurrent edits of the
// same rows / elements by several actors merged in random order. Engine (addon called directly, no JS fallback) against the live
// reference: patch text, materialised patch, save, patch after load, history after load. The engine may refuse; it must not differ.
//   LD_PRELOAD=tests/emu/libam355_emu.so NODE_PATH=oracle/js_shims/node_modules node oracle/js/structure_campaign.js [scenarios] [seed]
const path = require('path')
const { loadBackend } = require('./ref_loader')
const { Backend, Automerge: getA } = loadBackend()
const Automerge = getA()
const addon = require(path.join(__dirname, '..', '..', 'automerge_classic_amd', 'js', 'am355_napi.node'))
const { materialize } = require(path.join(__dirname, '..', '..', 'automerge_classic_amd', 'js', 'materialize.js'))
const ctx = addon.create(0)
const nScenarios = parseInt(process.argv[2] || '40'), seed0 = parseInt(process.argv[3] || '1')

function rng(seed) {
  let s = BigInt(seed) * 0x9e3779b97f4a7c15n & ((1n << 64n) - 1n)
  return () => { s = (s * 6364136223846793005n + 1442695040888963407n) & ((1n << 64n) - 1n); return Number(s >> 11n) / 9007199254740992 }
}
const WORDS = ['alpha', 'beta', 'gamma', 'delta', 'k', 'row', 'x']

function scenario(seed) {
  const rnd = rng(seed), pick = a => a[Math.floor(rnd() * a.length)], int = n => Math.floor(rnd() * n)
  const nActors = 2 + int(4), steps = 10 + int(40)
  const ids = []
  for (let i = 0; i < nActors; i++) { let s = 'abcdef'[int(6)]; while (s.length < 32) s += '0123456789abcdef'[int(16)]; ids.push(s) }
  let docs = ids.map(id => Automerge.init(id))
  docs[0] = Automerge.change(docs[0], d => {
    d.table = new Automerge.Table(); d.list = [1, 2, 3, 4, 5, 6]; d.grid = [[1, 2], [3]]; d.text = new Automerge.Text('abcdefghij'); d.texts = [new Automerge.Text('in a list')]
  })
  for (let i = 1; i < nActors; i++) docs[i] = Automerge.merge(docs[i], docs[0])
  const val = () => { const r = rnd(); return r < 0.3 ? int(100) : r < 0.5 ? pick(WORDS) + int(9) : r < 0.6 ? rnd() : r < 0.7 ? pick([true, false, null]) : r < 0.8 ? new Date(1.6e12 + int(1e6)) : pick(WORDS) }
  for (let s = 0; s < steps; s++) {
    const a = int(nActors)
    try {
      docs[a] = Automerge.change(docs[a], d => {
        const r = rnd()
        if (r < 0.25) {            // tables
          const rowIds = d.table.ids
          const q = rnd()
          if (rowIds.length === 0 || q < 0.5) d.table.add({ name: pick(WORDS), n: int(50), tags: [pick(WORDS)] })
          else if (q < 0.75) { const row = d.table.byId(pick(rowIds)); row.n = int(50); if (rnd() < 0.3) row.tags.push(pick(WORDS)); if (rnd() < 0.2) delete row.name }
          else d.table.remove(pick(rowIds))
        } else if (r < 0.55) {     // bulk list operations
          const l = rnd() < 0.7 ? d.list : pick(d.grid.concat([d.list]))
          const q = rnd()
          if (q < 0.45) l.splice(int(l.length + 1), 0, ...Array.from({ length: 1 + int(5) }, val))
          else if (q < 0.75 && l.length > 1) l.splice(int(l.length - 1), 1 + int(Math.min(3, l.length - 1)))
          else if (l.length > 0) l.splice(int(l.length), 1 + int(2), val(), val())
          else l.push(val())
        } else if (r < 0.8) {      // text: bursts and multi-character deletes
          const t = rnd() < 0.8 ? d.text : d.texts[0]
          const q = rnd()
          if (q < 0.5) t.insertAt(int(t.length + 1), ...(pick(WORDS) + pick(WORDS)).split(''))
          else if (t.length > 2) t.deleteAt(int(t.length - 2), 1 + int(Math.min(4, t.length - 2)))
          else t.insertAt(0, 'q')
        } else if (r < 0.9) {      // lists of lists / texts in lists
          const q = rnd()
          if (q < 0.4) d.grid.push([val(), val()])
          else if (q < 0.6 && d.grid.length > 1) d.grid.splice(int(d.grid.length), 1)
          else if (q < 0.8) d.texts.push(new Automerge.Text(pick(WORDS)))
          else if (d.texts.length > 1) d.texts.splice(1 + int(d.texts.length - 1), 1)
        } else {                   // replace a container by a scalar or the other way round
          const k = pick(['extra', 'other'])
          d[k] = rnd() < 0.5 ? val() : (rnd() < 0.5 ? [val(), [val()]] : { t: new Automerge.Table(), m: { deep: val() } })
        }
      })
    } catch (e) { /* the frontend refused this edit */ }
    if (rnd() < 0.4) { const b = int(nActors); if (b !== a) docs[b] = Automerge.merge(docs[b], docs[a]) }
  }
  let all = Automerge.init()
  const order = ids.map((_, i) => i).sort(() => rnd() - 0.5)
  for (const i of order) all = Automerge.merge(all, docs[i])
  return Automerge.getAllChanges(all)
}

let same = 0, refused = 0, bad = 0, stockDiffers = 0
const soft = e => e.am355Code === -4 || e.am355Code === -3
function check(what, name, got, want) {
  if (got === want) { same++; return true }
  bad++
  let i = 0
  while (i < got.length && got[i] === want[i]) i++
  console.error(`DISAGREE ${what} of ${name} at char ${i}: engine ...${JSON.stringify(got.slice(Math.max(0, i - 40), i + 60))} reference ...${JSON.stringify(want.slice(Math.max(0, i - 40), i + 60))}`)
  return false
}
for (let k = 0; k < nScenarios; k++) {
  const name = `scenario ${seed0 * 1000 + k}`
  const changes = scenario(seed0 * 1000 + k)
  let refPatch, refDoc, refLoad, refHistory
  try {
    const st = Backend.loadChanges(Backend.init(), changes)
    refPatch = JSON.stringify(Backend.getPatch(st))
    const doc = Backend.save(st)
    refDoc = Buffer.from(doc).toString('hex')
    const loaded = Backend.load(doc)
    refLoad = JSON.stringify(Backend.getPatch(loaded))
    try { refHistory = Backend.getAllChanges(loaded).map(c => Buffer.from(c).toString('hex')).join(' ') } catch (e) { refHistory = null }
  } catch (e) { console.log(`(reference backend throws on ${name}: ${e.message})`); continue }
  try {
    addon.loadChanges(ctx, changes); addon.replay(ctx)
    check('patch text', name, addon.patchJSON(ctx), refPatch)
    check('materialised patch', name, JSON.stringify(materialize(addon.fetchIR(ctx))), refPatch)
    check('save', name, Buffer.from(addon.save(ctx, 0)).toString('hex'), refDoc)
    addon.loadDocument(ctx, new Uint8Array(Buffer.from(refDoc, 'hex'))); addon.replay(ctx)
    check('patch after load', name, JSON.stringify(materialize(addon.fetchIR(ctx))), refLoad)
    try {
      const h = addon.docChanges(ctx, 1).changes.map(c => Buffer.from(c).toString('hex')).join(' ')
      if (refHistory === null) { bad++; console.error(`DISAGREE history of ${name}: the reference throws`) } else check('history', name, h, refHistory)
    } catch (e) { if (!soft(e)) throw e; refused++ }
  } catch (e) {
    if (soft(e)) { refused++; console.log(`(engine refuses ${name}: ${e.message.slice(0, 120)})`) } else throw e
  }
  // the same changes in a random delivery order, with duplicates; and with some changes missing (their dependents stay queued)
  const rnd = rng(seed0 * 7919 + k)
  const shuffled = changes.slice().sort(() => rnd() - 0.5)
  shuffled.push(changes[Math.floor(rnd() * changes.length)])
  const holes = changes.filter(() => rnd() > 0.15).sort(() => rnd() - 0.5)
  for (const [what, batch] of [['patch (shuffled delivery + a duplicate)', shuffled], ['patch (changes missing)', holes]]) {
    let want
    try { want = JSON.stringify(Backend.getPatch(Backend.loadChanges(Backend.init(), batch))) } catch (e) { console.log(`(reference throws on ${what} of ${name}: ${e.message})`); continue }
    try {
      addon.loadChanges(ctx, batch); addon.replay(ctx)
      check(what, name, JSON.stringify(materialize(addon.fetchIR(ctx))), want)
    } catch (e) { if (!soft(e)) throw e; refused++; console.log(`(engine refuses ${what} of ${name}: ${e.message.slice(0, 100)})`) }
  }
}
console.log(`${nScenarios} scenarios: ${same} results identical, ${refused} refusals, DISAGREE ${bad}`)
process.exit(bad ? 1 : 0)
