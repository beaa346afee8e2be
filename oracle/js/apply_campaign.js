// TEST INFRASTRUCTURE (build container only). Differential campaign for the INCREMENTAL patches of Backend.applyChanges
// (SURVEY.md 8f-2): random multi-actor documents made with the real frontend (the generators of make_golden.js), delivered to a
// fresh reference backend in random batches -- in causal order, or shuffled so that changes wait in the queue -- and every
// call's patch recorded.  One JSON line per session: {name, calls: [[base64 change...]...], patches: [JSON text | {error}]}.
//
//   NODE_PATH=oracle/js_shims/node_modules [REF_BLOCK_SIZE=100000000] node oracle/js/apply_campaign.js out.jsonl SPEC...
//   SPEC = seed:actors:steps:depth (mixed document) | t:seed:actors:rounds:burst (text) | m:seed:actors:steps:depth (see servedScenario)
//        | l:seed:actors:steps:p2 (see listScenario) | c:seed:actors:elements:keys (see conflictScenario: its own call plan);
//   every other SPEC yields 3 sessions.  With LOADED=1 in the environment every session is cut in two: the first calls build a
//   document that is saved and loaded again (Backend.load: objectMeta from the document's rows, hash graph rebuilt on the first
//   applyChanges), the recorded calls are those made onto the loaded document ({..., doc: base64 of the saved document}); every
//   session a second time with the hash graph rebuilt by a query before the first call (name "...+g", graph: true).
const fs = require('fs')
const { splitmix, frontendScenario, textScenario, Backend } = require('./make_golden.js')
const out = process.argv[2]
const Automerge = require('./make_golden.js').Automerge
const refColumnar = require(require('path').join(require('./ref_loader').REF, 'backend', 'columnar'))

// "m:" scenarios: what the engine's incremental-patch stage serves (automerge_classic_amd/csrc/am355_delta.hip) -- nested maps and
// tables with conflicting assignments, deletions and counters, lists and texts that grow and shrink by insertion and deletion
// (objects as list elements included) -- edited by several actors that merge at random. (THIS generator never assigns to list
// elements; the "l:" and "c:" generators below do.)
function servedScenario(seed, nActors, steps, depthLimit) {
  const rnd = splitmix(seed)
  const pick = arr => arr[Math.floor(rnd() * arr.length)]
  const KEYS = ['alpha', 'beta', 'gamma', 'x', 'y', 'k1', 'k2', '10', '9', '007', 'Ünï', '日本', '😀smile', '￮z', 'quote"q']
  const ids = []
  for (let i = 0; i < nActors; i++) { let a = 'abcdef'[Math.floor(rnd() * 6)]; while (a.length < 32) a += '0123456789abcdef'[Math.floor(rnd() * 16)]; ids.push(a) }
  let docs = ids.map(id => Automerge.init(id))
  docs[0] = Automerge.change(docs[0], d => { d.title = 'start'; d.items = ['a', 'b']; d.text = new Automerge.Text('hello'); d.n = new Automerge.Counter(1); d.cfg = { deep: { er: 1 } }; d.rows = new Automerge.Table() })
  for (let i = 1; i < nActors; i++) docs[i] = Automerge.merge(docs[i], docs[0])
  const scalar = () => {
    const r = rnd()
    if (r < 0.3) return Math.floor(rnd() * 2000) - 1000
    if (r < 0.4) return rnd() * 1e3
    if (r < 0.5) return pick([true, false, null])
    if (r < 0.55) return new Date(1600000000000 + Math.floor(rnd() * 1e9))
    if (r < 0.65) return new Automerge.Counter(Math.floor(rnd() * 10))
    return pick(KEYS) + Math.floor(rnd() * 10)
  }
  const plain = v => (v instanceof Automerge.Counter ? 7 : v)   // no counters inside lists
  const value = depth => {
    const r = rnd()
    if (depth < depthLimit && r < 0.15) return { [pick(KEYS)]: scalar(), n: scalar() }
    if (depth < depthLimit && r < 0.25) return [plain(scalar()), plain(scalar())]
    if (depth < depthLimit && r < 0.32) return new Automerge.Text(pick(KEYS) + 'txt')
    return scalar()
  }
  const walk = (obj, depth) => {
    const isList = Array.isArray(obj) || obj instanceof Automerge.Text
    const keys = isList ? [...Array(obj.length).keys()] : (obj instanceof Automerge.Table ? obj.ids : Object.keys(obj))
    const get = k => (obj instanceof Automerge.Text ? obj.get(k) : obj instanceof Automerge.Table ? obj.byId(k) : obj[k])
    const containers = keys.filter(k => { let v; try { v = get(k) } catch (e) { return false }; return v && typeof v === 'object' && !(v instanceof Date) && !(v instanceof Automerge.Counter) })
    if (containers.length > 0 && rnd() < 0.6 && depth < 4) return walk(get(pick(containers)), depth + 1)
    return [obj, depth]
  }
  for (let s = 0; s < steps; s++) {
    const a = Math.floor(rnd() * nActors)
    try {
      docs[a] = Automerge.change(docs[a], d => {
        const nOps = 1 + Math.floor(rnd() * 4)
        for (let i = 0; i < nOps; i++) {
          const [obj, depth] = walk(d, 0)
          if (obj instanceof Automerge.Text) {
            if (obj.length > 0 && rnd() < 0.35) obj.deleteAt(Math.floor(rnd() * obj.length), 1 + Math.floor(rnd() * Math.min(3, obj.length)))
            else obj.insertAt(Math.floor(rnd() * (obj.length + 1)), ...(pick(KEYS) + 'ab').split(''))
          } else if (Array.isArray(obj)) {
            if (obj.length > 0 && rnd() < 0.35) obj.splice(Math.floor(rnd() * obj.length), 1)
            else obj.splice(Math.floor(rnd() * (obj.length + 1)), 0, plain(value(depth + 1)))
          } else if (obj instanceof Automerge.Table) {
            if (obj.count > 0 && rnd() < 0.3) obj.remove(pick(obj.ids))
            else obj.add({ name: pick(KEYS), v: Math.floor(rnd() * 100) })
          } else {
            const keys = Object.keys(obj).filter(k => k !== 'id')
            const r = rnd()
            if (keys.length > 0 && r < 0.15) delete obj[pick(keys)]
            else if (keys.length > 0 && r < 0.45) {
              const k = pick(keys)
              if (obj[k] instanceof Automerge.Counter) { if (rnd() < 0.6) obj[k].increment(1 + Math.floor(rnd() * 5)); else obj[k].decrement(1) }
              else obj[k] = value(depth + 1)
            } else obj[pick(KEYS)] = value(depth + 1)
          }
        }
      })
    } catch (e) { /* the frontend refused this edit */ }
    if (rnd() < 0.35) { const b = Math.floor(rnd() * nActors); if (b !== a) docs[b] = Automerge.merge(docs[b], docs[a]) }
  }
  let all = Automerge.init()
  for (let i = 0; i < nActors; i++) all = Automerge.merge(all, docs[i])
  return Automerge.getAllChanges(all)
}
// "l:" scenarios: lists of plain values whose ELEMENTS ARE ASSIGNED TO (`list[i] = v`) besides growing and shrinking, by several
// actors that merge at random: concurrent assignments (several visible values per element), assignment against deletion (the element
// comes back), assignments to elements inserted by the same batch. Most changes hold one op (p2 = probability of a second op).
function listScenario(seed, nActors, steps, p2pct) {
  const rnd = splitmix(seed)
  const pick = arr => arr[Math.floor(rnd() * arr.length)]
  const ids = []
  for (let i = 0; i < nActors; i++) { let a = 'abcdef'[Math.floor(rnd() * 6)]; while (a.length < 32) a += '0123456789abcdef'[Math.floor(rnd() * 16)]; ids.push(a) }
  let docs = ids.map(id => Automerge.init(id))
  docs[0] = Automerge.change(docs[0], d => { d.list = ['a', 'b', 'c', 'd']; d.nums = [1, 2, 3]; d.text = new Automerge.Text('hey'); d.k = 0 })
  for (let i = 1; i < nActors; i++) docs[i] = Automerge.merge(docs[i], docs[0])
  const scalar = () => { const r = rnd(); return r < 0.4 ? Math.floor(rnd() * 100) : r < 0.5 ? pick([true, null]) : r < 0.6 ? rnd() * 10 : 'v' + Math.floor(rnd() * 1000) }
  const oneOp = d => {
    const r0 = rnd()
    if (r0 < 0.08) { d.k = Math.floor(rnd() * 50); return }
    if (r0 < 0.2) { const t = d.text; if (t.length > 0 && rnd() < 0.4) t.deleteAt(Math.floor(rnd() * t.length)); else t.insertAt(Math.floor(rnd() * (t.length + 1)), pick(['x', 'y', 'z'])); return }
    const l = rnd() < 0.7 ? d.list : d.nums
    const r = rnd()
    if (l.length > 0 && r < 0.45) l[Math.floor(rnd() * l.length)] = scalar()
    else if (l.length > 0 && r < 0.65) l.splice(Math.floor(rnd() * l.length), 1 + (rnd() < 0.2 && l.length > 2 ? 1 : 0))
    else l.splice(Math.floor(rnd() * (l.length + 1)), 0, ...(rnd() < 0.3 ? [scalar(), scalar()] : [scalar()]))
  }
  for (let s = 0; s < steps; s++) {
    const a = Math.floor(rnd() * nActors)
    try {
      docs[a] = Automerge.change(docs[a], d => { oneOp(d); if (rnd() * 100 < p2pct) oneOp(d) })
    } catch (e) { /* the frontend refused this edit */ }
    if (rnd() < 0.3) { const b = Math.floor(rnd() * nActors); if (b !== a) docs[b] = Automerge.merge(docs[b], docs[a]) }
  }
  let all = Automerge.init()
  for (let i = 0; i < nActors; i++) all = Automerge.merge(all, docs[i])
  return Automerge.getAllChanges(all)
}
// "c:" scenarios: WIDE conflicts -- every actor assigns (or deletes) the same `nElems` list elements and sets `nKeys` map keys in ONE
// change each, concurrently; the changes are then delivered one call per actor. Every element of the later calls yields an update
// edit with one record PER VISIBLE VALUE, so the edit records of a call outnumber its op rows (2-way: 2 per row, 3-way: 3 per row):
// the shape that overflowed the engine's edit table (ADVICE r3). Returns {changes, calls} (its own call plan).
function conflictScenario(seed, nActors, nElems, nKeys) {
  const rnd = splitmix(seed)
  const ids = []
  for (let i = 0; i < nActors; i++) { let a = 'abcdef'[i % 6]; while (a.length < 32) a += '0123456789abcdef'[Math.floor(rnd() * 16)]; ids.push(a) }
  let base = Automerge.init(ids[0])
  base = Automerge.change(base, d => { d.l = []; for (let i = 0; i < nElems; i++) d.l.push('e' + i); d.m = {}; d.t = new Automerge.Text('abc') })
  const baseChanges = Automerge.getAllChanges(base)
  const docs = ids.map((id, i) => (i === 0 ? base : Automerge.merge(Automerge.init(id), base)))
  for (let a = 0; a < nActors; a++) {
    docs[a] = Automerge.change(docs[a], d => {
      for (let i = 0; i < nElems; i++) {
        if (a > 0 && seed % 3 === 0 && i % 7 === 3) continue            // (holes: not every element is touched by everyone)
        d.l[i] = 'a' + a + '_' + i
      }
      for (let k = 0; k < nKeys; k++) d.m['k' + String(k).padStart(3, '0')] = a * 1000 + k
      if (seed % 2 === 1) d.t.insertAt(1, 'x', 'y')
    })
    if (seed % 5 === 2 && a === nActors - 1) {
      // a last actor that had seen nothing deletes a stretch while the others assign to it: those elements come back
      docs[a] = Automerge.change(docs[a], d => { d.l.splice(2, Math.min(5, d.l.length - 2)) })
    }
  }
  const per = docs.map(d => Automerge.getAllChanges(d).slice(baseChanges.length))
  const calls = [baseChanges.concat(per[0])]
  for (let a = 1; a < nActors; a++) calls.push(per[a])
  return { calls }
}
const b64 = u8 => Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength).toString('base64')
const lines = []
for (const spec of process.argv.slice(3)) {
  const f = spec.split(':')
  if (f[0] === 'c') {
    const { calls } = conflictScenario(+f[1], +f[2], +f[3], +f[4])
    for (let variant = 0; variant < 2; variant++) {
      // variant 1: the later actors' changes in ONE call (one scheduling pass, several merge calls)
      const plan = variant === 0 ? calls : [calls[0], [].concat(...calls.slice(1))]
      let backend = Backend.init()
      const patches = []
      for (const batch of plan) {
        const [b2, patch] = Backend.applyChanges(backend, batch)
        backend = b2
        patches.push(JSON.stringify(patch))
      }
      lines.push(JSON.stringify({ name: `${spec}#${variant}`, calls: plan.map(c => c.map(b64)), patches }))
    }
    console.error(`${spec}: ${calls.reduce((n, c) => n + c.length, 0)} changes`)
    continue
  }
  const changes = f[0] === 't' ? textScenario(+f[1], +f[2], +f[3], +f[4]) : f[0] === 'm' ? servedScenario(+f[1], +f[2], +f[3], +f[4]) : f[0] === 'l' ? listScenario(+f[1], +f[2], +f[3], +f[4]) : frontendScenario(+f[0], +f[1], +f[2], +f[3])
  const rnd = splitmix(0xABCD + (f[0] === 't' || f[0] === 'm' || f[0] === 'l' ? +f[1] : +f[0]))
  // (LOADED: every session twice -- as loaded, and with the hash graph rebuilt by a query before the first call, "+g")
  for (let run = 0; run < (process.env.LOADED ? 6 : 3); run++) {
    const variant = run % 3, graphFirst = run >= 3
    let order = changes.slice()
    if (variant === 2) {
      // local shuffles: a change may arrive before its dependencies and wait in the queue
      for (let i = 0; i + 1 < order.length; i++) {
        if (rnd() < 0.3) { const j = Math.min(order.length - 1, i + 1 + Math.floor(rnd() * 4)); [order[i], order[j]] = [order[j], order[i]] }
      }
    }
    const calls = []
    let i = 0
    while (i < order.length) {
      const r = rnd()
      let n = variant === 0 ? 1 + Math.floor(rnd() * 3) : (r < 0.3 ? 1 : r < 0.8 ? 2 + Math.floor(rnd() * 8) : 10 + Math.floor(rnd() * 60))
      if (variant === 1 && calls.length === 0) n = Math.floor(order.length / 2)   // a big first batch, then small ones
      calls.push(order.slice(i, i + n))
      i += n
    }
    let backend = Backend.init()
    const patches = []
    let doc = null
    if (process.env.LOADED) {
      // cut: after a third / the big first call / two thirds of the calls (changes queued at that moment are lost with the save, as
      // they are for any user of Backend.save)
      const cut = Math.max(1, variant === 0 ? Math.floor(calls.length / 3) : variant === 1 ? 1 : Math.floor(2 * calls.length / 3))
      try {
        for (const batch of calls.splice(0, cut)) backend = Backend.applyChanges(backend, batch)[0]
        doc = Backend.save(backend)
        backend = Backend.load(doc)
        if (graphFirst) Backend.getAllChanges(backend)   // (BackendDoc.getChanges rebuilds the hash graph in place, new.js:1922)
      } catch (e) { continue }
    }
    for (const batch of calls) {
      try {
        const [b2, patch] = Backend.applyChanges(backend, batch)
        backend = b2
        patches.push(JSON.stringify(patch))
      } catch (e) {
        patches.push({ error: String(e.message).split('\n')[0] })
        break
      }
    }
    const rec = { name: `${spec}#${variant}${graphFirst ? '+g' : ''}`, calls: calls.slice(0, patches.length).map(c => c.map(b64)), patches }
    if (doc) rec.doc = b64(doc)
    // (the hashes of the document's changes as the reference rebuilds them, computeHashGraph new.js:1887-1912: the oracle does not
    // restate that reconstruction and takes them from here, tests/oracle_lib.py OracleSession(doc, doc_hashes))
    if (doc) rec.doc_hashes = Buffer.concat(Backend.getAllChanges(Backend.load(doc)).map(c => Buffer.from(refColumnar.decodeChangeMeta(c, true).hash, 'hex'))).toString('base64')
    if (graphFirst) rec.graph = true
    lines.push(JSON.stringify(rec))
  }
  console.error(`${spec}: ${changes.length} changes`)
}
fs.writeFileSync(out, lines.join('\n') + '\n')
