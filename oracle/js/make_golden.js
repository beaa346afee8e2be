// TEST INFRASTRUCTURE (not product code). Generates tests/golden/*.json by running the UNMODIFIED reference
// (frontend + backend from /root/reference, under node with the stdlib shims in oracle/js_shims) on
// deterministic editing scenarios, and recording `Backend.getPatch(Backend.loadChanges(init, changes))`.
//
//   NODE_PATH=oracle/js_shims/node_modules node oracle/js/make_golden.js tests/golden
//
// Each fixture: { name, note, changes: [base64...], patch: <JSON.stringify of the stock reference patch>,
//                 doc: base64 of Backend.save(state), load_patch: JSON.stringify(getPatch(Backend.load(doc))),
//                 patch_bigblock: <same with REF_BLOCK_SIZE=1e8, see ref_loader.js>, stock_equals_bigblock }.
// The big-block variant is computed in a child process so both come from pristine module instances.
const fs = require('fs')
const path = require('path')
const { execFileSync } = require('child_process')
const { loadBackend } = require('./ref_loader')
const { Backend, columnar, Automerge: getAutomerge } = loadBackend()
const Automerge = getAutomerge()
const { encodeChange } = columnar

function splitmix(seed) {
  let s = BigInt(seed)
  const M = (1n << 64n) - 1n
  return () => {
    s = (s + 0x9e3779b97f4a7c15n) & M
    let z = s
    z = ((z ^ (z >> 30n)) * 0xbf58476d1ce4e5b9n) & M
    z = ((z ^ (z >> 27n)) * 0x94d049bb133111ebn) & M
    z = z ^ (z >> 31n)
    return Number(z >> 11n) / 9007199254740992
  }
}

function actorIds(rnd, n) {
  const out = []
  for (let i = 0; i < n; i++) {
    let s = 'abcdef'[Math.floor(rnd() * 6)]
    while (s.length < 32) s += '0123456789abcdef'[Math.floor(rnd() * 16)]
    out.push(s)
  }
  return out
}

const WORDS = ['alpha', 'beta', 'gamma', 'delta', 'x', 'y', 'zeta', 'k1', 'k2', '10', '9', '007', 'Ünï', '日本', '😀smile', 'tab\there', 'quote"q', 'back\\slash']

// One random mutation of the document through the real frontend proxies
function mutate(doc, rnd, depthLimit) {
  const pick = arr => arr[Math.floor(rnd() * arr.length)]
  const scalar = () => {
    const r = rnd()
    if (r < 0.25) return Math.floor(rnd() * 2000) - 1000
    if (r < 0.35) return rnd() * 1e6 - 5e5
    if (r < 0.45) return pick([true, false, null])
    if (r < 0.5) return new Date(1600000000000 + Math.floor(rnd() * 1e9))
    if (r < 0.55) return new Automerge.Counter(Math.floor(rnd() * 10))
    if (r < 0.6) return new Automerge.Int(Math.floor(rnd() * 100))
    if (r < 0.65) return new Automerge.Uint(Math.floor(rnd() * 100))
    if (r < 0.7) return new Automerge.Float64(Math.floor(rnd() * 100))
    return pick(WORDS) + Math.floor(rnd() * 10)
  }
  // counters inside lists hit a reference quirk (SURVEY.md §7 hard part 1) and the frontend refuses to delete them
  const noCounter = v => (v instanceof Automerge.Counter ? 7 : (v instanceof Date ? v.getTime() : v))
  const value = depth => {
    const r = rnd()
    if (depth < depthLimit && r < 0.12) return { [pick(WORDS)]: scalar(), n: scalar() }
    if (depth < depthLimit && r < 0.22) return [scalar(), scalar(), scalar()].map(noCounter)
    if (depth < depthLimit && r < 0.28) return new Automerge.Text(pick(WORDS) + 'text')
    return scalar()
  }
  // walk to a random container
  const walk = (obj, depth) => {
    const keys = Array.isArray(obj) || obj instanceof Automerge.Text ? [...Array(obj.length).keys()] : Object.keys(obj)
    const containers = keys.filter(k => {
      let v
      try { v = obj instanceof Automerge.Text ? obj.get(k) : obj[k] } catch (e) { return false }
      return v && typeof v === 'object' && !(v instanceof Date) && !(v instanceof Automerge.Counter)
    })
    if (containers.length > 0 && rnd() < 0.6 && depth < 4) {
      const k = pick(containers)
      return walk(obj instanceof Automerge.Text ? obj.get(k) : obj[k], depth + 1)
    }
    return [obj, depth]
  }
  return Automerge.change(doc, d => {
    const nOps = 1 + Math.floor(rnd() * 4)
    for (let i = 0; i < nOps; i++) {
      const [obj, depth] = walk(d, 0)
      if (obj instanceof Automerge.Text) {
        const r = rnd()
        if (obj.length > 0 && r < 0.3) obj.deleteAt(Math.floor(rnd() * obj.length), 1)
        else obj.insertAt(Math.floor(rnd() * (obj.length + 1)), ...(pick(WORDS) + 'ab').split(''))
      } else if (Array.isArray(obj)) {
        const r = rnd()
        if (obj.length > 0 && r < 0.25) obj.splice(Math.floor(rnd() * obj.length), 1)
        else if (obj.length > 0 && r < 0.5) {
          const idx = Math.floor(rnd() * obj.length)
          if (obj[idx] instanceof Automerge.Counter) obj[idx].increment(1 + Math.floor(rnd() * 3))
          else obj[idx] = noCounter(value(depth + 1))
        } else {
          const v = noCounter(value(depth + 1))
          obj.splice(Math.floor(rnd() * (obj.length + 1)), 0, v)
        }
      } else {
        const keys = Object.keys(obj)
        const r = rnd()
        if (keys.length > 0 && r < 0.15) delete obj[pick(keys)]
        else if (keys.length > 0 && r < 0.4) {
          const k = pick(keys)
          if (obj[k] instanceof Automerge.Counter) { if (rnd() < 0.5) obj[k].increment(1 + Math.floor(rnd() * 5)); else obj[k].decrement(1) }
          else obj[k] = value(depth + 1)
        } else obj[pick(WORDS)] = value(depth + 1)
      }
    }
  })
}

// n actors edit concurrently, merging pairwise at random; returns all binary changes in causal order
function frontendScenario(seed, nActors, steps, depthLimit) {
  const rnd = splitmix(seed)
  const ids = actorIds(rnd, nActors)
  let docs = ids.map(id => Automerge.init(id))
  docs[0] = Automerge.change(docs[0], d => { d.title = 'start'; d.list = [1, 2, 3]; d.text = new Automerge.Text('hello'); d.counter = new Automerge.Counter(5); d.nested = { a: { b: 1 } } })
  for (let i = 1; i < nActors; i++) docs[i] = Automerge.merge(docs[i], docs[0])
  for (let s = 0; s < steps; s++) {
    const a = Math.floor(rnd() * nActors)
    try { docs[a] = mutate(docs[a], rnd, depthLimit) } catch (e) { /* frontend refused this edit; skip it */ }
    if (rnd() < 0.35) {
      const b = Math.floor(rnd() * nActors)
      if (b !== a) docs[b] = Automerge.merge(docs[b], docs[a])
    }
  }
  let all = Automerge.init()
  for (let i = 0; i < nActors; i++) all = Automerge.merge(all, docs[i])
  return Automerge.getAllChanges(all)
}

// Text-editing scenario through the frontend: concurrent typing bursts + deletions
function textScenario(seed, nActors, rounds, burst) {
  const rnd = splitmix(seed)
  const ids = actorIds(rnd, nActors)
  let docs = ids.map(id => Automerge.init(id))
  docs[0] = Automerge.change(docs[0], d => { d.text = new Automerge.Text() })
  for (let i = 1; i < nActors; i++) docs[i] = Automerge.merge(docs[i], docs[0])
  for (let r = 0; r < rounds; r++) {
    for (let a = 0; a < nActors; a++) {
      docs[a] = Automerge.change(docs[a], d => {
        const pos = Math.floor(rnd() * (d.text.length + 1))
        const chars = []
        for (let i = 0; i < burst; i++) chars.push('abcdefghij klmnop'[Math.floor(rnd() * 17)])
        d.text.insertAt(pos, ...chars)
        const nd = Math.min(d.text.length, Math.floor(burst / 4))
        for (let i = 0; i < nd; i++) d.text.deleteAt(Math.floor(rnd() * d.text.length), 1)
      })
    }
    // full sync
    let all = Automerge.init()
    for (let i = 0; i < nActors; i++) all = Automerge.merge(all, docs[i])
    for (let i = 0; i < nActors; i++) docs[i] = Automerge.merge(docs[i], all)
  }
  return Automerge.getAllChanges(docs[0])
}

// Hand-built changes (the style of the reference's own backend tests) for cases the frontend cannot produce
// Multi-inserts whose values change byte length: mixed-width UTF-8 text typed in one change, lists of numbers pushed in one go
// (consecutive ops of one change, values of different LEB128 lengths / float64), plus concurrent edits and deletions.
function multibyteScenario() {
  const ids = ['aa11', 'bb22']
  let a = Automerge.from({ text: new Automerge.Text(), nums: [], mixed: [] }, ids[0])
  a = Automerge.change(a, d => {
    d.text.insertAt(0, ...'aé€😀bcßπ∑xyz日本語ok')
    d.nums.push(1, 2, 300, 70000, 5, -1, -200, 3.5, 2.25, 1e100, 7, 8)
    d.mixed.push('x', 'yy', 'é', 1, 2, true, false, null, 'z', new Date(86400000), new Date(1), new Automerge.Counter(3), 4)
  })
  let b = Automerge.merge(Automerge.init(ids[1]), a)
  a = Automerge.change(a, d => { d.text.insertAt(3, ...'ÄÖ12'); d.text.deleteAt(1); d.nums.insertAt(2, 128, 127, 16384) })
  b = Automerge.change(b, d => { d.text.insertAt(5, ...'→←ab'); d.text.deleteAt(9); d.nums.deleteAt(0); d.mixed.insertAt(1, 'é', 'è', 'ee') })
  a = Automerge.merge(a, b)
  a = Automerge.change(a, d => { d.text.insertAt(d.text.length, ...'end…') })
  return Automerge.getAllChanges(a)
}

function handBuilt() {
  const A = '01234567', B = '89abcdef', C = 'fedcba98'
  const hash = c => columnar.decodeChange(encodeChange(c)).hash
  const out = {}
  { // conflicting assignments, nested conflicting objects, counter with concurrent increments, delete of a conflict
    const c1 = { actor: A, seq: 1, startOp: 1, time: 0, deps: [], ops: [
      { action: 'set', obj: '_root', key: 'bird', value: 'magpie', pred: [] },
      { action: 'set', obj: '_root', key: 'cnt', value: 10, datatype: 'counter', pred: [] },
      { action: 'makeMap', obj: '_root', key: 'cfg', pred: [] },
      { action: 'set', obj: `3@${A}`, key: 'x', value: 1, datatype: 'uint', pred: [] },
      { action: 'makeList', obj: '_root', key: 'l', pred: [] },
      { action: 'set', obj: `5@${A}`, elemId: '_head', insert: true, value: 'a', pred: [] },
      { action: 'set', obj: `5@${A}`, elemId: `6@${A}`, insert: true, value: 'b', pred: [] },
      { action: 'set', obj: `5@${A}`, elemId: `7@${A}`, insert: true, value: 3.5, datatype: 'float64', pred: [] }] }
    const c2 = { actor: B, seq: 1, startOp: 9, time: 0, deps: [hash(c1)], ops: [
      { action: 'set', obj: '_root', key: 'bird', value: 'blackbird', pred: [`1@${A}`] },
      { action: 'inc', obj: '_root', key: 'cnt', value: 3, pred: [`2@${A}`] },
      { action: 'makeMap', obj: '_root', key: 'cfg', pred: [`3@${A}`] },
      { action: 'set', obj: `11@${B}`, key: 'y', value: -2, datatype: 'int', pred: [] },
      { action: 'set', obj: `5@${A}`, elemId: `6@${A}`, value: 'A', pred: [`6@${A}`] },
      { action: 'del', obj: `5@${A}`, elemId: `7@${A}`, pred: [`7@${A}`] }] }
    const c3 = { actor: C, seq: 1, startOp: 9, time: 0, deps: [hash(c1)], ops: [
      { action: 'set', obj: '_root', key: 'bird', value: 'robin', pred: [`1@${A}`] },
      { action: 'inc', obj: '_root', key: 'cnt', value: -1, pred: [`2@${A}`] },
      { action: 'set', obj: `3@${A}`, key: 'x', value: 2, datatype: 'uint', pred: [`4@${A}`] },
      { action: 'set', obj: `5@${A}`, elemId: `6@${A}`, value: 'Z', pred: [`6@${A}`] },
      { action: 'set', obj: `5@${A}`, elemId: `7@${A}`, value: 'kept', pred: [`7@${A}`] },
      { action: 'makeText', obj: `5@${A}`, elemId: `8@${A}`, insert: true, pred: [] },
      { action: 'set', obj: `14@${C}`, elemId: '_head', insert: true, value: 'q', pred: [] }] }
    const c4 = { actor: A, seq: 2, startOp: 16, time: 0, deps: [hash(c2), hash(c3)].sort(), ops: [
      { action: 'del', obj: '_root', key: 'bird', pred: [`9@${B}`] },
      { action: 'set', obj: '_root', key: 'ts', value: 1600000000000, datatype: 'timestamp', pred: [] },
      { action: 'set', obj: '_root', key: 'bytes', value: new Uint8Array([1, 2, 255]), pred: [] },
      { action: 'makeTable', obj: '_root', key: 'tbl', pred: [] },
      { action: 'makeMap', obj: `19@${A}`, key: 'row1', pred: [] },
      { action: 'set', obj: `20@${A}`, key: 'name', value: 'n', pred: [] }] }
    out.hand_conflicts = [c1, c2, c3, c4].map(encodeChange)
    out.hand_conflicts_pending = [c1, c4, c3].map(encodeChange) // c4 lacks dep c2 -> stays queued (no document fixture)
    out.hand_conflicts_shuffled = [c4, c3, c2, c1].map(encodeChange)
  }
  { // integer-like map keys and actor ids (JS property enumeration order), unicode keys beyond the BMP
    const N1 = '12345678', N2 = '00001234'
    const c1 = { actor: N2, seq: 1, startOp: 1, time: 0, deps: [], ops: [
      { action: 'set', obj: '_root', key: 'b', value: 1, pred: [] }, { action: 'set', obj: '_root', key: '10', value: 2, pred: [] },
      { action: 'set', obj: '_root', key: '9', value: 3, pred: [] }, { action: 'set', obj: '_root', key: 'a', value: 4, pred: [] },
      { action: 'set', obj: '_root', key: '01', value: 5, pred: [] }, { action: 'set', obj: '_root', key: '4294967295', value: 6, pred: [] },
      { action: 'set', obj: '_root', key: '4294967294', value: 7, pred: [] }, { action: 'set', obj: '_root', key: '😀', value: 8, pred: [] },
      { action: 'set', obj: '_root', key: '￮', value: 9, pred: [] }] }
    const c2 = { actor: N1, seq: 1, startOp: 10, time: 0, deps: [hash(c1)], ops: [
      { action: 'set', obj: '_root', key: '0', value: 'zero', pred: [] }] }
    out.hand_key_order = [c1, c2].map(encodeChange)
  }
  return out
}

function b64(u8) { return Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength).toString('base64') }

function refPatch(changes) {
  const state = Backend.loadChanges(Backend.init(), changes)
  return JSON.stringify(Backend.getPatch(state))
}

// Backend.save() of the replayed state, and getPatch() of Backend.load() of those bytes (SURVEY.md §8 row a21)
function refSaveLoad(changes) {
  const state = Backend.loadChanges(Backend.init(), changes)
  const doc = Backend.save(state)
  const loaded = Backend.load(doc)
  return { doc, patch: JSON.stringify(Backend.getPatch(loaded)) }
}

function main() {
  const outDir = process.argv[2]
  if (process.argv[3] === '--child') {
    // child mode: read changes from a fixture, print the patch of the big-block reference
    const fx = JSON.parse(fs.readFileSync(process.argv[4], 'utf8'))
    const chg = fx.changes.map(c => new Uint8Array(Buffer.from(c, 'base64')))
    const out = { patch: refPatch(chg) }
    if (fx.doc) out.load_patch = JSON.stringify(Backend.getPatch(Backend.load(new Uint8Array(Buffer.from(fx.doc, 'base64')))))
    process.stdout.write(JSON.stringify(out))
    return
  }
  fs.mkdirSync(outDir, { recursive: true })
  const scenarios = {}
  if (process.env.EXTRA === 'multibyte') {
    scenarios.frontend_multibyte_runs = { changes: multibyteScenario(), note: 'real frontend: multi-inserts whose values change byte length (mixed-width UTF-8 text, numbers of different widths, floats, mixed lists)' }
  } else if (process.env.CAMPAIGN) {
    // differential campaign: CAMPAIGN="seed:actors:steps:depth,..." (mixed) or "t:seed:actors:rounds:burst" (text) -> only these
    for (const spec of process.env.CAMPAIGN.split(',')) {
      const f = spec.split(':')
      if (f[0] === 't') scenarios[`campaign_text_${f[1]}`] = { changes: textScenario(+f[1], +f[2], +f[3], +f[4]), note: `campaign text ${spec}` }
      else scenarios[`campaign_mixed_${f[0]}`] = { changes: frontendScenario(+f[0], +f[1], +f[2], +f[3]), note: `campaign mixed ${spec}` }
    }
  } else {
  scenarios.frontend_mixed_3actors = { changes: frontendScenario(101, 3, 120, 2), note: 'real frontend, 3 actors, maps/lists/text/counters/tables, random merges' }
  scenarios.frontend_mixed_6actors = { changes: frontendScenario(202, 6, 260, 3), note: 'real frontend, 6 actors, deeper nesting' }
  scenarios.frontend_text_4actors = { changes: textScenario(303, 4, 6, 24), note: 'real frontend, Text typing bursts + deletes, synced rounds (multi-insert ops)' }
  scenarios.frontend_text_8actors = { changes: textScenario(404, 8, 5, 40), note: 'real frontend, 8 actors, crosses 600-op block boundaries' }
  const hb = handBuilt()
  for (const k of Object.keys(hb)) scenarios[k] = { changes: hb[k], note: 'hand-built changes via reference encodeChange', noDoc: k.includes('pending') }
  // delivery-order variants
  {
    const rnd = splitmix(7)
    const base = scenarios.frontend_mixed_3actors.changes.slice()
    for (let i = base.length - 1; i > 0; i--) { const j = Math.floor(rnd() * (i + 1)); [base[i], base[j]] = [base[j], base[i]] }
    scenarios.frontend_mixed_3actors_shuffled = { changes: base, note: 'same changes as frontend_mixed_3actors, shuffled delivery' }
    const dup = scenarios.frontend_text_4actors.changes.slice(0, 9)
    scenarios.frontend_text_4actors_dups = { changes: scenarios.frontend_text_4actors.changes.concat(dup), note: 'duplicate changes appended' }
  }
  }
  for (const name of Object.keys(scenarios)) {
    const sc = scenarios[name]
    const fx = { name, note: sc.note, changes: sc.changes.map(b64), patch: refPatch(sc.changes) }
    if (!sc.noDoc) {
      // the saved document (only when nothing is pending: save() of a state with queued changes drops them)
      const sl = refSaveLoad(sc.changes)
      fx.doc = b64(sl.doc)
      fx.load_patch = sl.patch
    }
    const file = path.join(outDir, name + '.json')
    fs.writeFileSync(file, JSON.stringify(fx))
    const big = JSON.parse(execFileSync(process.execPath, [__filename, outDir, '--child', file],
      { env: Object.assign({}, process.env, { REF_BLOCK_SIZE: '100000000' }), maxBuffer: 1 << 30 }).toString())
    fx.stock_equals_bigblock = (big.patch === fx.patch) && (!fx.doc || big.load_patch === fx.load_patch)
    if (!fx.stock_equals_bigblock) { fx.patch_bigblock = big.patch; fx.load_patch_bigblock = big.load_patch }
    fs.writeFileSync(file, JSON.stringify(fx))
    console.error(`${name}: ${sc.changes.length} changes, patch ${fx.patch.length} B, doc ${fx.doc ? Buffer.from(fx.doc, 'base64').length : 0} B, stock==bigblock: ${fx.stock_equals_bigblock}`)
  }
}
if (require.main === module) main()
else module.exports = { splitmix, frontendScenario, textScenario, Automerge, Backend }
