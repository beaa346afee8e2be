// TEST INFRASTRUCTURE (not product code). Counters and value-less rows INSIDE list / text objects: the whole-document patch of the
// UNMODIFIED reference (new.js:937-965 counter states, 1010-1018 remove -> update, 1026-1033 the `remove` edit of a visible row without
// a value) on hand-built changes (reference encodeChange), recorded for the oracle and the engine to reproduce.
//
//   NODE_PATH=oracle/js_shims/node_modules node oracle/js/make_list_quirk_golden.js tests/golden/list_quirks.json [tests/golden/apply_campaign_quirks.json.gz]
//
// File: { note, cases: [{ name, changes: [base64], patch | error, doc, load_patch | load_error }] } -- patch = JSON.stringify(
// Backend.getPatch(Backend.loadChanges(Backend.init(), changes))), doc = Backend.save of that state, load_patch = getPatch(load(doc)).
const fs = require('fs')
const { loadBackend } = require('./ref_loader')
const { Backend, columnar } = loadBackend()
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
const hashOf = c => columnar.decodeChange(encodeChange(c)).hash
function b64(u8) { return Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength).toString('base64') }

// A list (or text) edited in synced rounds by several actors: inserts of plain values and counters, increments (several per counter,
// concurrent ones), assignments over counters, deletions of counters that are being incremented, plain elements in between (so that
// multi-insert runs start and end at counters). The model only keeps what is needed to name valid preds.
function scenario(seed, nActors, nRounds, opts) {
  const rnd = splitmix(seed)
  const pick = arr => arr[Math.floor(rnd() * arr.length)]
  const actors = []
  for (let i = 0; i < nActors; i++) { let s = 'abcdef'[i % 6]; while (s.length < 8) s += '0123456789abcdef'[Math.floor(rnd() * 16)]; actors.push(s) }
  const A = actors[0]
  const listType = opts.text ? 'makeText' : 'makeList'
  const changes = []
  const c0 = { actor: A, seq: 1, startOp: 1, time: 0, deps: [], ops: [{ action: listType, obj: '_root', key: 'l', pred: [] },
    { action: 'set', obj: '_root', key: 'n', value: 1, datatype: 'counter', pred: [] }] }
  changes.push(c0)
  const obj = `1@${A}`
  let maxOp = 2, heads = [hashOf(c0)]
  const seqOf = {}; seqOf[A] = 1
  const elems = []   // { id, vals: [{ id, kind: 'plain' | 'counter', alive }], incs: [{ id, ctr, alive }] } in insertion (not document) order
  const plainValue = () => {
    const r = rnd()
    if (opts.text) return { value: pick(['a', 'b', 'c', 'é', '日']) }
    if (r < 0.4) return { value: Math.floor(rnd() * 100), datatype: 'int' }
    if (r < 0.6) return { value: pick(['x', 'yy', 'zzz']) }
    if (r < 0.7 && opts.pCounter > 0) return { value: Math.floor(rnd() * 9), datatype: 'counter', plainCounter: true }   // a counter nobody increments yet
    if (r < 0.8) return { value: true }
    return { value: Math.floor(rnd() * 1000), datatype: 'uint' }
  }
  for (let round = 0; round < nRounds; round++) {
    const roundChanges = [], effects = []
    const startOp = maxOp + 1
    let roundMax = maxOp
    for (const actor of actors) {
      if (round > 0 && rnd() < 0.25) continue
      const ops = []
      let ctr = startOp
      const nOps = 1 + Math.floor(rnd() * opts.opsPerChange)
      let lastInserted = null
      for (let k = 0; k < nOps; k++) {
        const r = rnd()
        // (opts.quirkRounds: from that round on only plain values, on elements that never held a counter -- the calls onto a loaded
        // document with counters in its lists)
        const plainPhase = opts.quirkRounds !== undefined && round >= opts.quirkRounds
        const live = plainPhase ? elems.filter(e => e.incs.length === 0 && e.vals.every(v => v.kind === 'plain')) : elems
        if (plainPhase) { opts = Object.assign({}, opts, { pCounter: 0, pInc: 0 }) }
        if (elems.length === 0 || r < opts.pInsert || live.length === 0) {
          // insert after the element this change inserted last (typing run), a random element, or _head
          const ref = lastInserted && rnd() < 0.6 ? lastInserted : (elems.length && rnd() < 0.7 ? pick(elems).id : '_head')
          const id = `${ctr}@${actor}`
          if (rnd() < opts.pCounter) {
            ops.push({ action: 'set', obj, elemId: ref, insert: true, value: Math.floor(rnd() * 20), datatype: 'counter', pred: [] })
            effects.push({ t: 'ins', id, kind: 'counter' })
          } else {
            const v = plainValue()
            ops.push({ action: 'set', obj, elemId: ref, insert: true, value: v.value, datatype: v.datatype, pred: [] })
            effects.push({ t: 'ins', id, kind: v.datatype === 'counter' ? 'counter' : 'plain' })
          }
          lastInserted = id
          ctr++
          continue
        }
        const e = pick(live)
        const aliveVals = e.vals.filter(v => v.alive)
        const counters = e.vals.filter(v => v.kind === 'counter' && (v.alive || rnd() < 0.15))
        if (r < opts.pInsert + opts.pInc && counters.length) {
          const c = pick(counters)
          ops.push({ action: 'inc', obj, elemId: e.id, value: 1 + Math.floor(rnd() * 5), pred: [c.id] })
          effects.push({ t: 'inc', elem: e, id: `${ctr}@${actor}`, ctr: c })
          ctr++
        } else if (r < opts.pInsert + opts.pInc + opts.pSet) {
          const pred = aliveVals.map(v => v.id)
          if (rnd() < opts.pCounter) {
            ops.push({ action: 'set', obj, elemId: e.id, value: Math.floor(rnd() * 20), datatype: 'counter', pred })
            effects.push({ t: 'set', elem: e, id: `${ctr}@${actor}`, kind: 'counter', pred: aliveVals })
          } else {
            const v = plainValue()
            ops.push({ action: 'set', obj, elemId: e.id, value: v.value, datatype: v.datatype, pred })
            effects.push({ t: 'set', elem: e, id: `${ctr}@${actor}`, kind: v.datatype === 'counter' ? 'counter' : 'plain', pred: aliveVals })
          }
          ctr++
        } else if (aliveVals.length) {
          // deletion: names the visible value ops -- or, like a frontend that sees increments as values, sometimes an increment too
          const pred = aliveVals.map(v => v.id)
          const predObjs = aliveVals.slice()
          const incs = e.incs.filter(i => i.alive)
          if (incs.length && rnd() < opts.pDelInc) { const i = pick(incs); pred.push(i.id); predObjs.push(i) }
          ops.push({ action: 'del', obj, elemId: e.id, pred })
          effects.push({ t: 'del', elem: e, pred: predObjs })
          ctr++
        }
      }
      if (!ops.length) continue
      seqOf[actor] = (seqOf[actor] || 0) + 1
      const ch = { actor, seq: seqOf[actor], startOp, time: 0, deps: heads.slice().sort(), ops }
      roundChanges.push(ch)
      roundMax = Math.max(roundMax, ctr - 1)
    }
    for (const ef of effects) {
      if (ef.t === 'ins') elems.push({ id: ef.id, vals: [{ id: ef.id, kind: ef.kind, alive: true }], incs: [] })
      else if (ef.t === 'inc') ef.elem.incs.push({ id: ef.id, ctr: ef.ctr, alive: true })
      else if (ef.t === 'set') { for (const p of ef.pred) p.alive = false; ef.elem.vals.push({ id: ef.id, kind: ef.kind, alive: true }) }
      else if (ef.t === 'del') for (const p of ef.pred) p.alive = false
    }
    if (roundChanges.length) { heads = roundChanges.map(hashOf); maxOp = roundMax; for (const c of roundChanges) changes.push(c) }
  }
  return changes
}

function handCases() {
  const A = '0a0a0a0a', B = 'b1b1b1b1', C = 'c2c2c2c2'
  const out = {}
  const base = extra => ({ actor: A, seq: 1, startOp: 1, time: 0, deps: [], ops: [{ action: 'makeList', obj: '_root', key: 'l', pred: [] }].concat(extra) })
  const L = `1@${A}`
  { // one counter, two increments in one change: remove edit at the first, update at the second (new.js:1026-1033, 1010-1018)
    const c1 = base([{ action: 'set', obj: L, elemId: '_head', insert: true, value: 1, datatype: 'counter', pred: [] }])
    const c2 = { actor: A, seq: 2, startOp: 3, time: 0, deps: [hashOf(c1)], ops: [
      { action: 'inc', obj: L, elemId: `2@${A}`, value: 2, pred: [`2@${A}`] }, { action: 'inc', obj: L, elemId: `2@${A}`, value: 3, pred: [`2@${A}`] }] }
    out.two_incs = [c1, c2]
  }
  { // plain, counter + inc, plain: the multi-insert run is cut at the counter
    const c1 = base([{ action: 'set', obj: L, elemId: '_head', insert: true, value: 7, datatype: 'counter', pred: [] },
      { action: 'set', obj: L, elemId: `2@${A}`, insert: true, value: 8, datatype: 'counter', pred: [] },
      { action: 'set', obj: L, elemId: `3@${A}`, insert: true, value: 9, datatype: 'counter', pred: [] },
      { action: 'set', obj: L, elemId: `4@${A}`, insert: true, value: 10, datatype: 'counter', pred: [] }])
    const c2 = { actor: B, seq: 1, startOp: 6, time: 0, deps: [hashOf(c1)], ops: [{ action: 'inc', obj: L, elemId: `3@${A}`, value: 5, pred: [`3@${A}`] }] }
    out.run_of_counters_one_incremented = [c1, c2]
    const c3 = { actor: C, seq: 1, startOp: 6, time: 0, deps: [hashOf(c1)], ops: [{ action: 'inc', obj: L, elemId: `5@${A}`, value: 1, pred: [`5@${A}`] }] }
    out.run_of_counters_last_incremented = [c1, c3]
    out.run_of_counters_two_incremented = [c1, c2, c3]
  }
  { // a counter deleted while another actor increments it: the increment is a visible row without a value -> `remove` edit
    const c1 = base([{ action: 'set', obj: L, elemId: '_head', insert: true, value: 'a', pred: [] },
      { action: 'set', obj: L, elemId: `2@${A}`, insert: true, value: 5, datatype: 'counter', pred: [] },
      { action: 'set', obj: L, elemId: `3@${A}`, insert: true, value: 'b', pred: [] }])
    const c2 = { actor: B, seq: 1, startOp: 5, time: 0, deps: [hashOf(c1)], ops: [{ action: 'del', obj: L, elemId: `3@${A}`, pred: [`3@${A}`] }] }
    const c3 = { actor: C, seq: 1, startOp: 5, time: 0, deps: [hashOf(c1)], ops: [{ action: 'inc', obj: L, elemId: `3@${A}`, value: 2, pred: [`3@${A}`] }] }
    out.counter_deleted_and_incremented = [c1, c2, c3]
    // ... and assigned again afterwards: remove -> update
    const c4 = { actor: A, seq: 2, startOp: 6, time: 0, deps: [hashOf(c2), hashOf(c3)].sort(), ops: [{ action: 'set', obj: L, elemId: `3@${A}`, value: 'again', pred: [] }] }
    out.counter_deleted_incremented_assigned = [c1, c2, c3, c4]
  }
  { // counter overwritten by a plain value while it is incremented; conflict of a counter and a plain value
    const c1 = base([{ action: 'set', obj: L, elemId: '_head', insert: true, value: 5, datatype: 'counter', pred: [] }])
    const c2 = { actor: B, seq: 1, startOp: 3, time: 0, deps: [hashOf(c1)], ops: [{ action: 'set', obj: L, elemId: `2@${A}`, value: 'plain', pred: [`2@${A}`] }] }
    const c3 = { actor: C, seq: 1, startOp: 3, time: 0, deps: [hashOf(c1)], ops: [{ action: 'inc', obj: L, elemId: `2@${A}`, value: 2, pred: [`2@${A}`] }] }
    out.counter_overwritten_and_incremented = [c1, c2, c3]
    const c4 = { actor: C, seq: 1, startOp: 3, time: 0, deps: [hashOf(c1)], ops: [{ action: 'set', obj: L, elemId: `2@${A}`, value: 9, datatype: 'counter', pred: [] },
      { action: 'inc', obj: L, elemId: `2@${A}`, value: 2, pred: [`3@${C}`] }] }
    out.two_counters_on_one_element = [c1, c4]
    out.two_counters_one_overwritten = [c1, c2, c4]
  }
  { // increment of an increment's counter after the element was deleted and the deletion names the increment
    const c1 = base([{ action: 'set', obj: L, elemId: '_head', insert: true, value: 5, datatype: 'counter', pred: [] },
      { action: 'inc', obj: L, elemId: `2@${A}`, value: 1, pred: [`2@${A}`] }, { action: 'set', obj: L, elemId: `2@${A}`, insert: true, value: 'tail', pred: [] }])
    const c2 = { actor: B, seq: 1, startOp: 5, time: 0, deps: [hashOf(c1)], ops: [{ action: 'del', obj: L, elemId: `2@${A}`, pred: [`3@${A}`] }] }
    out.increment_deleted = [c1, c2]
  }
  { // a link op on a list element (a visible row without a value)
    const c1 = base([{ action: 'set', obj: L, elemId: '_head', insert: true, value: 'a', pred: [] }, { action: 'makeMap', obj: '_root', key: 'm', pred: [] }])
    const c2 = { actor: B, seq: 1, startOp: 4, time: 0, deps: [hashOf(c1)], ops: [{ action: 'link', obj: L, elemId: `2@${A}`, child: `3@${A}`, pred: [`2@${A}`] }] }
    out.link_on_element = [c1, c2]
    const c3 = { actor: B, seq: 1, startOp: 4, time: 0, deps: [hashOf(c1)], ops: [{ action: 'link', obj: L, elemId: `2@${A}`, insert: true, child: `3@${A}`, pred: [] },
      { action: 'set', obj: L, elemId: `4@${B}`, insert: true, value: 'z', pred: [] }] }
    out.link_inserted = [c1, c3]
  }
  return out
}

function record(name, changes) {
  const bin = changes.map(encodeChange)
  const c = { name, changes: bin.map(b64) }
  let state
  try {
    state = Backend.loadChanges(Backend.init(), bin)
    c.patch = JSON.stringify(Backend.getPatch(state))
  } catch (e) { c.error = String(e.message); return c }
  try {
    const doc = Backend.save(state)
    c.doc = b64(doc)
    try { c.load_patch = JSON.stringify(Backend.getPatch(Backend.load(doc))) } catch (e) { c.load_error = String(e.message) }
  } catch (e) { c.save_error = String(e.message) }
  return c
}

function main() {
  const out = process.argv[2]
  const cases = []
  const hc = handCases()
  for (const k of Object.keys(hc)) cases.push(record('hand_' + k, hc[k]))
  let specs = [
    [11, 2, 6, { opsPerChange: 3, pInsert: 0.35, pInc: 0.35, pSet: 0.15, pCounter: 0.7, pDelInc: 0 }],
    [12, 3, 8, { opsPerChange: 4, pInsert: 0.3, pInc: 0.4, pSet: 0.15, pCounter: 0.6, pDelInc: 0 }],
    [13, 3, 10, { opsPerChange: 5, pInsert: 0.45, pInc: 0.25, pSet: 0.1, pCounter: 0.4, pDelInc: 0 }],
    [14, 4, 8, { opsPerChange: 4, pInsert: 0.3, pInc: 0.3, pSet: 0.2, pCounter: 0.5, pDelInc: 0.3 }],
    [15, 2, 12, { opsPerChange: 6, pInsert: 0.5, pInc: 0.3, pSet: 0.1, pCounter: 0.3, pDelInc: 0, text: true }],
    [16, 5, 6, { opsPerChange: 3, pInsert: 0.25, pInc: 0.45, pSet: 0.15, pCounter: 0.8, pDelInc: 0.2 }]
  ]
  if (process.env.QUIRK_WILD) {   // (campaigns only: deletions that name increments, many assignments over counters, long sessions)
    specs.length = 0
    specs.push([21, 3, 10, { opsPerChange: 4, pInsert: 0.2, pInc: 0.3, pSet: 0.25, pCounter: 0.7, pDelInc: 0.7 }],
      [22, 6, 8, { opsPerChange: 6, pInsert: 0.25, pInc: 0.35, pSet: 0.2, pCounter: 0.9, pDelInc: 0.5 }],
      [23, 2, 30, { opsPerChange: 8, pInsert: 0.3, pInc: 0.4, pSet: 0.1, pCounter: 0.5, pDelInc: 0.2 }],
      [24, 4, 12, { opsPerChange: 5, pInsert: 0.15, pInc: 0.5, pSet: 0.3, pCounter: 0.6, pDelInc: 0.4, text: true }])
  }
  // (QUIRK_SEED_BASE / QUIRK_REPS: a differential campaign with other seeds -- not the committed fixture)
  const seedBase = parseInt(process.env.QUIRK_SEED_BASE || '0'), reps = parseInt(process.env.QUIRK_REPS || '6')
  for (let rep = 0; rep < reps; rep++)
    for (const [seed, a, r, o] of specs) cases.push(record(`gen_${seed}_${rep}`, scenario(seedBase + seed * 1000 + rep, a, r, o)))
  // the same generated logs as sessions of Backend.applyChanges calls (a few changes per call): the patches of the live reference, in the
  // format of tests/golden/apply_campaign*.json.gz -- what the engine's incremental path must equal or refuse (counters inside lists in an
  // INCREMENTAL patch are left to the JS path; everything else of such a session is served)
  if (process.argv[3]) {
    const zlib = require('zlib')
    const pool = [], sessions = []
    const rnd = splitmix(99)
    for (let rep = 0; rep < (process.env.QUIRK_REPS ? reps : 4); rep++)
      for (const [seed, a, r, o] of specs) {
        const o2 = Object.assign({}, o, { pInc: o.pInc * (rep % 2 ? 0.3 : 1), pCounter: o.pCounter * (rep % 2 ? 0.5 : 1) })
        const bin = scenario(seedBase + seed * 7000 + rep, a, r + 4, o2).map(encodeChange)
        const calls = [], patches = []
        let state = Backend.init(), i = 0
        while (i < bin.length) {
          const k = Math.min(bin.length - i, 1 + Math.floor(rnd() * 3))
          const batch = bin.slice(i, i + k)
          calls.push(batch.map(c => { pool.push(b64(c)); return pool.length - 1 }))
          try { const [s2, patch] = Backend.applyChanges(state, batch); state = s2; patches.push(JSON.stringify(patch)) } catch (e) { patches.push({ error: String(e.message) }); break }
          i += k
        }
        sessions.push({ name: `q:${seed}:${rep}`, calls, patches })
      }
    // sessions onto LOADED documents that hold counters / rows without a value in their lists: the first rounds (with increments) are
    // saved and loaded by the reference, the later rounds (plain values on other elements) arrive in calls
    for (let rep = 0; rep < (process.env.QUIRK_REPS ? reps : 3); rep++)
      for (const [seed, a, r, o] of specs) {
        const o2 = Object.assign({}, o, { quirkRounds: r })
        const all = scenario(seedBase + seed * 9000 + rep, a, r + 6, o2)
        let nDoc = 0
        { // changes of the first r rounds: everything before the first change whose deps name a change of round r
          const o3 = Object.assign({}, o, { quirkRounds: r })
          nDoc = scenario(seedBase + seed * 9000 + rep, a, r, o3).length
        }
        const bin = all.map(encodeChange)
        const doc = Backend.save(Backend.loadChanges(Backend.init(), bin.slice(0, nDoc)))
        let state = Backend.load(doc), i = nDoc
        const calls = [], patches = []
        while (i < bin.length) {
          const k = Math.min(bin.length - i, 1 + Math.floor(rnd() * 3))
          const batch = bin.slice(i, i + k)
          calls.push(batch.map(c => { pool.push(b64(c)); return pool.length - 1 }))
          try { const [s2, patch] = Backend.applyChanges(state, batch); state = s2; patches.push(JSON.stringify(patch)) } catch (e) { patches.push({ error: String(e.message) }); break }
          i += k
        }
        // (doc_hashes: the hashes of the document's changes as the reference rebuilds them -- tests/oracle_lib.py OracleSession takes them from here)
        const docHashes = Buffer.concat(Backend.getAllChanges(Backend.load(doc)).map(c => Buffer.from(columnar.decodeChangeMeta(c, true).hash, 'hex'))).toString('base64')
        sessions.push({ name: `qdoc:${seed}:${rep}`, doc: b64(doc), doc_hashes: docHashes, calls, patches })
      }
    fs.writeFileSync(process.argv[3], zlib.gzipSync(JSON.stringify({ made_by: 'oracle/js/make_list_quirk_golden.js (unmodified reference, node ' + process.version + ')', pool, sessions })))
    console.error(`${sessions.length} sessions, ${sessions.reduce((x, y) => x + y.calls.length, 0)} calls`)
  }
  const n = { patch: 0, error: 0, load_patch: 0, load_error: 0 }
  for (const c of cases) for (const k of Object.keys(n)) if (k in c) n[k]++
  fs.writeFileSync(out, JSON.stringify({ note: 'counters / value-less rows inside lists: unmodified reference under node ' + process.version + ' (oracle/js/make_list_quirk_golden.js)', cases }))
  console.error(`${cases.length} cases:`, n)
  for (const c of cases) if (c.error || c.load_error || c.save_error) console.error('  ', c.name, c.error || c.load_error || c.save_error)
}
main()
