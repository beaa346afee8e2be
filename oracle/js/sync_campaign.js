// TEST INFRASTRUCTURE (build container only). Differential campaign for the sync protocol (SURVEY.md 8f-4): two peers with diverged
// histories run generateSyncMessage / receiveSyncMessage until they agree -- once with the unmodified reference backend on both
// sides, once with mi355x-backend (the engine, or its emulation preloaded) on one side and once on both. Every message must be
// byte-identical and every patch equal (the key order of `clock` aside) in all three runs.
//   NODE_PATH=oracle/js_shims/node_modules AUTOMERGE_BACKEND_PATH=/root/reference/backend [SYNC_LOADED=1] node oracle/js/sync_campaign.js [scenarios]
'use strict'
const path = require('path')
const { splitmix, Automerge } = require('./make_golden.js')
const Ref = require(path.join(process.env.AUTOMERGE_BACKEND_PATH || '/root/reference/backend'))
const Eng = require(path.join(__dirname, '..', '..', 'automerge_classic_amd', 'js', 'index.js'))

function scenario(seed) {
  const rnd = splitmix(seed)
  const pick = n => Math.floor(rnd() * n)
  let a = Automerge.from({ text: new Automerge.Text('shared'), notes: {}, list: ['x'] }, 'aaaa' + seed.toString(16).padStart(8, '0'))
  for (let i = 0; i < 5 + pick(20); i++) a = Automerge.change(a, d => { d.text.insertAt(pick(d.text.length + 1), ...'common '.split('')); d.notes['k' + pick(8)] = i })
  let b = Automerge.merge(Automerge.init('bbbb' + seed.toString(16).padStart(8, '0')), a)
  let c = Automerge.merge(Automerge.init('cccc' + seed.toString(16).padStart(8, '0')), a)
  const edit = (doc, tag, n) => {
    for (let i = 0; i < n; i++) {
      doc = Automerge.change(doc, d => {
        const r = rnd()
        if (r < 0.5) d.text.insertAt(pick(d.text.length + 1), ...(tag + i).split(''))
        else if (r < 0.7 && d.text.length > 0) d.text.deleteAt(pick(d.text.length))
        else if (r < 0.85) d.notes[tag + pick(6)] = { by: tag, n: i }
        else d.list.push(tag + i)
      })
    }
    return doc
  }
  a = edit(a, 'A', 10 + pick(60))
  b = edit(b, 'B', 10 + pick(60))
  c = edit(c, 'C', pick(15))
  if (rnd() < 0.5) b = Automerge.merge(b, c)   // a third author known to one side only
  if (rnd() < 0.3) a = edit(Automerge.merge(a, c), 'A2', pick(10))
  return [Automerge.getAllChanges(a), Automerge.getAllChanges(b)]
}

// SYNC_LOADED=1: both peers start from a SAVED document (Backend.load): the protocol's getChanges / getMissingDeps make the reference
// rebuild the document's hash graph, and the changes of the first message are applied onto the loaded document
const docOf = changes => Ref.save(Ref.loadChanges(Ref.init(), changes))
function syncRun(BackA, BackB, changesA, changesB) {
  let bA, bB
  if (process.env.SYNC_LOADED) { bA = BackA.load(docOf(changesA)); bB = BackB.load(docOf(changesB)) }
  else { bA = BackA.loadChanges(BackA.init(), changesA); bB = BackB.loadChanges(BackB.init(), changesB) }
  let sA = BackA.initSyncState(), sB = BackB.initSyncState()
  const log = []
  for (let round = 0; round < 30; round++) {
    let msgA, msgB, patch
    ;[sA, msgA] = BackA.generateSyncMessage(bA, sA)
    if (msgA) { [bB, sB, patch] = BackB.receiveSyncMessage(bB, sB, msgA); log.push(['A->B', Buffer.from(msgA).toString('hex'), patch]) }
    ;[sB, msgB] = BackB.generateSyncMessage(bB, sB)
    if (msgB) { [bA, sA, patch] = BackA.receiveSyncMessage(bA, sA, msgB); log.push(['B->A', Buffer.from(msgB).toString('hex'), patch]) }
    if (!msgA && !msgB) break
  }
  log.push(['heads', JSON.stringify(BackA.getHeads(bA)), JSON.stringify(BackB.getHeads(bB))])
  log.push(['docs', JSON.stringify(BackA.getPatch(bA).diffs), JSON.stringify(BackB.getPatch(bB).diffs)])
  return log
}

function samePatch(x, y) {
  if (x === null || y === null) return x === y
  const kx = Object.keys(x), ky = Object.keys(y)
  if (JSON.stringify(kx) !== JSON.stringify(ky)) return false
  for (const k of kx) {
    if (k === 'clock') { if (JSON.stringify(Object.entries(x.clock).sort()) !== JSON.stringify(Object.entries(y.clock).sort())) return false }
    else if (JSON.stringify(x[k]) !== JSON.stringify(y[k])) return false
  }
  return true
}
function sameLog(l1, l2) {
  if (l1.length !== l2.length) return `length ${l1.length} vs ${l2.length}`
  for (let i = 0; i < l1.length; i++) {
    if (l1[i][0] !== l2[i][0]) return `step ${i}: direction`
    if (l1[i][0] === 'heads' || l1[i][0] === 'docs') { if (l1[i][1] !== l2[i][1] || l1[i][2] !== l2[i][2]) return `step ${i}: final ${l1[i][0]} differ` }
    else {
      if (l1[i][1] !== l2[i][1]) return `step ${i}: message bytes differ (${l1[i][0]})`
      if (!samePatch(l1[i][2], l2[i][2])) return `step ${i}: patch differs (${l1[i][0]})`
    }
  }
  return null
}

const n = parseInt(process.argv[2] || '12')
let identical = 0, disagree = 0, messages = 0
for (let s = 0; s < n; s++) {
  const [ca, cb] = scenario(0x5C00 + s)
  const refLog = syncRun(Ref, Ref, ca, cb)
  messages += refLog.length - 2
  for (const [name, BA, BB] of [['engine receives', Ref, Eng], ['engine sends', Eng, Ref], ['engine on both sides', Eng, Eng]]) {
    let why
    try { why = sameLog(refLog, syncRun(BA, BB, ca, cb)) } catch (e) { why = 'threw ' + e.message }
    if (why) { disagree++; console.error(`scenario ${s} (${name}): ${why}`) } else identical++
  }
}
console.log(`sync campaign: ${n} scenarios, ${messages} messages in the reference runs, ${identical} runs identical, DISAGREE ${disagree}`)
if (Eng._counters) console.log('served by: ' + JSON.stringify(Eng._counters))
process.exit(disagree ? 1 : 0)
