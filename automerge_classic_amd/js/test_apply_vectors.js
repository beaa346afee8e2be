// Every Backend.applyChanges call of the reference's own test suites (tests/golden/ref_apply_vectors.json.gz: 1582 calls with the
// INCREMENTAL patch the unmodified reference returned, SURVEY.md 8f-2) through the JS host:
// node -> am355_napi.node -> engine (am355_apply_changes) -> record tables -> materialize.js. No reference tree needed (runs on the
// GPU box). A session is replayed call by call in one engine context; a call passes when JSON.stringify of the materialised patch's
// `diffs` (key order of every object included), `deps`, `maxOp`, `pendingChanges` equal the reference's and `clock` is deep-equal.
// The engine may refuse a call (the wrapper then serves it on the JS path): the session ends there.
//   node automerge_classic_amd/js/test_apply_vectors.js [file] [max calls per session] [max sessions]
'use strict'
const fs = require('fs')
const path = require('path')
const zlib = require('zlib')
const addon = require(path.join(__dirname, 'am355_napi.node'))
const { materialize } = require('./materialize.js')

const file = process.argv[2] || path.join(__dirname, '..', '..', 'tests', 'golden', 'ref_apply_vectors.json.gz')
const maxChain = parseInt(process.argv[3] || '1000000'), maxSessions = parseInt(process.argv[4] || '1000000')
const d = JSON.parse(zlib.gunzipSync(fs.readFileSync(file)).toString('utf8'))
const pool = d.pool.map(x => new Uint8Array(Buffer.from(x, 'base64')))
const V = d.vectors || []
const ctx = addon.create(parseInt(process.env.MI355X_DEVICE || '0'))

function samePatch(got, want) {
  const wk = Object.keys(want).filter(k => k !== 'actor' && k !== 'seq')   // (a call made by applyLocalChange adds these two)
  if (JSON.stringify(Object.keys(got)) !== JSON.stringify(wk)) return false
  for (const k of wk) {
    if (k === 'clock') {
      const a = Object.keys(got.clock).sort(), b = Object.keys(want.clock).sort()
      if (JSON.stringify(a) !== JSON.stringify(b) || a.some(x => got.clock[x] !== want.clock[x])) return false
    } else if (JSON.stringify(got[k]) !== JSON.stringify(want[k])) return false
  }
  return true
}

let equal = 0, refused = 0, rejected = 0, failed = 0, sessions = 0
const checked = new Set()
if (d.sessions) {
  // campaign files (oracle/make_apply_campaign.py): {pool, sessions: [{name, calls: [[pool index...]...], patches: [JSON text | {error}]}]}
  for (const s of d.sessions) {
    if (sessions >= maxSessions) break
    sessions++
    addon.reset(ctx)
    if (s.doc) {   // a session onto a LOADED document (apply_campaign_loaded.json.gz): Backend.load, then the calls onto it
      addon.loadDocument(ctx, new Uint8Array(Buffer.from(s.doc, 'base64')))
      addon.replay(ctx)
      if (s.graph) addon.hashGraphKnown(ctx, 1)   // (the reference had been asked for the document's changes first)
    }
    for (let ci = 0; ci < s.calls.length && ci < s.patches.length; ci++) {
      const id = s.name + '/' + ci
      checked.add(id)
      let patch
      try {
        addon.applyChanges(ctx, s.calls[ci].map(k => pool[k]))
        patch = materialize(addon.fetchApplyIR(ctx))
      } catch (e) {
        if (e.am355Code === -4) { refused++; break }
        if (e.am355Code === -3) { if (typeof s.patches[ci] === 'string') { failed++; console.error(`FAIL ${id}: rejected a batch the reference accepts: ${e.message}`) } else rejected++; break }
        throw e
      }
      if (typeof s.patches[ci] !== 'string') { failed++; console.error(`FAIL ${id}: accepted a batch the reference rejects`); break }
      if (!samePatch(patch, JSON.parse(s.patches[ci]))) { failed++; console.error(`FAIL ${id}: patch differs`); break }
      equal++
    }
  }
  addon.destroy(ctx)
  console.log(JSON.stringify({ sessions, calls: checked.size, equal, refused, rejected, failed }))
  process.exit(failed ? 1 : 0)
}
const hasChild = new Set(V.map(v => v.parent))
for (let leaf = 0; leaf < V.length && sessions < maxSessions; leaf++) {
  if (hasChild.has(leaf)) continue
  const chain = []
  for (let j = leaf; j !== -1; j = V[j].parent) chain.unshift(j)
  if (chain.length > maxChain) continue
  sessions++
  addon.reset(ctx)
  if (V[chain[0]].doc !== undefined) { addon.loadDocument(ctx, pool[V[chain[0]].doc]); addon.replay(ctx) }   // applyChanges onto Backend.load(doc)
  for (const j of chain) {
    const v = V[j]
    let patch
    try {
      addon.applyChanges(ctx, v.changes.map(k => pool[k]))
      patch = materialize(addon.fetchApplyIR(ctx))
    } catch (e) {
      if (e.am355Code === -4) { if (!checked.has(j)) { checked.add(j); refused++ } break }
      if (e.am355Code === -3) {
        if (!checked.has(j)) { checked.add(j); if (v.error === undefined) { failed++; console.error(`FAIL ${j}: rejected a batch the reference accepts: ${e.message}`) } else rejected++ }
        break
      }
      throw e
    }
    if (checked.has(j)) continue
    checked.add(j)
    if (v.patch === undefined) { failed++; console.error(`FAIL ${j}: accepted a batch the reference rejects`); break }
    if (!samePatch(patch, JSON.parse(v.patch))) { failed++; console.error(`FAIL ${j}: patch differs`); break }
    equal++
  }
}
addon.destroy(ctx)
console.log(JSON.stringify({ sessions, calls: checked.size, equal, refused, rejected, failed }))
process.exit(failed ? 1 : 0)
