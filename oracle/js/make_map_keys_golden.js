// TEST INFRASTRUCTURE (not product code). A root map (and one nested map) of ~900 keys of many shapes -- lengths 1..40, long common
// prefixes, keys that differ only behind their sixteenth byte, keys of one length that differ in one position, multi-byte UTF-8,
// supplementary-plane characters, integer-like keys -- assigned by two actors with conflicts: the order of the map records of the
// whole-document patch (UTF-16 code unit order of the keys, then op id; new.js:84, 1035-1039) through every pass the engine's map
// sort may skip or keep (MapKeyStats). Patches of the UNMODIFIED reference, in the format of make_golden.js.
//   NODE_PATH=oracle/js_shims/node_modules node oracle/js/make_map_keys_golden.js tests/golden/map_keys_mixed.json
// With a third argument s: every s-th key only -- s = 2 (~440 keys: the engine ranks 257 .. 512 map records by comparison in LDS,
// k_map_sort_small) and s = 5 (~180 keys: one workgroup ranks and finishes up to 256, map_small_finish); round 6:
//   ... make_map_keys_golden.js tests/golden/map_keys_mid.json 2 ;  ... tests/golden/map_keys_small.json 5
const fs = require('fs')
const { loadBackend } = require('./ref_loader')
const { Backend, columnar } = loadBackend()
const { encodeChange } = columnar
const hashOf = c => columnar.decodeChange(encodeChange(c)).hash
function b64(u8) { return Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength).toString('base64') }

const keys = []
for (let i = 0; i < 200; i++) keys.push('k' + String(i).padStart(5, '0'))                 // one length, two constant bytes
for (let i = 0; i < 120; i++) keys.push('a-very-long-common-prefix/' + i.toString(36))  // differ behind byte 16
for (let i = 0; i < 100; i++) keys.push('x'.repeat(1 + (i % 40)) + (i % 7))             // every length
for (let i = 0; i < 80; i++) keys.push(String(i * 37))                                  // integer-like (JS property order!)
for (let i = 0; i < 60; i++) keys.push(['é', '日本', '😀', '￮', 'ß', 'Ω'][i % 6] + i)
for (let i = 0; i < 100; i++) keys.push('same-len-' + String.fromCharCode(97 + (i % 26)) + String.fromCharCode(97 + Math.floor(i / 26)) + '-tail')
for (let i = 0; i < 120; i++) keys.push(String.fromCharCode(33 + (i % 90)))               // one byte (with repeats -> conflicts by assignment below)
for (let i = 0; i < 60; i++) keys.push('0123456789abcdef' + 'z'.repeat(i % 5) + i)      // equal in the first sixteen bytes
const stride = parseInt(process.argv[3] || '1')
const uniq = Array.from(new Set(keys)).filter((k, i) => i % stride === 0)

const A = '0a0a0a0a', B = 'b1b1b1b1'
const ops1 = [{ action: 'makeMap', obj: '_root', key: 'nested', pred: [] }]
uniq.forEach((k, i) => ops1.push({ action: 'set', obj: i % 5 === 0 ? `1@${A}` : '_root', key: k, value: i, datatype: 'int', pred: [] }))
const c1 = { actor: A, seq: 1, startOp: 1, time: 0, deps: [], ops: ops1 }
// B overwrites every third key it saw, A concurrently every fourth: conflicts on every twelfth
const idOf = i => `${2 + i}@${A}`
const ops2 = [], ops3 = []
uniq.forEach((k, i) => { if (i % 3 === 0) ops2.push({ action: 'set', obj: i % 5 === 0 ? `1@${A}` : '_root', key: k, value: 'b' + i, pred: [idOf(i)] }) })
uniq.forEach((k, i) => { if (i % 4 === 0) ops3.push({ action: 'set', obj: i % 5 === 0 ? `1@${A}` : '_root', key: k, value: 'a' + i, pred: [idOf(i)] }) })
const c2 = { actor: B, seq: 1, startOp: 2 + uniq.length, time: 0, deps: [hashOf(c1)], ops: ops2 }
const c3 = { actor: A, seq: 2, startOp: 2 + uniq.length, time: 0, deps: [hashOf(c1)], ops: ops3 }
const changes = [c1, c2, c3].map(encodeChange)
const state = Backend.loadChanges(Backend.init(), changes)
const doc = Backend.save(state)
const fx = { name: require('path').basename(process.argv[2], '.json'), note: 'hand-built: ' + uniq.length + ' map keys of many shapes, two actors, conflicts (oracle/js/make_map_keys_golden.js)',
  changes: changes.map(b64), patch: JSON.stringify(Backend.getPatch(state)), doc: b64(doc), load_patch: JSON.stringify(Backend.getPatch(Backend.load(doc))),
  stock_equals_bigblock: true }
fs.writeFileSync(process.argv[2], JSON.stringify(fx))
console.error(uniq.length + ' keys, patch ' + fx.patch.length + ' B, doc ' + doc.byteLength + ' B')
