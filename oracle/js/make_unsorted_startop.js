// TEST INFRASTRUCTURE (build container only): an actor whose changes carry DESCENDING startOps (legal for the backend, new.js:1550-1597 checks seq only) --
// the span tables of op-id lookup must come out sorted by startOp. Writes tests/golden/hand_unsorted_startop.json (changes in hex + the reference getPatch).
//   NODE_PATH=oracle/js_shims/node_modules node oracle/js/make_unsorted_startop.js
const { loadBackend } = require('./ref_loader')
const { Backend, columnar } = loadBackend()
const { encodeChange, decodeChange } = columnar
const fs = require('fs')
const A = 'aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa', B = 'bbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbb'
function mk(actor, seq, startOp, deps, ops) { return encodeChange({actor, seq, startOp, time: 0, message: '', deps, ops}) }
const hashOf = c => decodeChange(c).hash
const c1 = mk(A, 1, 10, [], [{action: 'set', obj: '_root', key: 'x', value: 1, datatype: 'int', pred: []}, {action: 'set', obj: '_root', key: 'w', value: 7, datatype: 'int', pred: []}])
const c2 = mk(A, 2, 1, [hashOf(c1)], [{action: 'set', obj: '_root', key: 'y', value: 2, datatype: 'int', pred: []}])
const c3 = mk(B, 1, 3, [hashOf(c2)], [{action: 'set', obj: '_root', key: 'z', value: 3, datatype: 'int', pred: []}])
const c4 = mk(A, 3, 5, [hashOf(c3)], [{action: 'set', obj: '_root', key: 'y', value: 4, datatype: 'int', pred: ['1@' + A]}])
const c5 = mk(A, 4, 20, [hashOf(c4)], [{action: 'set', obj: '_root', key: 'x', value: 9, datatype: 'int', pred: ['10@' + A]}, {action: 'del', obj: '_root', key: 'z', pred: ['3@' + B]}])
const chg = [c1, c2, c3, c4, c5]
const b64 = u8 => Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength).toString('base64')
let s = Backend.loadChanges(Backend.init(), chg)
const doc = Backend.save(s)
const out = { name: 'hand_unsorted_startop', note: 'one actor, startOps 10, 1, 5, 20 by seq: span tables must sort by startOp',
  changes: chg.map(b64), patch: JSON.stringify(Backend.getPatch(s)), doc: b64(doc), load_patch: JSON.stringify(Backend.getPatch(Backend.load(doc))),
  stock_equals_bigblock: true }
fs.writeFileSync(require('path').join(__dirname, '..', '..', 'tests', 'golden', 'hand_unsorted_startop.json'), JSON.stringify(out))
console.log(out.patch)
