// Patch IR (the record tables of include/am355.h, as written by the device) -> the patch object the frontend consumes.
//
// This is presentation of results computed on the GPU; the shapes and key orders are the reference's, so that
// JSON.stringify(materialize(ir)) === JSON.stringify(Backend.getPatch(state)) and assert.deepStrictEqual holds:
//   envelope                       backend/new.js:2064-2067   {maxOp, clock, deps, pendingChanges, diffs}
//   object patch                   backend/new.js:726-732     {objectId, type, props} | {objectId, type, edits}
//   map props                      backend/new.js:1035-1039   props[key][opId] = valueDiff
//   list edits                     backend/new.js:747-782     insert / multi-insert / update (appendEdit); remove in the patches of applyChanges
//   values                         backend/columnar.js:300-329 decodeValue, new.js:963 (counter), :971 ({type: 'value', ...})
// Values are real JS values (float64 NaN / Infinity stay what they are, byte arrays are Uint8Arrays): nothing goes through JSON text.
'use strict'

const OBJ_WORDS = 8, MAP_WORDS = 10, EDIT_WORDS = 10
const MAP_COUNTER = 1, MAP_CHILD = 2, MAP_EMPTY = 4, EDIT_UPDATE = 1, EDIT_CONT = 2, EDIT_CHILD = 4, EDIT_REMOVE = 8, EDIT_MULTI = 16, EDIT_COUNTER = 32
const TYPE_NAME = { 0: 'map', 2: 'list', 4: 'text', 6: 'table' }
const HEX = []
for (let i = 0; i < 256; i++) HEX.push((i < 16 ? '0' : '') + i.toString(16))
const MAX_SAFE = Number.MAX_SAFE_INTEGER
const ASCII = []
for (let i = 0; i < 128; i++) ASCII.push(String.fromCharCode(i))

function unsupported(msg) {
  // inputs the reference itself would throw on (number out of range, bad float length): the caller replays them on the JS path
  const e = new RangeError(msg)
  e.am355Code = -4
  return e
}

function hexOf(bytes, start, end) {
  let s = ''
  for (let i = start; i < end; i++) s += HEX[bytes[i]]
  return s
}

class Materializer {
  constructor(ir) {
    this.ir = ir
    this.obj = new Uint32Array(ir.objects)
    this.map = new Uint32Array(ir.map)
    this.mapCounter = ir.map.byteLength ? new DataView(ir.map) : null
    this.edit = new Uint32Array(ir.edits)
    this.arena = new Uint8Array(ir.arena)
    this.arenaBuf = Buffer.from(ir.arena)
    this.decoder = new TextDecoder('utf-8')
    const actorOff = new Uint32Array(ir.actorOff), actorBytes = new Uint8Array(ir.actorBytes)
    this.actors = []
    for (let a = 0; a < ir.nActors; a++) this.actors.push('@' + hexOf(actorBytes, actorOff[a], actorOff[a + 1]))
    this.depth = 0
  }

  opId(ctr, actor) { return ctr + this.actors[actor] }

  str(off, len) {
    const a = this.arena
    if (len === 1 && a[off] < 0x80) return ASCII[a[off]]
    let ascii = true
    for (let i = off, e = off + len; i < e; i++) if (a[i] >= 0x80) { ascii = false; break }
    if (ascii) return this.arenaBuf.latin1Slice(off, off + len)
    return this.decoder.decode(a.subarray(off, off + len))   // utf8ToString of the reference (encoding.js): replacement on malformed input
  }

  // LEB128 as Decoder.readUint53 / readInt53 (encoding.js:341-488): value with the 53-bit range check
  leb(off, len, signed) {
    const a = this.arena
    let result = 0, mul = 1
    for (let i = 0; i < len; i++) {
      const b = a[off + i]
      if (!(b & 0x80)) {
        if (signed && (b & 0x40)) result += ((b & 0x7f) - 0x80) * mul
        else result += (b & 0x7f) * mul
        if (result > MAX_SAFE || result < -MAX_SAFE || i > 9) throw unsupported('number out of range')
        return result
      }
      result += (b & 0x7f) * mul
      mul *= 128
      if (i >= 9) throw unsupported('number out of range')
    }
    throw unsupported('buffer ended with incomplete number')
  }

  // {value, datatype?} exactly as decodeValue (key order: value, then datatype)
  decode(tl, off) {
    if (tl === 0) return { value: null }
    if (tl === 1) return { value: false }
    if (tl === 2) return { value: true }
    const tag = tl & 15, len = tl >>> 4
    switch (tag) {
      case 6: return { value: this.str(off, len) }
      case 3: return { value: this.leb(off, len, false), datatype: 'uint' }
      case 4: return { value: this.leb(off, len, true), datatype: 'int' }
      case 5:
        if (len !== 8) throw unsupported(`Invalid length for floating point number: ${len}`)
        return { value: new DataView(this.ir.arena, off, 8).getFloat64(0, true), datatype: 'float64' }
      case 8: return { value: this.leb(off, len, true), datatype: 'counter' }
      case 9: return { value: this.leb(off, len, true), datatype: 'timestamp' }
      default: return { value: this.arena.slice(off, off + len), datatype: tag }
    }
  }

  valueDiff(tl, off, child) {
    if (child) return this.object(off)
    if ((tl & 15) === 6) return { type: 'value', value: this.str(off, tl >>> 4) }   // (the common case without the intermediate object)
    const v = this.decode(tl, off)
    return v.datatype === undefined ? { type: 'value', value: v.value } : { type: 'value', value: v.value, datatype: v.datatype }
  }

  sameKey(i, j) {
    const m = this.map, a = this.arena
    const len = m[i * MAP_WORDS + 3]
    if (len !== m[j * MAP_WORDS + 3]) return false
    const p = m[i * MAP_WORDS + 2], q = m[j * MAP_WORDS + 2]
    if (p === q) return true
    for (let k = 0; k < len; k++) if (a[p + k] !== a[q + k]) return false
    return true
  }

  props(begin, end) {
    const m = this.map, props = {}
    for (let i = begin; i < end;) {
      let j = i + 1
      while (j < end && this.sameKey(i, j)) j++
      const values = {}
      for (let k = i; k < j; k++) {
        const w = k * MAP_WORDS, flags = m[w + 6]
        if (flags & MAP_EMPTY) continue   // incremental patch: the key is left without a value, `props[key] = {}` (new.js:1037)
        const id = this.opId(m[w], m[w + 1])
        if (flags & MAP_COUNTER) {
          const lo = this.mapCounter.getUint32(k * 40 + 32, true), hi = this.mapCounter.getInt32(k * 40 + 36, true)
          values[id] = { type: 'value', datatype: 'counter', value: hi * 4294967296 + lo }
        } else {
          values[id] = this.valueDiff(m[w + 4], m[w + 5], (flags & MAP_CHILD) !== 0)
        }
      }
      const ko = m[i * MAP_WORDS + 2], kl = m[i * MAP_WORDS + 3]
      // (a key that starts with U+FEFF loses it in the reference's utf8ToString and then collides with other keys: JS path)
      if (kl >= 3 && this.arena[ko] === 0xef && this.arena[ko + 1] === 0xbb && this.arena[ko + 2] === 0xbf) throw unsupported('map key starts with a byte order mark')
      props[this.str(ko, kl)] = values   // (integer-like keys take their JS property order by themselves)
      i = j
    }
    return props
  }

  edits(begin, end) {
    const e = this.edit, out = [], actors = this.actors
    for (let k = begin; k < end;) {
      const w = k * EDIT_WORDS, flags = e[w], index = e[w + 1]
      const first = e[w + 6], tl = e[w + 7], off = e[w + 8]
      const count = e[w + EDIT_WORDS + 6] - first
      let j = k + 1
      while (j < end && (e[j * EDIT_WORDS] & EDIT_CONT)) j++   // further records of the same multi-insert (its values change length)
      if (flags & EDIT_REMOVE) {
        out.push({ action: 'remove', index, count })   // incremental patch (new.js:1029, 775-777)
      } else if (count >= 2 || j > k + 1 || (flags & EDIT_MULTI)) {   // (EDIT_MULTI: a multi-insert that lost its second value, new.js:812-814)
        let values
        if (j === k + 1 && (tl & 15) === 6 && (tl >>> 4) === 1) {
          // the common record: a run of typed single-byte characters, back to back in the arena
          const a = this.arena
          values = new Array(count)
          for (let i = 0, o = off; i < count; i++, o++) values[i] = a[o] < 0x80 ? ASCII[a[o]] : this.str(o, 1)
        } else {
          values = []
          for (let r = k; r < j; r++) {
            const rw = r * EDIT_WORDS, rtl = e[rw + 7], len = rtl >>> 4
            let roff = e[rw + 8]
            const rcount = e[rw + EDIT_WORDS + 6] - e[rw + 6]
            if (e[rw] & EDIT_COUNTER) values.push((e[rw + 9] | 0) * 4294967296 + roff)   // the total of a counter inside a list: one value per record
            else if ((rtl & 15) === 6) for (let i = 0; i < rcount; i++, roff += len) values.push(this.str(roff, len))
            else for (let i = 0; i < rcount; i++, roff += len) values.push(this.decode(rtl, roff).value)
          }
        }
        // (every key at once, in the reference's order -- action, index, elemId, [datatype,] values -- : one hidden class per form, no
        // transition and no out-of-object property store per edit)
        let datatype
        if (flags & EDIT_COUNTER) datatype = 'counter'
        else if ((tl & 15) !== 6) { const head = this.decode(tl, off); if (head.datatype) datatype = head.datatype }   // only truthy datatypes (new.js:762)
        const elemId = e[w + 4] + actors[e[w + 5]]
        out.push(datatype === undefined ? { action: 'multi-insert', index, elemId, values } : { action: 'multi-insert', index, elemId, datatype, values })
      } else if (flags & EDIT_COUNTER) {
        // (a counter inside a list: its total, new.js:963 -- low word in the value-offset field, high word behind it; rare: kept out of
        // the two ordinary branches below)
        const value = { type: 'value', datatype: 'counter', value: (e[w + 9] | 0) * 4294967296 + off }
        if (flags & EDIT_UPDATE) out.push({ action: 'update', index, opId: this.opId(e[w + 2], e[w + 3]), value })
        else out.push({ action: 'insert', index, elemId: this.opId(e[w + 4], e[w + 5]), opId: this.opId(e[w + 2], e[w + 3]), value })
      } else if (flags & EDIT_UPDATE) {
        out.push({ action: 'update', index, opId: this.opId(e[w + 2], e[w + 3]), value: this.valueDiff(tl, off, (flags & EDIT_CHILD) !== 0) })
      } else {
        out.push({ action: 'insert', index, elemId: this.opId(e[w + 4], e[w + 5]), opId: this.opId(e[w + 2], e[w + 3]),
                   value: this.valueDiff(tl, off, (flags & EDIT_CHILD) !== 0) })
      }
      k = j
    }
    return out
  }

  object(oi) {
    if (oi >= this.ir.nObjects) throw new Error('patch IR: object index out of range')
    if (++this.depth > 10000) throw unsupported('object nesting too deep')
    const o = this.obj, w = oi * OBJ_WORDS
    const type = oi === 0 ? 0 : o[w + 2]
    const res = { objectId: oi === 0 ? '_root' : this.opId(o[w], o[w + 1]), type: TYPE_NAME[type] === undefined ? null : TYPE_NAME[type] }
    if (oi !== 0 && (type === 2 || type === 4)) res.edits = this.edits(o[w + 5], o[w + 6])
    else res.props = this.props(o[w + 3], o[w + 4])
    this.depth--
    return res
  }

  patch() {
    const ir = this.ir, clock = {}
    const clockActor = new Uint32Array(ir.clockActor), clockSeq = new Float64Array(ir.clockSeq)
    for (let i = 0; i < clockActor.length; i++) clock[this.actors[clockActor[i]].slice(1)] = clockSeq[i]
    const heads = new Uint8Array(ir.heads), deps = []
    for (let i = 0; i + 32 <= heads.length; i += 32) deps.push(hexOf(heads, i, i + 32))
    return { maxOp: ir.maxOp, clock, deps, pendingChanges: ir.pending, diffs: this.object(0) }
  }
}

function materialize(ir) {
  return new Materializer(ir).patch()
}

module.exports = { materialize }
