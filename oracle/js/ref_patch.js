// TEST INFRASTRUCTURE (not product code). Runs the *unmodified reference* backend from /root/reference
// on a change log and prints/saves its getPatch() result. Only usable inside the build container
// (the reference tree does not travel to the GPU box); its outputs are committed under tests/golden/.
//
//   NODE_PATH=oracle/js_shims/node_modules [REF_BLOCK_SIZE=n] node oracle/js/ref_patch.js <log.bin> [--check-encode] [--out patch.json] [--save doc.bin] [--time N]
// (REF_BLOCK_SIZE: see ref_loader.js)
//
// Log file layout (little endian): u32 n_changes, u64 n_ops, u64 offsets[n+1], arena.
const fs = require('fs')
const { loadBackend } = require('./ref_loader')
const { Backend, columnar } = loadBackend()
const { decodeChange, encodeChange } = columnar
const zlib = require('zlib')

function readLog(path) {
  const buf = fs.readFileSync(path)
  const n = buf.readUInt32LE(0)
  const nOps = Number(buf.readBigUInt64LE(4))
  const offs = []
  for (let i = 0; i <= n; i++) offs.push(Number(buf.readBigUInt64LE(12 + 8 * i)))
  const base = 12 + 8 * (n + 1)
  const changes = []
  for (let i = 0; i < n; i++) changes.push(new Uint8Array(buf.buffer, buf.byteOffset + base + offs[i], offs[i + 1] - offs[i]))
  return { changes, nOps }
}

// Uncompressed (chunk type 1) form of a change, for byte comparison independent of the DEFLATE implementation
function rawForm(change) {
  if (change[8] !== 2) return Buffer.from(change)
  let off = 9, len = 0, shift = 0
  for (;;) { const b = change[off++]; len |= (b & 0x7f) << shift; shift += 7; if (!(b & 0x80)) break }
  const body = zlib.inflateRawSync(Buffer.from(change.buffer, change.byteOffset + off, len))
  const lenBytes = []
  let v = body.length
  do { let x = v & 0x7f; v = Math.floor(v / 128); if (v) x |= 0x80; lenBytes.push(x) } while (v)
  return Buffer.concat([Buffer.from(change.slice(0, 8)), Buffer.from([1]), Buffer.from(lenBytes), body])
}

const args = process.argv.slice(2)
const path = args[0]
const { changes, nOps } = readLog(path)
if (args.includes('--check-encode')) {
  // The generator's encoder must agree byte-for-byte with the reference's encodeChange
  let rows = 0
  for (let i = 0; i < changes.length; i++) {
    const decoded = decodeChange(changes[i])
    rows += decoded.ops.length
    const again = encodeChange(decoded)
    if (!rawForm(changes[i]).equals(rawForm(again))) {
      console.error(`change ${i}: generator bytes differ from reference encodeChange`)
      process.exit(1)
    }
  }
  if (rows !== nOps) { console.error(`op rows ${rows} != header n_ops ${nOps}`); process.exit(1) }
  console.error(`check-encode ok: ${changes.length} changes, ${rows} op rows`)
}
const timeIdx = args.indexOf('--time')
const reps = timeIdx >= 0 ? parseInt(args[timeIdx + 1]) : 0
let patch
{
  const t0 = process.hrtime.bigint()
  const state = Backend.loadChanges(Backend.init(), changes)
  const saveIdx = args.indexOf('--save')
  if (saveIdx >= 0) fs.writeFileSync(args[saveIdx + 1], Backend.save(state))  // the reference's document bytes, for am355_save parity
  const t1 = process.hrtime.bigint()
  patch = Backend.getPatch(state)
  const t2 = process.hrtime.bigint()
  console.error(`reference: loadChanges ${(Number(t1 - t0) / 1e6).toFixed(1)} ms, getPatch ${(Number(t2 - t1) / 1e6).toFixed(1)} ms, ` +
    `${nOps} ops -> ${(nOps / (Number(t2 - t0) / 1e9)).toFixed(0)} ops/s (first run, includes JIT warm-up)`)
}
if (reps > 0) {
  const times = []
  for (let r = 0; r < reps; r++) {
    const t0 = process.hrtime.bigint()
    const state = Backend.loadChanges(Backend.init(), changes)
    Backend.getPatch(state)
    times.push(Number(process.hrtime.bigint() - t0) / 1e9)
  }
  times.sort((a, b) => a - b)
  const med = times[Math.floor(times.length / 2)]
  console.error(`reference median of ${reps}: ${(med * 1e3).toFixed(1)} ms = ${(nOps / med).toFixed(0)} ops/s (1 core)`)
}
const outIdx = args.indexOf('--out')
const json = JSON.stringify(patch)
if (outIdx >= 0) fs.writeFileSync(args[outIdx + 1], json)
else process.stdout.write(json + '\n')
