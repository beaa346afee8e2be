// One worker process per GPU of the sharded bulk replay (js/sharded.js forks these; SURVEY.md §8e, include/am355.h am355_shard_init /
// am355_sharded_replay). The worker owns one engine context on its GPU; the collective -- ncclAllGather of the patch-IR fragments over
// xGMI -- runs inside the library, between the workers' contexts. Messages from the parent:
//   {type: 'id'}                        rank 0 only: make the 128-byte RCCL unique id           -> {type: 'id', id: hex}
//   {type: 'init', id: hex}             ncclCommInitRank                                         -> {type: 'ready'}
//   {type: 'replay', file, kind}        stage the batch in `file` (kind 'changes': the log layout of bench_e2e.js; 'document': Backend.save
//                                       bytes), sharded replay; rank 0 stitches and materialises  -> {type: 'done', patch?: JSON text, ms, fragmentBytes?}
//   {type: 'close'}                                                                              -> exits
'use strict'
const fs = require('fs')
const path = require('path')
const addon = require(path.join(__dirname, 'am355_napi.node'))
const { materialize } = require('./materialize.js')

const rank = parseInt(process.env.AM355_SHARD_RANK), world = parseInt(process.env.AM355_SHARD_WORLD)
const device = process.env.AM355_SHARD_DEVICE !== undefined ? parseInt(process.env.AM355_SHARD_DEVICE) : rank
const ctx = addon.create(device)

function readChanges(file) {
  const buf = fs.readFileSync(file)
  const n = buf.readUInt32LE(0)
  const base = 12 + 8 * (n + 1)
  const changes = []
  let prev = Number(buf.readBigUInt64LE(12))
  for (let i = 0; i < n; i++) {
    const next = Number(buf.readBigUInt64LE(12 + 8 * (i + 1)))
    changes.push(new Uint8Array(buf.buffer, buf.byteOffset + base + prev, next - prev))
    prev = next
  }
  return changes
}

process.on('message', msg => {
  try {
    if (msg.type === 'id') {
      process.send({ type: 'id', id: Buffer.from(addon.shardUniqueId()).toString('hex') })
    } else if (msg.type === 'init') {
      addon.shardInit(ctx, new Uint8Array(Buffer.from(msg.id, 'hex')), rank, world)
      process.send({ type: 'ready' })
    } else if (msg.type === 'replay') {
      // (msg.reps > 1, bench_sharded.js: the engine part -- stage, sharded replay, record tables to the host -- repeated on the batch
      // read once; the times of the last repetition are reported)
      const input = msg.kind === 'document' ? new Uint8Array(fs.readFileSync(msg.file)) : readChanges(msg.file)
      const reps = msg.reps > 1 ? msg.reps : 1
      let error = null, patch, ir, times = null
      const t0 = process.hrtime.bigint()
      for (let r = 0; r < reps && !error; r++) {
        const a = process.hrtime.bigint()
        let staged = true
        try {
          if (msg.kind === 'document') addon.loadDocument(ctx, input)
          else addon.loadChanges(ctx, input)
        } catch (e) {
          staged = false   // (the collective is entered all the same: the library tells the other ranks that this one failed)
          addon.reset(ctx)
        }
        const b = process.hrtime.bigint()
        try {
          addon.shardedReplay(ctx, false)
          const c = process.hrtime.bigint()
          if (!staged) error = 'staging failed'
          else if (rank === 0) ir = addon.fetchIR(ctx, true)
          times = { stage: Number(b - a) / 1e6, replay: Number(c - b) / 1e6, fetch: Number(process.hrtime.bigint() - c) / 1e6 }
        } catch (e) {
          error = String(e.message || e)
        }
      }
      if (!error && rank === 0) patch = JSON.stringify(materialize(ir))
      const ms = Number(process.hrtime.bigint() - t0) / 1e6
      process.send({ type: 'done', rank, error, patch, ms, times, fragmentBytes: rank === 0 && !error ? addon.shardFragmentBytes(ctx, world) : undefined })
    } else if (msg.type === 'close') {
      try { addon.shardFinalize(ctx) } catch (e) { /* the communicator goes with the process */ }
      addon.destroy(ctx)
      process.exit(0)
    }
  } catch (e) {
    process.send({ type: 'error', rank, error: String(e.message || e) })
  }
})
process.send({ type: 'up', rank })
