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
      const t0 = process.hrtime.bigint()
      let staged = true
      try {
        if (msg.kind === 'document') addon.loadDocument(ctx, new Uint8Array(fs.readFileSync(msg.file)))
        else addon.loadChanges(ctx, readChanges(msg.file))
      } catch (e) {
        staged = false   // (the collective is entered all the same: the library tells the other ranks that this one failed)
        addon.reset(ctx)
      }
      let error = null, patch
      try {
        addon.shardedReplay(ctx, false)
        if (!staged) error = 'staging failed'
        else if (rank === 0) patch = JSON.stringify(materialize(addon.fetchIR(ctx)))
      } catch (e) {
        error = String(e.message || e)
      }
      const ms = Number(process.hrtime.bigint() - t0) / 1e6
      process.send({ type: 'done', rank, error, patch, ms, fragmentBytes: rank === 0 && !error ? addon.shardFragmentBytes(ctx, world) : undefined })
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
