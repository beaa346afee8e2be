// ONE document sharded by objectId over N GPUs from the JS host: js/sharded.js (one worker process per GPU, the collective inside
// the library: am355_sharded_replay = replay + ncclAllGather of the patch-IR fragments over xGMI + stitch on rank 0).
//   node automerge_classic_amd/js/bench_sharded.js <log.bin> <gpus> [reps]
// Prints one JSON line: the engine part of the last of `reps` repetitions per rank (stage = am355_load_changes, replay =
// am355_sharded_replay, fetch = record tables to the host on rank 0), the step time = max over ranks of stage + replay, + rank 0's
// fetch, the same batch unsharded on one worker (gpus = 1 pool... measured by the caller with gpus 1), and the sha256 of the patch text.
// Log file layout (little endian): u32 n_changes, u64 n_ops, u64 offsets[n+1], arena.
'use strict'
const crypto = require('crypto')
const fs = require('fs')
const { createPool } = require('./sharded.js')

const buf = fs.readFileSync(process.argv[2])
const gpus = parseInt(process.argv[3] || '1'), reps = parseInt(process.argv[4] || '5')
const n = buf.readUInt32LE(0), nOps = Number(buf.readBigUInt64LE(4))
const base = 12 + 8 * (n + 1)
const changes = []
for (let i = 0; i < n; i++) {
  const a = Number(buf.readBigUInt64LE(12 + 8 * i)), b = Number(buf.readBigUInt64LE(12 + 8 * (i + 1)))
  changes.push(new Uint8Array(buf.buffer, buf.byteOffset + base + a, b - a))
}
const devices = process.env.AM355_SHARD_ALL_ON_DEVICE0 === '1' ? new Array(gpus).fill(0) : undefined

async function main() {
  const pool = await createPool({ gpus, devices })
  try {
    const patch = await pool.getPatchOfChanges(changes, reps)
    const t = pool.last.times
    const step = Math.max(...t.map(x => x.stage + x.replay)) + t[0].fetch
    console.log(JSON.stringify({
      gpus, reps, n_ops: nOps, n_changes: n, engine_ms_per_step: step, ops_per_s: nOps / (step / 1e3), per_rank_ms: t, fragment_bytes: pool.last.fragmentBytes,
      patch_sha256: crypto.createHash('sha256').update(JSON.stringify(patch)).digest('hex')
    }))
  } finally {
    await pool.close()
  }
}
main().catch(e => { console.error(e); process.exit(1) })
