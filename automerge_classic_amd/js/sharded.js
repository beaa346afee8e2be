// Sharded bulk replay from the JS host: ONE document over the GPUs of a node, one worker process per GPU (js/shard_worker.js), the
// collective inside the library (RCCL over xGMI: am355_shard_init / am355_sharded_replay, include/am355.h). north_star: "host code
// stays JavaScript ... Changes shard by objectId across the 8 GPUs of one node with an RCCL ... over xGMI".
//
//   const { createPool } = require('mi355x-backend/sharded')
//   const pool = await createPool({ gpus: 8 })
//   const patch = await pool.getPatchOfChanges(changes)      // = Backend.getPatch(Backend.loadChanges(Backend.init(), changes))
//   const patch2 = await pool.getPatchOfDocument(bytes)      // = Backend.getPatch(Backend.load(bytes))
//   await pool.close()
//
// The reference's Backend calls are synchronous (columnar.js:8-12); a pool of processes cannot be, so this is an API of its own
// beside the drop-in module (index.js), for the bulk calls that are worth more than one GPU: documents of many objects
// (c4_text_multi, config 5). A document dominated by ONE list/Text object does not shard (SURVEY.md §8e): use index.js.
// Every rank stages the whole batch (the batch goes through a file: node 12's IPC is JSON) and merges the objects it owns; rank 0
// stitches the fragments and materialises the patch.
'use strict'
const { fork } = require('child_process')
const fs = require('fs')
const os = require('os')
const path = require('path')

function writeLog(file, changes) {
  const n = changes.length
  const head = Buffer.alloc(12 + 8 * (n + 1))
  head.writeUInt32LE(n, 0)
  head.writeBigUInt64LE(0n, 4)
  let off = 0n
  for (let i = 0; i < n; i++) {
    head.writeBigUInt64LE(off, 12 + 8 * i)
    off += BigInt(changes[i].length)
  }
  head.writeBigUInt64LE(off, 12 + 8 * n)
  const fd = fs.openSync(file, 'w')
  fs.writeSync(fd, head)
  for (const c of changes) fs.writeSync(fd, c)
  fs.closeSync(fd)
}

class Pool {
  constructor(workers, dir) { this.workers = workers; this.dir = dir; this.seq = 0; this.busy = false }

  // one message to every worker (or to `only`), one answer of `type` from each
  _round(msg, type, only) {
    const targets = only === undefined ? this.workers : [this.workers[only]]
    return Promise.all(targets.map(w => new Promise((resolve, reject) => {
      const onMsg = m => {
        if (m.type === type) { cleanup(); resolve(m) } else if (m.type === 'error') { cleanup(); reject(new Error(`shard worker ${m.rank}: ${m.error}`)) }
      }
      const onExit = code => { cleanup(); reject(new Error(`shard worker exited with code ${code}`)) }
      const cleanup = () => { w.removeListener('message', onMsg); w.removeListener('exit', onExit) }
      w.on('message', onMsg)
      w.on('exit', onExit)
      w.send(msg)
    })))
  }

  async _replay(kind, write, reps) {
    if (this.busy) throw new Error('one sharded replay at a time per pool')
    this.busy = true
    const file = path.join(this.dir, `batch${this.seq++}.bin`)
    try {
      write(file)
      const done = await this._round({ type: 'replay', file, kind, reps }, 'done')
      const failed = done.filter(d => d.error)
      if (failed.length) {
        const e = new RangeError(`sharded replay rejected: ${failed.map(d => `rank ${d.rank}: ${d.error}`).join('; ')}`)
        e.am355Sharded = true   // (the caller replays the batch on the JS path for the reference's exact exception, as index.js does)
        throw e
      }
      const r0 = done.find(d => d.rank === 0)
      this.last = { ms: done.map(d => d.ms), times: done.map(d => d.times), fragmentBytes: r0.fragmentBytes }
      return JSON.parse(r0.patch)
    } finally {
      this.busy = false
      try { fs.unlinkSync(file) } catch (e) { /* already gone */ }
    }
  }

  getPatchOfChanges(changes, reps) { return this._replay('changes', file => writeLog(file, changes), reps) }
  getPatchOfDocument(bytes, reps) { return this._replay('document', file => fs.writeFileSync(file, bytes), reps) }

  async close() {
    for (const w of this.workers) { try { w.send({ type: 'close' }) } catch (e) { /* gone */ } }
    await Promise.all(this.workers.map(w => new Promise(resolve => { if (w.exitCode !== null) resolve(); else w.on('exit', resolve) })))
    try { fs.rmdirSync(this.dir) } catch (e) { /* not empty: a batch file of a failed call */ }
  }
}

// gpus: number of worker processes = GPUs; devices: the HIP device of each rank (default: rank r on device r)
async function createPool({ gpus, devices } = {}) {
  if (!(gpus >= 1)) throw new TypeError('createPool({gpus: N})')
  const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'am355_shard_'))
  const workers = []
  const up = []
  for (let r = 0; r < gpus; r++) {
    const env = Object.assign({}, process.env, { AM355_SHARD_RANK: String(r), AM355_SHARD_WORLD: String(gpus) })
    if (devices) env.AM355_SHARD_DEVICE = String(devices[r])
    const w = fork(path.join(__dirname, 'shard_worker.js'), [], { env })
    up.push(new Promise((resolve, reject) => {
      w.once('message', m => (m.type === 'up' ? resolve() : reject(new Error('shard worker did not start'))))
      w.once('exit', code => reject(new Error(`shard worker ${r} exited with code ${code} while starting (no MI355X on device ${devices ? devices[r] : r}?)`)))
    }))
    workers.push(w)
  }
  const pool = new Pool(workers, dir)
  try {
    await Promise.all(up)
    const [{ id }] = await pool._round({ type: 'id' }, 'id', 0)
    await pool._round({ type: 'init', id }, 'ready')   // (ncclCommInitRank blocks until every rank has called it: all at once)
  } catch (e) {
    for (const w of workers) w.kill()
    throw e
  }
  return pool
}

module.exports = { createPool }
