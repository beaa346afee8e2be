// node automerge_classic_amd/js/test_sharded.js <golden dir> [gpus]
// The sharded bulk replay through the JS host (js/sharded.js: one worker process per GPU, RCCL inside the library) against the
// golden patches of the unmodified reference (tests/golden) and against the unsharded engine of this process.
'use strict'
const fs = require('fs')
const path = require('path')
const { createPool } = require('./sharded.js')

const dir = process.argv[2], gpus = parseInt(process.argv[3] || '2')
const devices = process.env.AM355_SHARD_ALL_ON_DEVICE0 === '1' ? new Array(gpus).fill(0) : undefined   // (a one-GPU box, the emulation)

async function main() {
  const pool = await createPool({ gpus, devices })
  let n = 0, docs = 0
  try {
    for (const name of fs.readdirSync(dir).filter(f => f.endsWith('.json')).sort()) {
      const fx = JSON.parse(fs.readFileSync(path.join(dir, name)))
      if (!Array.isArray(fx.changes) || typeof fx.patch !== 'string') continue
      const changes = fx.changes.map(c => new Uint8Array(Buffer.from(c, 'base64')))
      const expected = fx.stock_equals_bigblock === false ? fx.patch_bigblock : fx.patch   // (the reference's patch, JSON text)
      const got = await pool.getPatchOfChanges(changes)
      if (JSON.stringify(got) !== expected) throw new Error(`${name}: sharded patch differs from the reference's`)
      n++
      if (fx.doc) {
        const wantDoc = fx.stock_equals_bigblock === false ? fx.load_patch_bigblock : fx.load_patch
        const gotDoc = await pool.getPatchOfDocument(new Uint8Array(Buffer.from(fx.doc, 'base64')))
        if (JSON.stringify(gotDoc) !== wantDoc) throw new Error(`${name}: sharded Backend.load patch differs from the reference's`)
        docs++
      }
    }
    // a batch one rank rejects is rejected as a whole, and the pool serves the next call
    const fx = JSON.parse(fs.readFileSync(path.join(dir, 'frontend_mixed_3actors.json')))
    const good = fx.changes.map(c => new Uint8Array(Buffer.from(c, 'base64')))
    const bad = good.map(c => Uint8Array.from(c))
    bad[1][20] ^= 0x55
    let threw = false
    try { await pool.getPatchOfChanges(bad) } catch (e) { threw = e.am355Sharded === true }
    if (!threw) throw new Error('a corrupt batch was not rejected')
    if (JSON.stringify(await pool.getPatchOfChanges(good)) !== fx.patch) throw new Error('the pool did not recover after a rejected batch')
  } finally {
    await pool.close()
  }
  console.log(JSON.stringify({ sharded_fixtures: n, sharded_documents: docs, gpus, fragmentBytes: pool.last && pool.last.fragmentBytes }))
  console.log('sharded fixtures reproduced')
}
main().catch(e => { console.error(e); process.exit(1) })
