// TEST INFRASTRUCTURE (build container only). Bloom filters of the unmodified reference (backend/sync.js:38-128 `BloomFilter`,
// exported "for testing purposes") over given sets of change hashes: stdin = {"hashes": [hex...], "sets": [[index...]...]},
// stdout = {"filters": [hex of new BloomFilter(set).bytes ...], "contains": [[0|1 per hash of `hashes` ...] per set]}.
const path = require('path')
const { REF } = require('./ref_loader.js')
const { BloomFilter } = require(path.join(REF, 'backend', 'sync.js'))
let text = ''
process.stdin.on('data', d => { text += d })
process.stdin.on('end', () => {
  const { hashes, sets } = JSON.parse(text)
  const filters = [], contains = []
  for (const set of sets) {
    const bytes = new BloomFilter(set.map(i => hashes[i])).bytes
    filters.push(Buffer.from(bytes).toString('hex'))
    const parsed = new BloomFilter(bytes)   // (as a receiving peer reads it)
    contains.push(hashes.map(h => (parsed.containsHash(h) ? 1 : 0)))
  }
  process.stdout.write(JSON.stringify({ filters, contains }))
})
