// TEST INFRASTRUCTURE (not product code). Loads the reference backend from /root/reference.
//
// REF_BLOCK_SIZE=<n> (optional) loads backend/new.js with its MAX_BLOCK_SIZE constant (new.js:6) replaced
// in memory by <n>; nothing else is changed and nothing is written to disk. This exists because the
// reference has a defect in list insertion across op-block boundaries: when the scan over concurrently
// inserted elements with greater opIds (new.js:144-163) reaches the end of a block, seekToOp re-enters
// seekWithinBlock with resumeInsertion=true (new.js:303-306), where nextObjCtr/nextObjActor are never
// loaded (new.js:112-118 only runs when !resumeInsertion), so the skip loop's object test fails at once
// and the element is placed at the start of the next block regardless of the ids that follow. The result
// then depends on where 600-op blocks happened to split, i.e. on delivery order (replicas diverge).
// With a block size larger than the document the defect cannot fire, giving the algorithm as documented.
const path = require('path')
const fs = require('fs')
const Module = require('module')
const REF = process.env.AUTOMERGE_REF || '/root/reference'

function loadBackend() {
  const blockSize = process.env.REF_BLOCK_SIZE
  if (blockSize) {
    const file = path.join(REF, 'backend', 'new.js')
    let src = fs.readFileSync(file, 'utf8')
    const needle = 'const MAX_BLOCK_SIZE = 600'
    if (!src.includes(needle)) throw new Error('MAX_BLOCK_SIZE constant not found in reference new.js')
    src = src.replace(needle, `const MAX_BLOCK_SIZE = ${parseInt(blockSize)}`)
    const m = new Module(file, module)
    m.filename = file
    m.paths = Module._nodeModulePaths(path.dirname(file))
    m._compile(src, file)
    require.cache[file] = m
    m.loaded = true
  }
  return {
    Backend: require(path.join(REF, 'backend')),
    columnar: require(path.join(REF, 'backend', 'columnar')),
    Automerge: () => require(path.join(REF, 'src', 'automerge'))
  }
}
module.exports = { loadBackend, REF }
