#!/usr/bin/env python3
"""TEST INFRASTRUCTURE. Generates tests/golden/doc_history_longkey.json with the UNMODIFIED reference (node, /root/reference):
documents in which ONE long map key is overwritten by many changes. The saved document holds such a key once (RLE), the changes
rebuilt by Backend.getAllChanges(Backend.load(doc)) hold it once per change -- the rebuilt key columns are many times longer than
the document's own (the device encoder's output was sized from the latter).

  python oracle/make_longkey_history_golden.py

Per case: the document (base64), Backend.getPatch after load, number / bytes / digests of the rebuilt changes (same digests as
oracle/make_history_golden.py).
"""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NODE_SNIPPET = r"""
const crypto = require('crypto')
const { loadBackend } = require(process.argv[1])
const { Backend, columnar, Automerge: am } = loadBackend()
const Automerge = am()
const cases = JSON.parse(process.argv[2])
const out = {}
for (const cs of cases) {
  let doc = Automerge.init({actorId: cs.actor})
  for (let i = 0; i < cs.n_changes; i++) {
    doc = Automerge.change(doc, {time: 0}, d => {
      for (let k = 0; k < cs.keys.length; k++) if (i % (k + 1) === 0) d[cs.keys[k]] = i * 7 + k
    })
  }
  const bytes = Automerge.save(doc)
  const state = Backend.load(bytes)
  const patch = Backend.getPatch(state)
  const changes = Backend.getAllChanges(state)
  const all = crypto.createHash('sha256'), hs = crypto.createHash('sha256')
  let total = 0
  for (const c of changes) {
    const len = Buffer.alloc(4); len.writeUInt32LE(c.byteLength)
    all.update(len); all.update(c)
    hs.update(Buffer.from(columnar.decodeChangeMeta(c, true).hash, 'hex'))
    total += c.byteLength
  }
  out[cs.name] = {doc: Buffer.from(bytes).toString('base64'), patch, n_changes: changes.length, bytes: total,
                  changes_sha256: all.digest('hex'), hashes_sha256: hs.digest('hex')}
}
process.stdout.write(JSON.stringify(out))
"""

CASES = [
    {"name": "one_key_48_chars_400_changes", "actor": "aabbccdd00112233aabbccdd00112233", "n_changes": 400, "keys": ["k" * 48]},
    {"name": "three_keys_up_to_200_chars_150_changes", "actor": "0123456789abcdef0123456789abcdef", "n_changes": 150,
     "keys": ["a-rather-long-property-name/" * 7, "é" * 60, "short"]},
]


def main():
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    res = json.loads(subprocess.check_output(["node", "-e", NODE_SNIPPET, os.path.join(ROOT, "oracle", "js", "ref_loader.js"), json.dumps(CASES)], env=env).decode())
    out = {"note": "Backend.getAllChanges(Backend.load(doc)) of the unmodified reference on documents with long, often-overwritten map keys; made by oracle/make_longkey_history_golden.py",
           "cases": res}
    path = os.path.join(ROOT, "tests", "golden", "doc_history_longkey.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"), sort_keys=True)
    for name, r in res.items():
        print(name, "doc bytes", len(r["doc"]) * 3 // 4, "changes", r["n_changes"], "bytes", r["bytes"])


if __name__ == "__main__":
    main()
