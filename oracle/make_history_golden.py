#!/usr/bin/env python3
"""TEST INFRASTRUCTURE. Generates tests/golden/doc_history.json: what the UNMODIFIED reference returns for
Backend.getAllChanges(Backend.load(doc)) (history reconstruction, new.js:1887-1912 computeHashGraph) on every committed document
fixture and on generated documents (loggen docgen; only their parameters and digests are stored).

  python oracle/make_history_golden.py

Per document: number of changes, SHA-256 over the changes (each prefixed by its length as 4 little-endian bytes), SHA-256 over the
concatenated change hashes, the document heads -- or the error text when the reference throws.
"""
import base64
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import loggen  # noqa: E402

NODE_SNIPPET = """
const fs = require('fs'), crypto = require('crypto')
const { loadBackend } = require(process.argv[1])
const { Backend, columnar } = loadBackend()
let out
try {
  const state = Backend.load(new Uint8Array(fs.readFileSync(process.argv[2])))
  const changes = Backend.getAllChanges(state)
  const all = crypto.createHash('sha256'), hs = crypto.createHash('sha256')
  let bytes = 0
  for (const c of changes) {
    const len = Buffer.alloc(4); len.writeUInt32LE(c.byteLength)
    all.update(len); all.update(c)
    hs.update(Buffer.from(columnar.decodeChangeMeta(c, true).hash, 'hex'))
    bytes += c.byteLength
  }
  out = {n_changes: changes.length, bytes, changes_sha256: all.digest('hex'), hashes_sha256: hs.digest('hex'), heads: Backend.getHeads(state)}
} catch (e) {
  out = {error: String(e.message)}
}
process.stdout.write(JSON.stringify(out))
"""

# generated logs (loggen.config): the document is the block-size-patched reference's save() of the replayed log (see
# oracle/make_save_golden.py for why patched), its history then comes from the reference's load + getAllChanges
GENERATED = [("c2_text_typing", 0.03, True), ("c3_map_lww", 0.13, True), ("c4_text_multi", 0.05, True), ("c4_text_multi", 0.02, False)]


def ref_history(doc):
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    with tempfile.NamedTemporaryFile(suffix=".doc") as f:
        f.write(doc)
        f.flush()
        return json.loads(subprocess.check_output(["node", "-e", NODE_SNIPPET, os.path.join(ROOT, "oracle", "js", "ref_loader.js"), f.name], env=env).decode())


def main():
    out = {"note": "Backend.getAllChanges(Backend.load(doc)) of the unmodified reference; made by oracle/make_history_golden.py", "fixtures": {}}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.json"))):
        name = os.path.basename(path)[:-5]
        if name in ("doc_history", "save_generated"):
            continue
        with open(path) as f:
            fx = json.load(f)
        if not isinstance(fx, dict) or "doc" not in fx:
            continue
        r = ref_history(base64.b64decode(fx["doc"]))
        out["fixtures"][name] = r
        print(name, r.get("n_changes"), r.get("error", ""))
    out["generated"] = []
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"), REF_BLOCK_SIZE="100000000")
    with tempfile.TemporaryDirectory() as tmp:
        for wl, scale, deflate in GENERATED:
            log = loggen.config(wl, scale, deflate)
            lp, dp = os.path.join(tmp, "l.bin"), os.path.join(tmp, "d.bin")
            log.save(lp)
            subprocess.check_call(["node", os.path.join(ROOT, "oracle", "js", "ref_patch.js"), lp, "--save", dp, "--out", os.path.join(tmp, "p.json")],
                                  env=env, stderr=subprocess.DEVNULL)
            r = ref_history(open(dp, "rb").read())
            r.update({"workload": wl, "scale": scale, "deflate": deflate, "n_ops": int(log.n_ops)})
            out["generated"].append(r)
            print(wl, scale, deflate, r.get("n_changes"), r.get("bytes"), r.get("error", ""))
    with open(os.path.join(ROOT, "tests", "golden", "doc_history.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
