#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only). For every saved document a session of tests/golden/ref_apply_vectors.json.gz starts from:
the hashes of its changes as the unmodified reference rebuilds them (Backend.getAllChanges(Backend.load(doc)), computeHashGraph
new.js:1887-1912) -> tests/golden/ref_apply_vector_doc_hashes.json {pool index of the document: base64 of the 32-byte hashes}.
The oracle does not restate that reconstruction (oracle/am_oracle.h amo_set_document_history).

  python oracle/make_vector_doc_hashes.py
"""
import base64
import gzip
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = """
const path = require('path')
const { loadBackend } = require(path.join(process.argv[1], 'oracle', 'js', 'ref_loader'))
const { Backend, columnar } = loadBackend()
const docs = JSON.parse(require('fs').readFileSync(0, 'utf8'))
const out = {}
for (const [k, b64] of Object.entries(docs)) {
  const changes = Backend.getAllChanges(Backend.load(new Uint8Array(Buffer.from(b64, 'base64'))))
  out[k] = Buffer.concat(changes.map(c => Buffer.from(columnar.decodeChangeMeta(c, true).hash, 'hex'))).toString('base64')
}
console.log(JSON.stringify(out))
"""


def main():
    with open(os.path.join(ROOT, "tests", "golden", "ref_apply_vectors.json.gz"), "rb") as f:
        d = json.loads(gzip.decompress(f.read()))
    docs = {str(v["doc"]): d["pool"][v["doc"]] for v in d["vectors"] if "doc" in v}
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    env.pop("REF_BLOCK_SIZE", None)
    res = subprocess.run(["node", "-e", JS, ROOT], input=json.dumps(docs), capture_output=True, text=True, env=env, check=True)
    out = os.path.join(ROOT, "tests", "golden", "ref_apply_vector_doc_hashes.json")
    with open(out, "w") as f:
        json.dump({"made_by": "oracle/make_vector_doc_hashes.py on the unmodified reference", "doc_hashes": json.loads(res.stdout)}, f)
    print(f"{len(docs)} documents -> {out}")


if __name__ == "__main__":
    main()
