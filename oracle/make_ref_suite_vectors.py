#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only). Captures (changes -> getPatch) vectors from the reference's own test suites run
against the unmodified reference backend (oracle/js/capture_ref_vectors.js) and stores them compactly: unique change blobs in
a pool, every vector a list of pool indexes plus the reference's patch -- tests/golden/ref_suite_vectors.json.gz.

  python oracle/make_ref_suite_vectors.py
"""
import gzip
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["new_backend_test.js", "backend_test.js", "test.js", "text_test.js", "table_test.js", "sync_test.js", "proxies_test.js", "frontend_test.js"]


def main():
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "v.jsonl")
        subprocess.check_call(["node", os.path.join(ROOT, "oracle", "js", "capture_ref_vectors.js"), raw] + SUITES, env=env)
        pool, plist, vecs = {}, [], []
        with open(raw) as f:
            for line in f:
                d = json.loads(line)
                idx = []
                for c in d["changes"]:
                    if c not in pool:
                        pool[c] = len(plist)
                        plist.append(c)
                    idx.append(pool[c])
                v = {"kind": d["kind"], "changes": idx}
                for k in ("patch", "error", "doc_len", "doc_sha256"):
                    if k in d:
                        v[k] = d[k]
                vecs.append(v)
    blob = json.dumps({"made_by": "oracle/make_ref_suite_vectors.py: reference suites " + ", ".join(SUITES) + " on the unmodified reference backend",
                       "pool": plist, "vectors": vecs}).encode()
    out = os.path.join(ROOT, "tests", "golden", "ref_suite_vectors.json.gz")
    with open(out, "wb") as f:
        f.write(gzip.compress(blob, 9, mtime=0))
    print(f"{len(vecs)} vectors, {len(plist)} distinct blobs -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
