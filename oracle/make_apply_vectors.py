#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only). Captures the INCREMENTAL patches of Backend.applyChanges from the
reference's own test suites run against the unmodified reference backend (oracle/js/capture_apply_vectors.js) and stores them
compactly in tests/golden/ref_apply_vectors.json.gz:

  pool     distinct binary blobs (changes, saved documents), base64
  vectors  one per recorded applyChanges call: {"parent": index of the vector whose session this call continues (or -1: the call
           was made on an empty document / on `doc`), "doc": pool index of the saved document the session started from (optional),
           "local": the call came from applyLocalChange (the patch then carries actor + seq), "changes": pool indexes of the batch,
           "patch": JSON.stringify of the patch the reference returned | "error": first line of the exception message}
           Parents always precede their children.

  python oracle/make_apply_vectors.py
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
    env.pop("REF_BLOCK_SIZE", None)
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "v.jsonl")
        subprocess.check_call(["node", os.path.join(ROOT, "oracle", "js", "capture_apply_vectors.js"), raw] + SUITES, env=env)
        pool, plist = {}, []

        def intern(b64):
            if b64 not in pool:
                pool[b64] = len(plist)
                plist.append(b64)
            return pool[b64]

        sessions = []
        with open(raw) as f:
            for line in f:
                d = json.loads(line)
                calls = tuple((bool(c["local"]), tuple(intern(x) for x in c["changes"])) for c in d["calls"])
                doc = intern(d["doc"]) if "doc" in d else -1
                sessions.append(((doc, calls), d))
    # a session that failed leaves the document unchanged, so it is never a parent; parents = successful sessions one call shorter
    sessions.sort(key=lambda s: (len(s[0][1]), s[0]))
    index, vecs = {}, []
    for key, d in sessions:
        doc, calls = key
        parent = index.get((doc, calls[:-1]), -1) if len(calls) > 1 else -1
        if len(calls) > 1 and parent < 0:
            raise SystemExit("session without a recorded parent: capture every call of a lineage")
        v = {"parent": parent, "local": calls[-1][0], "changes": list(calls[-1][1])}
        if doc >= 0:
            v["doc"] = doc
        if "patch" in d:
            v["patch"] = d["patch"]
            index[key] = len(vecs)
        else:
            v["error"] = d["error"]
        vecs.append(v)
    blob = json.dumps({"made_by": "oracle/make_apply_vectors.py: every applyChanges call of the reference suites " + ", ".join(SUITES) +
                                  " on the unmodified reference backend, with the patch the call returned",
                       "pool": plist, "vectors": vecs}).encode()
    out = os.path.join(ROOT, "tests", "golden", "ref_apply_vectors.json.gz")
    with open(out, "wb") as f:
        f.write(gzip.compress(blob, 9, mtime=0))
    print(f"{len(vecs)} vectors, {len(plist)} distinct blobs -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
