#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only). A delivery in which a LATER copy of a change is ready before its first copy: the
reference applies whichever copy becomes ready first (backend/new.js:1566 drops the other as a duplicate once that one is known), so
the application order -- visible in the key order of the patch's `clock` -- follows the later copy's position. The changes are those
of tests/golden/frontend_mixed_6actors.json (real frontend); the patch is the unmodified reference's (stock and block-size-patched
agree: the document is below one block).  -> tests/golden/frontend_mixed_6actors_dups_later_copy_first.json

  python oracle/make_dup_golden.py
"""
import base64
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd.loggen import ChangeLog  # noqa: E402


LOAD_JS = """
const fs = require('fs'); const { loadBackend } = require(process.argv[1]); const { Backend } = loadBackend()
process.stdout.write(JSON.stringify(Backend.getPatch(Backend.load(new Uint8Array(fs.readFileSync(process.argv[2]))))))
"""


def ref_patch(log, block_size=None):
    """(patch of loadChanges + getPatch, Backend.save bytes, patch of load(those bytes) + getPatch) of the reference."""
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    env.pop("REF_BLOCK_SIZE", None)
    if block_size:
        env["REF_BLOCK_SIZE"] = str(block_size)
    with tempfile.TemporaryDirectory() as tmp:
        path, out, doc = os.path.join(tmp, "log.bin"), os.path.join(tmp, "patch.json"), os.path.join(tmp, "doc.bin")
        log.save(path)
        subprocess.check_call(["node", os.path.join(ROOT, "oracle", "js", "ref_patch.js"), path, "--out", out, "--save", doc], env=env, stdout=subprocess.DEVNULL)
        loaded = subprocess.run(["node", "-e", LOAD_JS, os.path.join(ROOT, "oracle", "js", "ref_loader.js"), doc], env=env, capture_output=True, text=True, check=True).stdout
        return open(out).read(), open(doc, "rb").read(), loaded


def main():
    with open(os.path.join(ROOT, "tests", "golden", "frontend_mixed_6actors.json")) as f:
        base = json.load(f)
    changes = [base64.b64decode(c) for c in base["changes"]]
    n = len(changes)
    rng = np.random.default_rng(13)   # (a delivery on which applying only FIRST copies -- the engine of round 3 -- gives another clock order)
    perm = [int(i) for i in rng.permutation(n)]
    # copies behind everything (ready in the first pass although their first copies wait for later passes), one copy in front of
    # its dependencies (never the one applied), one change three times
    order = [perm[-1]] + perm + [perm[0], perm[0], perm[3], perm[n // 2], n - 1]
    log = ChangeLog.from_changes([changes[i] for i in order])
    (stock, doc, load_patch), (big, doc_big, load_big) = ref_patch(log), ref_patch(log, 100000000)
    assert stock == big and doc == doc_big and load_patch == load_big, "the block-boundary defect fired: pick a smaller document"
    in_order = json.loads(base["patch"])
    got = json.loads(stock)
    assert got["diffs"] == in_order["diffs"] and got["clock"] == in_order["clock"] and got["pendingChanges"] == 0, \
        "expected the same document with another application order"
    fx = {"name": "frontend_mixed_6actors_dups_later_copy_first",
          "note": "changes of frontend_mixed_6actors delivered shuffled with copies: later copies that are ready before their first copies (oracle/make_dup_golden.py)",
          "changes": [base64.b64encode(changes[i]).decode() for i in order], "patch": stock, "doc": base64.b64encode(doc).decode(), "load_patch": load_patch,
          "stock_equals_bigblock": True}
    path = os.path.join(ROOT, "tests", "golden", fx["name"] + ".json")
    with open(path, "w") as f:
        json.dump(fx, f)
    print(f"{len(order)} changes ({n} distinct) -> {path}")


if __name__ == "__main__":
    main()
