#!/usr/bin/env python3
"""TEST INFRASTRUCTURE. Demonstrates, in-tree, the stock reference's block-boundary defect and pins the
engine/oracle answer on inputs where it fires.

The stock reference (backend/new.js, MAX_BLOCK_SIZE = 600) mis-places a concurrent list insertion when the skip scan over
greater-id siblings (new.js:144-163) reaches the end of a 600-op block: seekToOp re-enters seekWithinBlock with
resumeInsertion=true (new.js:303-306), nextObjCtr/nextObjActor are never loaded there (new.js:112-118 runs only when
!resumeInsertion), the loop at :144 exits at once, and the element lands at the start of the next block. Where blocks split
depends on delivery order, so two replicas that received the same changes in two causally valid orders hold DIFFERENT
documents. With the one constant MAX_BLOCK_SIZE raised above the document size (oracle/js/ref_loader.js, in memory) the
defect cannot fire and the reference follows its documented rule -- which is what the oracle and the engine implement.

Writes tests/golden/defect_block_boundary.json:
  changes                    binary changes (base64), generated order: loggen KIND_TEXT_CONCURRENT, 4 actors x 1 round x 200
                             chained inserts, seed 0x5EED0004 (801 ops, 5 changes)
  order_reversed             the same changes with the four concurrent round-1 changes delivered in reverse order
  patch / patch_reversed     STOCK reference patches for the two delivery orders (they differ from each other)
  patch_bigblock / patch_bigblock_reversed   block-size-patched reference (REF_BLOCK_SIZE=1e8): identical diffs
  doc / load_patch_bigblock  Backend.save of the block-size-patched reference and its load patch
  stock_equals_bigblock      false
  larger                     digests only: c4_text_single x0.02 (12,801 ops, 65 changes) in both orders, stock vs big-block

  python oracle/make_defect_fixture.py
"""
import base64
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import loggen  # noqa: E402


def ref(log, big, save=False):
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    if big:
        env["REF_BLOCK_SIZE"] = "100000000"
    with tempfile.TemporaryDirectory() as tmp:
        lp, out, dp = os.path.join(tmp, "l.bin"), os.path.join(tmp, "p.json"), os.path.join(tmp, "d.bin")
        log.save(lp)
        cmd = ["node", os.path.join(ROOT, "oracle", "js", "ref_patch.js"), lp, "--out", out]
        if save:
            cmd += ["--save", dp]
        subprocess.check_call(cmd, env=env, stderr=subprocess.DEVNULL)
        patch = open(out).read()
        return (patch, open(dp, "rb").read()) if save else patch


def ref_load(doc, big):
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    if big:
        env["REF_BLOCK_SIZE"] = "100000000"
    js = ("const {loadBackend}=require(process.argv[1]);const {Backend}=loadBackend();const fs=require('fs');"
          "process.stdout.write(JSON.stringify(Backend.getPatch(Backend.load(new Uint8Array(fs.readFileSync(process.argv[2]))))))")
    with tempfile.TemporaryDirectory() as tmp:
        dp = os.path.join(tmp, "d.bin")
        open(dp, "wb").write(doc)
        return subprocess.check_output(["node", "-e", js, os.path.join(ROOT, "oracle", "js", "ref_loader.js"), dp], env=env).decode()


def reversed_round1(log, n_actors):
    """Change 0 creates the Text object; changes 1..n_actors are round 1 (mutually concurrent): deliver those in reverse."""
    order = [0] + list(range(n_actors, 0, -1)) + list(range(n_actors + 1, log.n_changes))
    return order, log.reordered(order)


def sha(s):
    return hashlib.sha256(s.encode() if isinstance(s, str) else s).hexdigest()


def main():
    na = 4
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=na, n_rounds=1, ins_per_change=200, del_per_change=0, n_objects=1, seed=0x5EED0004)
    order, rlog = reversed_round1(log, na)
    stock, stock_r = ref(log, False), ref(rlog, False)
    (big, doc), big_r = ref(log, True, save=True), ref(rlog, True)
    d = lambda p: json.dumps(json.loads(p)["diffs"])  # noqa: E731
    assert d(stock) != d(big), "the defect did not fire on the small case"
    assert d(stock) != d(stock_r), "stock reference: both delivery orders gave the same document"
    assert d(big) == d(big_r), "block-size-patched reference: delivery order changed the document"
    load_big = ref_load(doc, True)
    big_log = loggen.config("c4_text_single", 0.02)
    _, big_rlog = reversed_round1(big_log, 64)
    L = {"stock": ref(big_log, False), "stock_reversed": ref(big_rlog, False), "bigblock": ref(big_log, True), "bigblock_reversed": ref(big_rlog, True)}
    assert d(L["stock"]) != d(L["bigblock"]) and d(L["stock"]) != d(L["stock_reversed"]) and d(L["bigblock"]) == d(L["bigblock_reversed"])
    fx = {
        "name": "defect_block_boundary",
        "note": "stock reference mis-places concurrent inserts across its 600-op block boundary (new.js:303-306, 112-118, 144): the two STOCK patches "
                "below differ from each other although they hold the same changes; expected = block-size-patched reference (oracle/make_defect_fixture.py)",
        "changes": [base64.b64encode(c).decode() for c in log.changes()],
        "order_reversed": order,
        "patch": stock, "patch_reversed": stock_r, "patch_bigblock": big, "patch_bigblock_reversed": big_r,
        "doc": base64.b64encode(doc).decode(), "load_patch": load_big, "load_patch_bigblock": load_big,
        "stock_equals_bigblock": False,
        "larger": {"workload": "c4_text_single", "scale": 0.02, "n_ops": int(big_log.n_ops), "n_changes": int(big_log.n_changes), "reversed": "changes 1..64 reversed",
                   "diffs_sha256": {k: sha(d(v)) for k, v in L.items()}, "patch_sha256": {k: sha(v) for k, v in L.items()}},
    }
    with open(os.path.join(ROOT, "tests", "golden", "defect_block_boundary.json"), "w") as f:
        json.dump(fx, f)
    print("written: stock != bigblock, stock != stock_reversed, bigblock == bigblock_reversed (diffs) on", log.n_ops, "and", big_log.n_ops, "ops")


if __name__ == "__main__":
    main()
