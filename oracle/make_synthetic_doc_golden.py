#!/usr/bin/env python3
"""TEST INFRASTRUCTURE. Generates tests/golden/synthetic_doc_*.json: synthetic saved documents (loggen docgen) with the
patch the UNMODIFIED reference returns for Backend.getPatch(Backend.load(bytes)) (run under node in the build container).

  python oracle/make_synthetic_doc_golden.py
"""
import base64
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import loggen  # noqa: E402

NODE_SNIPPET = """
const fs = require('fs')
const { loadBackend } = require(process.argv[1])
const { Backend } = loadBackend()
const state = Backend.load(new Uint8Array(fs.readFileSync(process.argv[2])))
process.stdout.write(JSON.stringify(Backend.getPatch(state)))
"""

CASES = {
    "synthetic_doc_small": dict(n_actors=5, n_texts=2, text_len=120, n_maps=2, keys_per_map=30, n_submaps=2, n_lists=2, list_len=60, deflate=True, seed=11),
    "synthetic_doc_medium": dict(n_actors=24, n_texts=6, text_len=900, n_maps=5, keys_per_map=300, n_submaps=3, n_lists=4, list_len=500, deflate=True, seed=12),
    "synthetic_doc_nodeflate": dict(n_actors=3, n_texts=1, text_len=700, n_maps=1, keys_per_map=90, n_submaps=1, n_lists=1, list_len=100, deflate=False, seed=13),
}


def ref_patch(doc, block_size=None):
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    if block_size:
        env["REF_BLOCK_SIZE"] = str(block_size)
    with tempfile.NamedTemporaryFile(suffix=".doc") as f:
        f.write(doc)
        f.flush()
        return subprocess.check_output(["node", "-e", NODE_SNIPPET, os.path.join(ROOT, "oracle", "js", "ref_loader.js"), f.name], env=env).decode()


def main():
    for name, kw in CASES.items():
        doc, rows = loggen.generate_document(**kw)
        stock = ref_patch(doc)
        big = ref_patch(doc, 100000000)
        fx = {"name": name, "note": f"synthetic document ({rows} op rows) from loggen/docgen.cpp; patch from the unmodified reference's Backend.load + getPatch",
              "doc": base64.b64encode(doc).decode(), "load_patch": stock, "stock_equals_bigblock": stock == big, "rows": rows}
        if stock != big:
            fx["load_patch_bigblock"] = big
        with open(os.path.join(ROOT, "tests", "golden", name + ".json"), "w") as f:
            json.dump(fx, f)
        print(name, rows, "rows,", len(doc), "bytes, patch", len(stock), "stock==bigblock:", stock == big)


if __name__ == "__main__":
    main()
