#!/usr/bin/env python3
"""Records the UNMODIFIED reference JS backend's rate on the bench workloads in the BUILD container (node + /root/reference exist here,
not on the GPU box) into profiles/r06_reference_js_baseline.json -- dated, with the box and the node version. bench.py embeds the file
as cpu_baseline.reference_js_recorded next to the live C-port leg it can run on the GPU box (VERDICT r4 missing #4: the reference tree
cannot travel; a committed measurement made by the bench's own reference_js_baseline leg can).

  python tools/record_reference_js.py
"""
import datetime
import json
import os
import platform
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from automerge_classic_amd import loggen  # noqa: E402

NODE_LOAD = r"""
const fs = require('fs')
const { loadBackend } = require(process.argv[1])
const { Backend } = loadBackend()
const bytes = new Uint8Array(fs.readFileSync(process.argv[2]))
const times = []
for (let i = 0; i < 3; i++) {
  const t0 = process.hrtime.bigint()
  const state = Backend.load(bytes)
  const patch = Backend.getPatch(state)
  times.push(Number(process.hrtime.bigint() - t0) / 1e9)
  if (!patch.diffs) throw new Error('no patch')
}
times.sort((a, b) => a - b)
process.stdout.write(JSON.stringify({median_s: times[1]}))
"""


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    out = {"made_by": "tools/record_reference_js.py (bench.reference_js_baseline: oracle/js/ref_patch.js --time 3 on the unmodified reference)",
           "date": datetime.date.today().isoformat(), "host": {"cpu": cpu_model(), "cores_visible": os.cpu_count(), "note": "build container, 1 core used"},
           "node": subprocess.run(["node", "--version"], capture_output=True, text=True).stdout.strip(), "reference": "automerge-classic v1.0.1-preview.7 (/root/reference)",
           "workloads": {}}
    for name in ("c4_text_single", "c4_text_multi", "c3_map_lww", "c2_text_typing"):
        t0 = time.time()
        # (the reference replays the map workload at ~1.5 k ops/s: a fifth of it is 16 k ops, ~10 s per run)
        r = bench.reference_js_baseline(name, 0.2 if name == "c3_map_lww" else 1.0, bench.BASE_SEED[name], timeout_s=1200)
        if r is None:
            raise SystemExit(f"{name}: node or the reference tree is missing, or the run exceeded its limit")
        out["workloads"][name] = {"ops_per_s": r["value"], "cores": 1, "sample": r["sample"]}
        print(name, "%.0f ops/s" % r["value"], "(%.0f s)" % (time.time() - t0), flush=True)
    # Backend.load + getPatch of a config-5 shaped document at 2 % of the rows (the reference loads ~10^5 rows/s)
    doc, rows = loggen.document_config(0.02)
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    with tempfile.NamedTemporaryFile(suffix=".doc") as f:
        f.write(doc)
        f.flush()
        r = json.loads(subprocess.check_output(["node", "-e", NODE_LOAD, os.path.join(ROOT, "oracle", "js", "ref_loader.js"), f.name], env=env, timeout=900).decode())
    out["workloads"]["c5_doc_mixed"] = {"ops_per_s": rows / r["median_s"], "cores": 1,
                                        "sample": f"c5_doc_mixed x0.02: Backend.load + getPatch of a {len(doc)}-byte saved document, {rows} op rows, median of 3"}
    print("c5_doc_mixed", "%.0f rows/s" % (rows / r["median_s"]), flush=True)
    path = os.path.join(ROOT, "profiles", "r06_reference_js_baseline.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
