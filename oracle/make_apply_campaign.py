#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only). Sessions of Backend.applyChanges calls on random multi-actor documents made with the
real frontend (oracle/js/apply_campaign.js), every call with the patch the unmodified reference returned -- the incremental
patches of SURVEY.md 8f-2 on documents larger and more concurrent than the reference's own test suites hold.
-> tests/golden/apply_campaign.json.gz: {"pool": [base64 change...], "sessions": [{"name", "calls": [[pool index...]...],
"patches": [JSON text | {"error": message}], "doc": base64 saved document the session starts from, "doc_hashes": base64 of the 32-byte hashes of its changes (loaded sessions only)}]}

  python oracle/make_apply_campaign.py
"""
import gzip
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPECS = ["m:11:3:120:2", "m:12:4:160:3", "m:15:2:100:3", "m:16:6:140:2", "t:13:4:4:16", "t:14:6:3:24", "t:17:3:8:10", "21:3:70:2"]
# lists whose elements are assigned to (apply_campaign.js listScenario) -> tests/golden/apply_campaign_lists.json.gz
LIST_SPECS = ["l:31:2:60:0", "l:32:3:90:0", "l:33:4:120:0", "l:34:3:100:10", "l:35:5:150:0", "l:36:2:80:25"]
# wide conflicts: every actor assigns the same list elements in one change (apply_campaign.js conflictScenario): more edit records than
# op rows per call -> tests/golden/apply_campaign_conflicts.json.gz
CONFLICT_SPECS = ["c:41:2:60:40", "c:42:3:30:10", "c:43:3:200:0", "c:44:4:64:5", "c:45:2:300:100", "c:47:5:40:3", "c:48:2:5:0"]


# sessions onto LOADED documents (LOADED=1: the first calls are saved and loaded again) -> tests/golden/apply_campaign_loaded.json.gz
LOADED_SPECS = ["m:51:3:120:2", "m:52:4:160:3", "m:55:2:100:3", "m:56:6:140:2", "t:53:4:4:16", "l:54:3:90:0", "l:57:4:120:10", "61:3:70:2"]


def main(specs=SPECS, name="apply_campaign.json.gz", loaded=False):
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    env.pop("REF_BLOCK_SIZE", None)
    env.pop("LOADED", None)
    if loaded:
        env["LOADED"] = "1"
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "c.jsonl")
        subprocess.check_call(["node", os.path.join(ROOT, "oracle", "js", "apply_campaign.js"), raw] + specs, env=env)
        pool, plist, sessions = {}, [], []
        for line in open(raw):
            d = json.loads(line)
            calls = []
            for call in d["calls"]:
                idx = []
                for c in call:
                    if c not in pool:
                        pool[c] = len(plist)
                        plist.append(c)
                    idx.append(pool[c])
                calls.append(idx)
            sessions.append({"name": d["name"], "calls": calls, "patches": d["patches"]})
            for k in ("doc", "graph", "doc_hashes"):
                if k in d:
                    sessions[-1][k] = d[k]
    blob = json.dumps({"made_by": "oracle/make_apply_campaign.py: oracle/js/apply_campaign.js " + " ".join(specs) + " on the unmodified reference",
                       "pool": plist, "sessions": sessions}).encode()
    out = os.path.join(ROOT, "tests", "golden", name)
    with open(out, "wb") as f:
        f.write(gzip.compress(blob, 9, mtime=0))
    print(f"{len(sessions)} sessions, {sum(len(s['calls']) for s in sessions)} calls, {len(plist)} changes -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
    main(LIST_SPECS, "apply_campaign_lists.json.gz")
    main(CONFLICT_SPECS, "apply_campaign_conflicts.json.gz")
    main(LOADED_SPECS, "apply_campaign_loaded.json.gz", loaded=True)
