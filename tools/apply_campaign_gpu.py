"""Diagnostic: the committed applyChanges campaign sessions through the engine, one line per call (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from automerge_classic_amd import engine  # noqa: E402
from automerge_classic_amd.loggen import ChangeLog  # noqa: E402
from test_apply_engine import load_campaign  # noqa: E402
from test_apply_vectors import same_patch  # noqa: E402

sessions, pool = load_campaign()
for s in sessions:
    eng = engine.Engine(0)
    for ci, (call, want) in enumerate(zip(s["calls"], s["patches"])):
        print(s["name"], "call", ci, len(call), "changes", flush=True)
        try:
            eng.apply_changes(ChangeLog.from_changes([pool[k] for k in call]))
            got = eng.apply_patch_json()
        except engine.UnsupportedChanges as e:
            print("   refused:", str(e)[:100], flush=True)
            break
        print("   equal" if same_patch(got, want) else "   DIFFERENT", flush=True)
    eng.close()
