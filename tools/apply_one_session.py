"""Diagnostic: one committed campaign session through the engine, the whole-document patch after every call checked against the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from automerge_classic_amd import engine  # noqa: E402
from automerge_classic_amd.loggen import ChangeLog  # noqa: E402
from test_apply_engine import load_campaign  # noqa: E402
from test_apply_vectors import same_patch  # noqa: E402

sessions, pool = load_campaign()
lib = sys.argv[2] if len(sys.argv) > 2 else engine.DEFAULT_LIB
for s in sessions:
    if s["name"] != sys.argv[1]:
        continue
    eng = engine.Engine(0, lib)
    ses = oracle_lib.OracleSession()
    given = []
    for ci, call in enumerate(s["calls"]):
        batch = [pool[k] for k in call]
        given += batch
        want = ses.apply(batch)
        print("call", ci, flush=True)
        e2 = engine.Engine(0, lib)
        e2.load_changes(ChangeLog.from_changes(given))
        e2.replay()
        print("   bulk replay of everything given so far == oracle getPatch:", same_patch(e2.patch_json(), ses.patch_json()), "fast path", e2.stats().fast_path, flush=True)
        e2.close()
        try:
            eng.apply_changes(ChangeLog.from_changes(batch))
            print("   apply patch == oracle:", same_patch(eng.apply_patch_json(), want), flush=True)
        except engine.UnsupportedChanges as e:
            print("   refused", str(e)[:80])
            print("   state after the refused call == oracle getPatch:", same_patch(eng.patch_json(), ses.patch_json()))
            break
