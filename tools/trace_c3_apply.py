"""Per-call wall time of am355_apply_changes of the whole c3_map_lww log onto an empty context, ten calls in a row (A/B of engine switches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import engine, loggen
name = sys.argv[1] if len(sys.argv) > 1 else "c3_map_lww"
log = loggen.config(name, 1.0)
eng = engine.Engine(0)
ts = []
for rep in range(12):
    eng.reset()
    t0 = time.perf_counter()
    eng.apply_changes(log)
    ts.append((time.perf_counter() - t0) * 1e3)
print(name, " ".join("%.3f" % t for t in ts))
