"""Host-side lap times (AM355_TRACE) of am355_save of a replayed bench workload: workload, scale."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import engine, loggen  # noqa: E402

name, scale = sys.argv[1], float(sys.argv[2])
log = loggen.config(name, scale)
eng = engine.Engine(0)
eng.load_changes(log)
eng.replay()
for i in range(4):
    if i == 3:
        os.environ["AM355_TRACE"] = "1"
    t0 = time.perf_counter()
    doc = eng.save()
    dt = time.perf_counter() - t0
print({"workload": name, "scale": scale, "save_ms": round(dt * 1e3, 3), "doc_bytes": len(doc)})
