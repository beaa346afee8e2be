"""am355_replay alone on a staged log (what bench.py's t_device_ms times): workload, scale, [library]. AM355_TRACE=1 for the last call's laps."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import engine, loggen  # noqa: E402

name, scale = sys.argv[1], float(sys.argv[2])
lib = sys.argv[3] if len(sys.argv) > 3 else None
log = loggen.config(name, scale)
eng = engine.Engine(0, lib) if lib else engine.Engine(0)
eng.load_changes(log)
times = []
for i in range(30):
    if i == 29 and os.environ.get("TRACE_LAST"):
        os.environ["AM355_TRACE"] = "1"
    t0 = time.perf_counter()
    eng.replay()
    times.append((time.perf_counter() - t0) * 1e3)
os.environ.pop("AM355_TRACE", None)
print(name, lib or "current", "replay ms: first %.3f  median %.3f  min %.3f" % (times[0], sorted(times)[15], min(times)))
eng.close()
