"""Lap times (AM355_TRACE) of am355_load_changes on the DEFLATEd headline log (every change a chunk of type 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import engine, loggen
log = loggen.config("c4_text_single", 1.0, True)
print("compressed bytes", int(log.offsets[-1]), "changes", log.n_changes, file=sys.stderr)
eng = engine.Engine(0)
for i in range(6):
    if i == 5:
        os.environ["AM355_TRACE"] = "1"
    t0 = time.perf_counter()
    eng.load_changes(log)
    t1 = time.perf_counter()
    eng.replay()
    t2 = time.perf_counter()
    os.environ.pop("AM355_TRACE", None)
    print("load_changes %.3f ms, replay %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), file=sys.stderr)
