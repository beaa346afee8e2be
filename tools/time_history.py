"""History after load of the saved headline log on the GPU: per-phase trace of am355_doc_changes + best-of timings."""
import os, sys, time
sys.path.insert(0, '.')
from automerge_classic_amd import engine, loggen
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
log = loggen.config("c4_text_single", scale, False)
eng = engine.Engine(0)
eng.load_changes(log); eng.replay(); doc = eng.save()
for deflate in (False, True):
    best = None
    for i in range(6):
        eng.load_document(doc); eng.replay()
        if i == 5: os.environ["AM355_TRACE"] = "1"
        t = time.perf_counter(); r = eng.doc_changes(deflate=deflate); dt = (time.perf_counter() - t) * 1e3
        os.environ.pop("AM355_TRACE", None)
        best = dt if best is None else min(best, dt)
    print(f"doc_changes(deflate={deflate}): best of 6 {best:.3f} ms, {len(r[1]) - 1} changes, {int(r[1][-1])} bytes", flush=True)
