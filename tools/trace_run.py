"""Debug / trace helper: stage the headline log once, replay it a few times, host-side laps of the last replay on stderr
(AM355_TRACE). `python tools/trace_run.py [torch] [quiet] [restage] [n_replays]`"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "torch" in sys.argv:
    import torch  # noqa: F401  (its bundled HIP runtime then serves the engine too, as under pytest / bench.py)
from automerge_classic_amd import engine, loggen  # noqa: E402
n = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 6
quiet = "quiet" in sys.argv
name = next((a for a in sys.argv[1:] if a.startswith("c")), "c4_text_single")
log = loggen.config(name, 1.0, False)
eng = engine.Engine(0)
eng.load_changes(log)
print("staged", file=sys.stderr, flush=True)
for i in range(n):
    if i == n - 1 and not quiet:
        os.environ["AM355_TRACE"] = "1"
    if "restage" in sys.argv:
        eng.load_changes(log)
    eng.replay()
    if not quiet:
        print("replay", i, "ok", file=sys.stderr, flush=True)
print("done", file=sys.stderr, flush=True)
