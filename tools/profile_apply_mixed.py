"""Map changes onto a document that also holds a large Text (am355_apply_changes onto a kept state): the headline text log + a map log side by
side (two sets of actors); the base holds the whole text log and the map log's first round, then `calls` calls of `per` map changes each.
  python tools/profile_apply_mixed.py [text scale = 1.0] [per = 1] [calls = 40] [both]     AM355_NO_MAPS_ONLY=1: the whole merge per call
  both: every call also holds one change of the text log (held back from the base): list rows and map rows in one batch"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import engine, loggen  # noqa: E402
from automerge_classic_amd.loggen import ChangeLog  # noqa: E402


def changes_of(log):
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    return [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]


scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 40
NA = 8
text = changes_of(loggen.config("c4_text_single", scale))
maps = changes_of(loggen.generate(loggen.KIND_MAP_LWW, n_actors=NA, n_rounds=2 + (per * calls + NA - 1) // NA, n_keys=64, seed=4242))
eng = engine.Engine(0, os.environ["AM355_TOOL_LIB"]) if os.environ.get("AM355_TOOL_LIB") else engine.Engine(0)
both = len(sys.argv) > 4 and sys.argv[4] == "both"
held = calls if both else 0
eng.apply_changes(ChangeLog.from_changes(text[:len(text) - held] + maps[:NA]))
times = []
for j in range(calls):
    b = ChangeLog.from_changes(([text[len(text) - held + j]] if both else []) + maps[NA + j * per:NA + (j + 1) * per])
    t0 = time.perf_counter()
    eng.apply_changes(b)
    times.append((time.perf_counter() - t0) * 1e3)
print("ms per call:", " ".join("%.3f" % t for t in times), " median %.3f" % sorted(times)[len(times) // 2], " resident counters", eng.resident_counters(),
      " maps only", eng.resident_maps_only_calls())
eng.close()
