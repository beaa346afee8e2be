"""A kept state and then small am355_apply_changes calls onto it, one after the other (for rocprofv3 --kernel-trace): workload, scale,
changes per call, calls. The base is everything but the last calls x per changes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import engine, loggen  # noqa: E402
from automerge_classic_amd.loggen import ChangeLog  # noqa: E402

name, scale, per, calls = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
log = loggen.config(name, scale)
arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
changes = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
k = per * calls
eng = engine.Engine(0, os.environ["AM355_TOOL_LIB"]) if os.environ.get("AM355_TOOL_LIB") else engine.Engine(0)   # (AM355_TOOL_LIB: the CPU emulation of tests/emu, for host-side timings)
eng.apply_changes(ChangeLog.from_changes(changes[:len(changes) - k]))
times = []
for j in range(calls):
    b = ChangeLog.from_changes(changes[len(changes) - k + j * per:len(changes) - k + (j + 1) * per])
    t0 = time.perf_counter()
    eng.apply_changes(b)
    times.append((time.perf_counter() - t0) * 1e3)
print("ms per call:", " ".join("%.3f" % t for t in times), " median %.3f" % sorted(times)[len(times) // 2], " resident counters", eng.resident_counters())
eng.close()
