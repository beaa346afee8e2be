"""One am355_apply_changes case repeated (for rocprofv3 --kernel-trace --stats): workload, scale, fraction of the changes in the batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import engine, loggen  # noqa: E402
from automerge_classic_amd.loggen import ChangeLog  # noqa: E402

name, scale, frac, reps = sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
log = loggen.config(name, scale)
arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
changes = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
k = max(1, int(len(changes) * frac))
base = ChangeLog.from_changes(changes[:len(changes) - k]) if len(changes) > k else None
batch = ChangeLog.from_changes(changes[len(changes) - k:])
eng = engine.Engine(0)
for _ in range(reps):
    eng.reset()
    if base is not None:
        eng.apply_changes(base)
    eng.apply_changes(batch)
eng.close()
