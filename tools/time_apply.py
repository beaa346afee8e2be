"""Timing of am355_apply_changes (Backend.applyChanges with its incremental patch) on the bench workloads: host buffers in, patch
record tables in host memory out. Run on the GPU box."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from automerge_classic_amd import engine, loggen  # noqa: E402
from automerge_classic_amd.loggen import ChangeLog  # noqa: E402
from test_apply_engine import split_log  # noqa: E402


def measure(name, scale, fractions, reps=5):
    log = loggen.config(name, scale)
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    changes = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    n = len(changes)
    out = []
    for frac in fractions:
        k = max(1, int(n * frac))
        base, batch = ChangeLog.from_changes(changes[:n - k]) if n - k else None, ChangeLog.from_changes(changes[n - k:])
        best = None
        eng = engine.Engine(0)
        for rep in range(reps):
            eng.reset()
            if base is not None:
                eng.apply_changes(base)
            if rep == reps - 1 and os.environ.get("TRACE_LAST"):
                os.environ["AM355_TRACE"] = "1"
                print("---", name, "batch", k, "of", n, file=sys.stderr)
            t0 = time.perf_counter()
            eng.apply_changes(batch)
            t1 = time.perf_counter()
            os.environ.pop("AM355_TRACE", None)
            st = eng.stats()
            best = min(best, t1 - t0) if best else t1 - t0
        eng.close()
        ops_batch = None
        out.append({"workload": name, "scale": scale, "batch_changes": k, "doc_changes": n - k, "ms": round(best * 1e3, 3), "total_ops": int(st.n_ops)})
    return out


if __name__ == "__main__":
    res = []
    res += measure("c4_text_single", 1.0, [1.0, 0.5, 0.1, 0.01])
    res += measure("c4_text_multi", 1.0, [1.0, 0.1])
    res += measure("c3_map_lww", 1.0, [1.0, 0.125])
    res += measure("c2_text_typing", 1.0, [1.0, 0.1])
    for r in res:
        print(json.dumps(r))
