"""Host-side lap times (AM355_TRACE) of the last repetition of one am355_apply_changes case: workload, scale, batch fraction."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TRACE_LAST"] = "1"
import time_apply  # noqa: E402

for r in time_apply.measure(sys.argv[1], float(sys.argv[2]), [float(sys.argv[3])], reps=4):
    print(r)
