#!/usr/bin/env python3
"""Per-step wall clock of T_replay (load_changes + replay + fetch_ir) right after start-up: shows outliers the mean of bench.py hides.
  python tools/step_times.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from automerge_classic_amd import engine, loggen  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
log = loggen.config("c4_text_single", 1.0, False)
eng = engine.Engine(0)
ts = []
for i in range(n):
    t0 = time.perf_counter()
    eng.load_changes(log)
    t1 = time.perf_counter()
    eng.replay()
    t2 = time.perf_counter()
    eng.fetch_ir()
    t3 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
for i, (a, b, c) in enumerate(ts):
    print(f"step {i:2d}: load {a:7.3f}  replay {b:7.3f}  fetch {c:7.3f}  total {a + b + c:7.3f} ms")
