#!/usr/bin/env python3
"""Replay of larger batches than the headline (c4 logs at 4x / 8x): T_replay, T_device, parity against the oracle.
  python tools/big_batch.py [workload] [scales...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (first: its HIP runtime then serves the engine too)
from automerge_classic_amd import engine, loggen  # noqa: E402
import oracle_lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c4_text_single"
scales = [float(x) for x in sys.argv[2:]] or [4.0]
eng = engine.Engine(0)
for s in scales:
    t0 = time.time()
    log = loggen.config(wl, s, False)
    t_gen = time.time() - t0
    best_r, best_d = 1e9, 1e9
    for it in range(6):
        t0 = time.perf_counter()
        eng.load_changes(log)
        eng.replay()
        eng.fetch_ir()
        dt = (time.perf_counter() - t0) * 1e3
        st = eng.stats()
        if it >= 1:
            best_r, best_d = min(best_r, dt), min(best_d, st.ms_total)
    got = eng.patch_json()
    t0 = time.time()
    want = oracle_lib.OracleDoc(log).patch_json()
    t_or = time.time() - t0
    print(f"{wl} x{s}: {log.n_ops} ops, {log.n_changes} changes, {len(log.arena)} bytes (generated in {t_gen:.1f} s): T_replay {best_r:.2f} ms = "
          f"{log.n_ops / best_r / 1e3:.0f} M ops/s, replay call {best_d:.2f} ms = {log.n_ops / best_d / 1e3:.0f} M ops/s, parity with the oracle ({t_or:.1f} s): {got == want}",
          flush=True)
