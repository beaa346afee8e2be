#!/usr/bin/env python3
"""Wall-clock split of one T_replay step into its three C-ABI calls: am355_load_changes / am355_load_document
(host gather or inflate + H2D), am355_replay (device), am355_fetch_ir (D2H of the patch IR).  Median of K runs after 2 warm-ups.

  python tools/time_stages.py [workload ...]      # c4_text_single c4_text_single+deflate c3_map_lww c5_doc_mixed
"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automerge_classic_amd import engine, loggen  # noqa: E402


def main():
    names = sys.argv[1:] or ["c4_text_single", "c4_text_single+deflate", "c3_map_lww", "c5_doc_mixed"]
    eng = engine.Engine(0)
    out = []
    for name in names:
        base, _, opt = name.partition("+")
        if base == "c5_doc_mixed":
            doc, rows = loggen.document_config(1.0)
            stage = lambda: eng.load_document(doc)  # noqa: E731
        else:
            log = loggen.config(base, 1.0, opt == "deflate")
            stage = lambda: eng.load_changes(log)  # noqa: E731
        ts = {"stage": [], "replay": [], "fetch_ir": []}
        k = 3 if base == "c5_doc_mixed" else 15
        for it in range(k + 2):
            t0 = time.perf_counter(); stage()
            t1 = time.perf_counter(); eng.replay()
            t2 = time.perf_counter(); eng.fetch_ir()
            t3 = time.perf_counter()
            if it >= 2:
                ts["stage"].append(t1 - t0); ts["replay"].append(t2 - t1); ts["fetch_ir"].append(t3 - t2)
        st = eng.stats()
        med = {k2: statistics.median(v) * 1e3 for k2, v in ts.items()}
        tot = sum(med.values())
        out.append({"workload": name, "n_ops": int(st.n_ops), "raw_bytes": int(st.raw_bytes), "ir_bytes": int(st.ir_bytes), "ms": med, "ms_total": tot,
                    "ops_per_s": st.n_ops / (tot * 1e-3)})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
