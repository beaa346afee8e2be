#!/usr/bin/env python3
"""Timeline of ONE replay out of a rocprofv3 --kernel-trace rocpd database: every dispatch between the n-th and the (n+1)-th
k_parse_changes launch, with its start offset, duration, queue and the idle gap since the previous dispatch ended on that queue.

  python tools/rocpd_timeline.py gpurun_out/prof/run_results.db [which=-2] [anchor kernel=k_parse_changes] > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    q = f"select s.display_name, d.start, d.end, {('d.' + qcol) if qcol else '0'} from {disp} d join {sym} s on d.kernel_id=s.id order by d.start"
    rows = list(cur.execute(q))
    anchor = sys.argv[3] if len(sys.argv) > 3 else "k_parse_changes"   # (document loads: k_scan_term_sums)
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    a = marks[which]
    b = marks[which + 1] if which + 1 < 0 or which + 1 < len(marks) and which >= 0 else len(rows)
    if which == -1:
        b = len(rows)
    t0 = rows[a][1]
    last_end = {}
    busy = 0.0
    print(f"# replay #{which}: dispatches {a}..{b - 1}; times in us relative to the {anchor} launch")
    print(f"{'start':>9s} {'dur':>8s} {'gap':>7s} {'q':>3s}  kernel")
    for name, s, e, qid in rows[a:b]:
        gap = (s - last_end[qid]) / 1000.0 if qid in last_end else 0.0
        last_end[qid] = e
        busy += (e - s) / 1000.0
        short = name.replace("am355::", "").split("(")[0][:70]
        print(f"{(s - t0) / 1000.0:9.2f} {(e - s) / 1000.0:8.2f} {gap:7.2f} {qid:3d}  {short}")
    end = max(r[2] for r in rows[a:b])
    print(f"# span {(end - t0) / 1000.0:.1f} us, kernel-busy (all queues) {busy:.1f} us, {b - a} dispatches")


if __name__ == "__main__":
    main()
