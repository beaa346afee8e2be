#!/usr/bin/env python3
"""Kernels AND memory copies of one step (load + replay + fetch) out of a rocprofv3 --kernel-trace --memory-copy-trace rocpd database,
in time order relative to the first H2D copy of the step.   python tools/rocpd_step_timeline.py db [which=-2]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
mc = next((t for t in tabs if t.startswith("rocpd_memory_copy")), None)
ev = [(s, e, "K " + n.replace("am355::", "").split("(")[0][:60]) for n, s, e in cur.execute(f"select s.display_name, d.start, d.end from {disp} d join {sym} s on d.kernel_id=s.id")]
if mc:
    cols = [r[1] for r in cur.execute(f"pragma table_info({mc})")]
    size = "size" if "size" in cols else ("bytes" if "bytes" in cols else "0")
    name = "name" if "name" in cols else "''"
    for s, e, b, n in cur.execute(f"select start, end, {size}, {name} from {mc}"):
        ev.append((s, e, "C %s %d bytes" % (n, b)))
ev.sort()
marks = [i for i, x in enumerate(ev) if "k_parse_changes" in x[2]]
a = marks[which]
# back up to the first copy of the step (the H2D groups in front of the parse kernel)
lo = a
while lo > 0 and ev[lo - 1][2].startswith("C") and ev[a][0] - ev[lo - 1][0] < 1_000_000:
    lo -= 1
hi = marks[which + 1] if which + 1 < len(marks) and which + 1 != 0 else len(ev)
while hi > a and ev[hi - 1][2].startswith("C") and hi - 1 > a and ev[hi - 1][0] > ev[a][0] + 600_000:
    hi -= 1
t0 = ev[lo][0]
for s, e, n in ev[lo:hi]:
    print("%9.2f %9.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
