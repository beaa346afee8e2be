#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace) as a per-kernel table (calls, total, avg, min, max, %).

  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o run -- python bench.py ...
  python tools/rocpd_summary.py gpurun_out/prof/run_results.db [replays] > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    q = f"""select s.display_name, count(*), sum(d.end-d.start)/1000.0, avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0,
            max(d.end-d.start)/1000.0 from {disp} d join {sym} s on d.kernel_id=s.id group by s.display_name order by 3 desc"""
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}")
    for r in rows:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]:10.1f} {r[3]:9.2f} {r[4]:8.2f} {r[5]:8.2f} {100*r[2]/tot:6.1f}")
    print(f"total kernel time: {tot:.1f} us" + (f"  ({tot/replays:.1f} us per replay over {replays} replays)" if replays else ""))


if __name__ == "__main__":
    main()
