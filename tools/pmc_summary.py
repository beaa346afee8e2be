#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, CSV output) into a per-kernel HBM-traffic table and,
optionally, the JSON bench.py reads for `roofline.traffic`.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d F -o f --output-format csv -- <cmd>
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d W -o w --output-format csv -- <cmd>
  python tools/pmc_summary.py F/f_counter_collection.csv W/w_counter_collection.csv [--json out.json --kernel k_decode_wave]

Units: the counters are reported in KB. Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE under-reports wide
coalesced streaming reads by 2x on gfx950: the FETCHx2 column applies that correction; WRITE_SIZE is quoted raw."""
import argparse
import collections
import csv
import json
import re


def short(name):
    m = re.match(r"(?:void )?(?:am355::)?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    base = m.group(1) if m else name
    if m and m.group(2) and "WaveLdsT" in m.group(2):
        base += "<small>" if "256" in m.group(2).split(",")[0] else "<large>"
    return base


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = agg[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--json")
    ap.add_argument("--kernel", default="k_decode_wave<small>")
    ap.add_argument("--top", type=int, default=24)
    args = ap.parse_args()
    fe, wr = load(args.fetch_csv, "FETCH_SIZE"), load(args.write_csv, "WRITE_SIZE")
    names = sorted(set(fe) | set(wr), key=lambda k: -(fe[k][1] * 2 + wr[k][1]))
    print(f"{'kernel':36s} {'launches':>8s} {'FETCH_KB':>11s} {'FETCHx2_KB':>11s} {'WRITE_KB':>11s}   (per launch)")
    for k in names[:args.top]:
        n = max(fe[k][0], wr[k][0], 1)
        f_kb = fe[k][1] / max(fe[k][0], 1)
        w_kb = wr[k][1] / max(wr[k][0], 1)
        print(f"{k:36s} {n:8d} {f_kb:11.1f} {2 * f_kb:11.1f} {w_kb:11.1f}")
    if args.json:
        k = args.kernel
        f_b = fe[k][1] / max(fe[k][0], 1) * 1024
        w_b = wr[k][1] / max(wr[k][0], 1) * 1024
        with open(args.json, "w") as f:
            json.dump({"kernel": k, "launches": max(fe[k][0], wr[k][0]), "fetch_bytes_per_launch": f_b, "write_bytes_per_launch": w_b,
                       "fetch_x2_corrected_bytes": 2 * f_b, "traffic_bytes_per_launch": 2 * f_b + w_b,
                       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/pmc_summary.py; FETCH corrected x2 per "
                                 "MI355X_MICROARCH.md (gfx950 streaming reads), WRITE raw"}, f)


if __name__ == "__main__":
    main()
