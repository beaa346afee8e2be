#!/usr/bin/env python3
"""Per-kernel roofline table of the headline replay (what bench.py embeds as roofline.kernels):

  python tools/kernel_table.py --db run_results.db --fetch F_counter_collection.csv --write W_counter_collection.csv \
         --line bench_line.json --replays 26 --out profiles/r02_kernel_table.json > profiles/r02_kernel_table.txt

Inputs: the rocpd database of `rocprofv3 --kernel-trace --stats -- python bench.py --no-sublines --no-cpu-baseline` (kernel
durations, launches), the two PMC passes of the same command (FETCH_SIZE / WRITE_SIZE per dispatch, KB), and that run's bench line
(n_ops, encoded bytes, pred entries, patch-IR bytes, list elements). For every kernel: launches per replay, average duration,
ALGORITHMIC bytes per launch (the formula is printed with it), HBM bytes from the counters (FETCH doubled for the wide streaming
reads as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950, WRITE raw), and algorithmic bytes / duration against the
8 TB/s peak."""
import argparse
import collections
import csv
import json
import re
import sqlite3


def short(name):
    m = re.match(r"(?:void )?(?:am355::)?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    base = m.group(1) if m else name
    if m and m.group(2) and "WaveLdsT" in m.group(2):
        base += "<small>" if "256" in m.group(2).split(",")[0] else "<large>"
    elif m and m.group(2) and base in ("k_child_order", "k_scan_apply", "k_scan2_apply"):
        base += m.group(2)
    return base


def pmc(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    if not path:
        return agg
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                a = agg[short(row["Kernel_Name"])]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--db", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--line", required=True)
    ap.add_argument("--replays", type=int, required=True)
    ap.add_argument("--out")
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    with open(a.line) as f:
        line = json.load(f)
    wl = line["config"]["workload"]
    N = int(re.search(r"(\d+) ops", wl).group(1))
    C = int(re.search(r"(\d+) changes", wl).group(1))
    RAW = int(re.search(r"(\d+) encoded bytes", wl).group(1))
    P = int(line["roofline"].get("n_preds", 0))
    IR = int(round(line["algorithmic_bytes_per_op"]["P_patch_ir"] * N))
    L = int(line.get("n_list_elems", 0)) or int(0.8 * N)
    # algorithmic bytes per launch: what the kernel must read and write once (u32 fields unless noted)
    ALG = {
        "k_parse_changes": (RAW + 176 * C, "change bytes read once + one 176-byte ChangeMeta per change"),
        "k_hash_changes": (RAW + 32 * C, "change bytes read once + 32-byte digest per change"),
        "k_actor_intern": (17 * 65 * C, "one actor-table entry (16-byte id + length) per (change, actor)"),
        "k_actor_first": (4 * 65 * C + 8 * C, "one provisional actor number per (change, actor) read, the change's latest first-use written"),
        "k_actor_check": (176 * C + 32 * C, "ChangeMeta read + 32-byte brief written per change (its last workgroup ranks the distinct actor ids: three dependent loads)"),
        "k_plan": (32 * C + 24 * C, "brief read + plan written per change"),
        "k_decode_wave<small>": (RAW + 53 * N + 8 * P, "encoded bytes read once + 53-byte op row + 8 bytes per pred written once"),
        "k_decode_wave<large>": (RAW + 53 * N + 8 * P, "as the small class"),
        "k_resolve": (29 * N + 8 * P + 9 * N + 4 * P, "row fields read (obj, key, id, action, insert, pred range: 29 B) + pred ids + obj_row/ref_row/kind written + one succ counter per pred"),
        "k_emit": (10 * N, "kind, action, succ count read per row; kind byte written for visible inserts"),
        "k_compact_rows": (5 * N + 4 * L, "kind + action read per row, one insert-list entry written per list element"),
        "k_child_push": (8 * L + 8 * L, "insert list + reference row read, child link written, one exchange per element"),
        "k_child_order<false>": (24 * L, "insert list, parent, child list, own id read; sibling link + run flag written"),
        "k_run_heads": (8 * L + 4 * L, "run flags read, run prefix written"),
        "k_list_order": (12 * L + 4 * L, "insert list, run prefix, object read; order written"),
        "k_list_order_objs": (12 * L + 4 * L, "insert list, run prefix, object read; order written (+ the object counts and their prefix, per workgroup, from L2)"),
        "k_map_small_finish": (0, "a handful of map emissions: latency"),
        "k_list_counts": (9 * L + 8 * L, "order, value count, kind read; visibility + count written"),
        "k_list_scan": (8 * L + 8 * L, "visibility + count read, two prefixes written"),
        "k_list_edits": (16 * L + IR, "order, prefixes, kind read; per-element edit entries written"),
        "k_edit_runs": (24 * IR // 8, "per-element edit entries + row ids / value classes read, flags written"),
        "k_edit_pack": (IR + 20 * IR // 8, "per-element entries read, value + edit records (the patch IR) written"),
        "k_euler_rank_lds": (0, "working set in LDS (2 x runs + 1 tour entries of 8 bytes): no HBM stream"),
    }
    db = sqlite3.connect(a.db)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = collections.defaultdict(lambda: [0, 0.0])
    for name, s, e in cur.execute(f"select s.display_name, d.start, d.end from {disp} d join {sym} s on d.kernel_id=s.id"):
        r = rows[short(name)]
        r[0] += 1
        r[1] += (e - s) / 1000.0
    if "k_parse_changes" in rows:
        a.replays = rows["k_parse_changes"][0]  # (exactly one launch per replay: the pre-warm of bench.py makes the count box dependent)
    fe, wr = pmc(a.fetch, "FETCH_SIZE"), pmc(a.write, "WRITE_SIZE")
    out = []
    order = sorted(rows, key=lambda k: -rows[k][1])
    total_us = sum(v[1] for v in rows.values())
    print(f"# per-kernel table of the headline replay ({N} ops, {C} changes, {RAW} encoded bytes, {P} pred entries, {IR} patch-IR bytes); {a.replays} replays in the trace")
    print(f"{'kernel':28s} {'per_replay':>10s} {'avg_us':>8s} {'%time':>6s} {'alg_MB':>8s} {'pmc_MB':>8s} {'GB/s':>8s} {'frac':>7s}")
    for k in order:
        calls, tot = rows[k]
        if k.startswith("ke_") or k.startswith("ks_"):
            continue  # (Backend.save kernels: outside the replay)
        avg = tot / calls
        alg = ALG.get(k, (None, ""))[0]
        f_b = fe[k][1] / fe[k][0] * 1024 if fe[k][0] else None
        w_b = wr[k][1] / wr[k][0] * 1024 if wr[k][0] else None
        traffic = (2 * f_b + w_b) if f_b is not None and w_b is not None else None
        gbs = alg / (avg * 1e-6) / 1e9 if alg else None
        rec = {"kernel": k, "launches_per_replay": round(calls / a.replays, 2), "avg_us": round(avg, 2), "pct_of_kernel_time": round(100 * tot / total_us, 1),
               "algorithmic_bytes": alg, "algorithmic_bytes_formula": ALG.get(k, (None, ""))[1] or None,
               "pmc_fetch_bytes_x2": round(2 * f_b) if f_b is not None else None, "pmc_write_bytes": round(w_b) if w_b is not None else None,
               "pmc_traffic_bytes": round(traffic) if traffic is not None else None,
               "algorithmic_GB_per_s": round(gbs, 1) if gbs else None, "frac_of_hbm_peak": round(gbs / 8000.0, 4) if gbs else None}
        out.append(rec)
        print(f"{k:28s} {calls / a.replays:10.2f} {avg:8.2f} {100 * tot / total_us:6.1f} {(alg or 0) / 1e6:8.2f} {(traffic or 0) / 1e6:8.2f} "
              f"{(gbs or 0):8.1f} {(gbs or 0) / 8000.0:7.4f}")
    per_replay_traffic = sum((r["pmc_traffic_bytes"] or 0) * r["launches_per_replay"] for r in out)
    print(f"# HBM traffic per replay from the counters (sum over kernels, FETCH x2 + WRITE): {per_replay_traffic / 1e6:.1f} MB; kernel time per replay {total_us / a.replays:.1f} us")
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"source": a.source, "n_ops": N, "n_changes": C, "encoded_bytes": RAW, "n_preds": P, "patch_ir_bytes": IR,
                       "traffic_bytes_per_replay": round(per_replay_traffic), "kernels": out[:16]}, f, indent=1)


if __name__ == "__main__":
    main()
