#!/usr/bin/env python3
"""bench.py -- bulk change-replay throughput of the MI355X engine (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic changes, timed as SURVEY.md §8(d) defines T_replay:
from "array of binary changes in HOST memory" to "patch IR + envelope in HOST memory" -- host inflate + staging, H2D,
container parse + SHA-256 + column decode -> causal schedule -> op-set merge -> RGA order -> whole-document patch IR, D2H.
`value` = ops / T_replay.

stdout carries ONE compact JSON line (< 6 KB, strict JSON: compact_line) -- the contract keys, `t_device_ms`, the whole-path `roofline`
(N_ops x A / T_device with A = E + R + P algorithmic bytes per op) with its dominant kernel as measured by a
rocprofv3-traced child of this run, `cpu_baseline`, and one short row per sub-workload. The whole record -- live per-kernel table with PMC
bytes, phase brackets, every sub-workload in full (c4_text_multi, c3_map_lww, c2_text_typing, the headline log shuffled / DEFLATEd / x4,
c5_doc_mixed = Backend.load), sharding model, save / history / applyChanges / JS end-to-end timings -- goes to --detail
(gpurun_out/bench_detail.json) and to stderr. Before anything is timed the engine's getPatch text of the workload is compared with
the CPU oracle's on the same bytes (sha256; N = 1): a mismatch ends the run without a line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4_text_single] [--scale 1.0] [--no-sublines]

For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU: every rank replays its own document of
the same shape (different seed); value = total ops of all ranks / max time over ranks ("weak": one document needs one GPU for
0.6 ms, a node serves many documents). The same line then carries `sharded`: ONE document (c4_text_multi, 64 Text objects) split
by objectId over the N ranks (automerge_classic_amd/shard.py) -- every rank decodes the batch and merges the
objects it owns, the patch-IR fragments are all-gathered over RCCL and stitched on rank 0 -- as strong-scaling figures next to
the single-GPU time of the same log measured in the same run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (generator kind, parameters at scale 1.0)   -- BASELINE.json configs 2, 3, 4
    "c2_text_typing": ("typing", dict(n_ops=100_000, ops_per_change=100)),
    "c3_map_lww": ("map", dict(n_actors=32, n_rounds=8, n_keys=10_000)),
    "c4_text_single": ("text", dict(n_actors=64, n_rounds=64, ins_per_change=200, del_per_change=50, n_objects=1)),
    "c4_text_multi": ("text", dict(n_actors=64, n_rounds=64, ins_per_change=200, del_per_change=50, n_objects=64)),
    # BASELINE config 5: Backend.load of a saved document (~10 M rows at scale 1.0: Text / nested maps / lists)
    "c5_doc_mixed": ("doc", dict()),
}
SHAPE = {"c2_text_typing": "one Text object, 1 actor", "c3_map_lww": "root map of 10 k keys, multi-value conflicts",
         "c4_text_single": "one Text object", "c4_text_multi": "64 Text objects at root keys", "c5_doc_mixed": "256 Text + nested maps + lists"}
BASE_SEED = {"c2_text_typing": 0x5EED0002, "c3_map_lww": 0x5EED0003, "c4_text_single": 0x5EED0004, "c4_text_multi": 0x5EED0004, "c5_doc_mixed": 0x5EED0005}
PARITY = ("sha256(getPatch text) == CPU oracle's on this log, checked in this run before timing; oracle == block-size-patched reference "
          "on this shape at 124,801 ops (tests/golden/headline_pin.json)")
assert len(PARITY) <= 200
LINE_LIMIT = 6144   # bytes: the driver keeps ~8 KB of stdout tail; the ONE json line must fit whole


def make_log(name, scale, seed, deflate=False):
    from automerge_classic_amd import loggen
    kind, kw = WORKLOADS[name]
    kw = dict(kw, deflate=deflate)
    if kind == "typing":
        kw["n_ops"] = max(1, int(kw["n_ops"] * scale))
        return loggen.generate(loggen.KIND_TEXT_TYPING, seed=seed, name=name, **kw)
    kw["n_rounds"] = max(1, int(kw["n_rounds"] * scale))
    return loggen.generate(loggen.KIND_MAP_LWW if kind == "map" else loggen.KIND_TEXT_CONCURRENT, seed=seed, name=name, **kw)


def oracle_patch_sha256(log):
    """The in-run parity gate's checker (parity before any timing counts): sha256 of the CPU oracle's getPatch text
    for this very log. The oracle is test infrastructure; here it only checks, the engine's own patch is what is compared."""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.lib()
    doc = oracle_lib.OracleDoc(log)
    try:
        return hashlib.sha256(doc.patch_json().encode()).hexdigest()
    finally:
        doc.close()


def oracle_document_patch_sha256(doc_bytes):
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.lib()
    doc = oracle_lib.OracleDoc.load_document(doc_bytes)
    try:
        return hashlib.sha256(doc.patch_json().encode()).hexdigest()
    finally:
        doc.close()


def cpu_baseline(log, budget_s=12.0):
    """The CPU oracle (plain-C port of the reference's algorithm, 1 thread) on the same workload: loadChanges + getPatch
    (as JSON text). A reported baseline, not the optimisation target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.lib()
    best, reps, t_all = None, 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t_all < budget_s and reps < 40):
        t0 = time.perf_counter()
        doc = oracle_lib.OracleDoc(log)
        doc.patch_json()
        dt = time.perf_counter() - t0
        doc.close()
        best = dt if best is None else min(best, dt)
        reps += 1
    return {"value": log.n_ops / best, "unit": "ops/s", "cores": 1, "kind": "port",
            "sample": f"{log.name}: {log.n_ops} ops, {log.n_changes} changes, best of {reps} runs of oracle loadChanges+getPatch "
                      f"({time.perf_counter() - t_all:.1f} s of CPU work)"}


def reference_js_baseline(name, scale_full, seed, timeout_s=240):
    """The UNMODIFIED reference JS backend (north_star: "next to the reference JS backend timed on the box's own host cores in the same
    run"): `Backend.loadChanges(Backend.init(), changes)` + `Backend.getPatch` under node, 1 core, on a BOUNDED sample of the same
    workload (the reference replays ~25-30 k ops/s: the full 1 M-op log would take ~40 s per run) -- oracle/js/ref_patch.js --time.
    Needs node and the reference tree (AUTOMERGE_REF, default /root/reference): present in the build container, absent on the GPU box
    (nothing there may read it), where the caller reports the C port instead and says so. Returns None when it cannot run."""
    import shutil
    import subprocess
    import tempfile
    ref = os.environ.get("AUTOMERGE_REF", "/root/reference")
    node = shutil.which("node")
    if node is None or not os.path.isdir(os.path.join(ref, "backend")):
        return None
    scale = min(scale_full, 0.125 if name.startswith("c4") else 0.5 if name == "c3_map_lww" else 1.0)   # ~100-130 k ops: 10-30 s of CPU work
    log = make_log(name, scale, seed)
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"), AUTOMERGE_REF=ref)
    env.pop("REF_BLOCK_SIZE", None)   # (the stock reference: this leg is about time, not about the patch)
    t_all = time.perf_counter()
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "log.bin")
            log.save(path)
            out = subprocess.run([node, os.path.join(ROOT, "oracle", "js", "ref_patch.js"), path, "--time", "3", "--out", os.path.join(tmp, "patch.json")],
                                 env=env, capture_output=True, text=True, timeout=timeout_s)
        line = [l for l in out.stderr.splitlines() if l.startswith("reference median of")]
        if out.returncode != 0 or not line:
            return None
        ops_per_s = float(line[-1].split("=")[1].split("ops/s")[0])
    except Exception:  # (a baseline that cannot run must not cost the bench line)
        return None
    ver = subprocess.run([node, "--version"], capture_output=True, text=True).stdout.strip()
    return {"value": ops_per_s, "unit": "ops/s", "cores": 1, "kind": "reference",
            "sample": f"{name} x{scale}: {log.n_ops} ops, {log.n_changes} changes (the shape of the timed workload at reduced size), median of 3 runs after 1 warm-up of "
                      f"the unmodified reference's Backend.loadChanges + getPatch under node {ver}, 1 core of {os.cpu_count()} ({time.perf_counter() - t_all:.1f} s of CPU work)"}


def live_kernel_table(timeout_s=170):
    """Per-kernel table measured IN THIS RUN: a traced child of this very script -- `rocprofv3 --kernel-trace --stats`
    for the durations, then one pass each with `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` for the HBM bytes (counters in passes of their
    own, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) -- summarised by tools/kernel_table.py. None when rocprofv3 is not on PATH
    or the kernel-trace pass fails; the byte columns are simply absent when a counter pass fails. The committed table of the round is
    the fallback and the line says which one it carries."""
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None or os.environ.get("AM355_BENCH_CHILD"):
        return None
    tmp = tempfile.mkdtemp(prefix="am355_live_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", AM355_BENCH_CHILD="1")
    # (the child's whole record goes to a file of its own: tools/kernel_table.py reads n_preds / n_list_elems from it, and the parent's
    # --detail file must not be overwritten by a child's)
    child_detail = os.path.join(tmp, "child_detail.json")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--prewarm", "0.3", "--no-sublines", "--no-cpu-baseline",
             "--detail", child_detail]
    t0 = time.perf_counter()
    try:
        def find(sub, suffix):
            for d, _, fs in os.walk(os.path.join(tmp, sub)):
                for f in fs:
                    if f.endswith(suffix):
                        return os.path.join(d, f)
            return None
        with open(os.path.join(tmp, "line.json"), "w") as line:
            r = subprocess.run([prof, "--kernel-trace", "--stats", "-d", os.path.join(tmp, "kt"), "-o", "run", "--"] + child, env=env, cwd="/tmp",
                               stdout=line, stderr=subprocess.DEVNULL, timeout=timeout_s)
        db = find("kt", "_results.db")
        if r.returncode != 0 or db is None:
            return None
        csvs = {}
        for tag, counter in (("pf", "FETCH_SIZE"), ("pw", "WRITE_SIZE")):
            try:
                subprocess.run([prof, "--kernel-trace", "--pmc", counter, "-d", os.path.join(tmp, tag), "-o", tag, "--output-format", "csv", "--"] + child, env=env,
                               cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
                csvs[tag] = find(tag, "counter_collection.csv")
            except Exception:
                csvs[tag] = None
        with open(child_detail) as f:   # (written by the kernel-trace pass; the counter passes rewrite it with the same workload)
            child_rec = json.load(f)
        cmd = [sys.executable, os.path.join(ROOT, "tools", "kernel_table.py"), "--db", db, "--line", child_detail, "--replays", "13",
               "--out", os.path.join(tmp, "table.json"), "--source",
               "measured in this run: rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 10 --warmup 3 --no-sublines --no-cpu-baseline"]
        if csvs.get("pf") and csvs.get("pw"):
            cmd += ["--fetch", csvs["pf"], "--write", csvs["pw"]]
        if subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120).returncode != 0:
            return None
        with open(os.path.join(tmp, "table.json")) as f:
            t = json.load(f)
        t["seconds"] = round(time.perf_counter() - t0, 1)
        t["t_device_ms_under_trace"] = child_rec.get("t_device_ms")
        return t
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def reference_js_recorded(name):
    """The reference JS backend's rate on this workload as recorded in the build container (tools/record_reference_js.py ->
    profiles/rNN_reference_js_baseline.json, the latest round's): the reference tree cannot travel to the GPU box, its dated
    measurement can."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_reference_js_baseline.json")))
    try:
        with open(files[-1]) as f:
            rec = json.load(f)
        w = rec["workloads"][name]
    except Exception:
        return None
    rel = os.path.relpath(files[-1], ROOT)
    return {"value": w["ops_per_s"], "unit": "ops/s", "cores": w["cores"], "kind": "reference", "sample": w["sample"], "recorded": rec["date"], "host": rec["host"],
            "node": rec["node"], "file": rel,
            "note": "NOT measured in this run: recorded with bench.reference_js_baseline in the build container, where node and the reference tree exist"}


def js_end_to_end(log):
    """T_e2e: node -> N-API addon -> GPU -> record tables -> js/materialize.js = the patch OBJECT the frontend consumes,
    through automerge_classic_amd/js/bench_e2e.js on the same log. None when node or the addon is missing."""
    import shutil
    import subprocess
    import tempfile
    node = shutil.which("node")
    js = os.path.join(ROOT, "automerge_classic_amd", "js")
    if node is None or not os.path.exists(os.path.join(js, "am355_napi.node")):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "log.bin")
            log.save(path)
            out = subprocess.run([node, os.path.join(js, "bench_e2e.js"), path, "7"], capture_output=True, text=True, timeout=240)
        if out.returncode != 0:
            return {"error": (out.stderr or out.stdout)[-300:]}
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": str(e)[:300]}
    return {"t_e2e_ms": d["T_e2e_ms"], "t_e2e_ops_per_s": d["T_e2e_ops_per_s"], "t_replay_ms_through_node": d["T_replay_ms"], "ms": d["ms"],
            "timed_region": "change Uint8Arrays in node -> addon.loadChanges -> addon.replay -> addon.fetchIR -> materialize.js (the JS patch object); median of 7"}


def js_apply_latency(log, calls=60):
    """Backend.applyChanges through the JS host (index.js -> addon -> am355_apply_changes -> fetchApplyIR -> materialize.js -> the new
    backend state), one change per call onto the document the rest of the log made: js/bench_apply.js, median of the calls after the
    first. None when node or the addon is missing."""
    import shutil
    import subprocess
    import tempfile
    node = shutil.which("node")
    js = os.path.join(ROOT, "automerge_classic_amd", "js")
    if node is None or not os.path.exists(os.path.join(js, "am355_napi.node")):
        return None
    env = dict(os.environ, AM355_JS_PROFILE="1", NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))   # (index.js requires pako's shim at load)
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "log.bin")
            log.save(path)
            out = subprocess.run([node, os.path.join(js, "bench_apply.js"), path, "1", str(calls)], capture_output=True, text=True, timeout=180, env=env)
        if out.returncode != 0:
            return {"error": (out.stderr or out.stdout)[-300:]}
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": str(e)[:300]}
    pr, n = d.get("profile") or {}, max(1, (d.get("profile") or {}).get("calls", 1))
    return {"ms_per_call": d["median_ms"], "first_call_ms": d["first_call_ms"], "calls": d["calls"], "served_by_engine": d["counters"]["gpuApplyChanges"] - 1,
            "fallback_to_js": d["counters"]["fallbackToJs"],
            "ms": {"engine": pr.get("engine_ms", 0) / n, "fetch_and_materialize": pr.get("patch_ms", 0) / n, "new_state": pr.get("state_ms", 0) / n},
            "reference_js_recorded_ms_per_call": 21.5,
            "timed_region": "Backend.applyChanges(state, [one change]) in node, change Uint8Array in -> [new state, patch object] out; the reference's own Backend "
                            "on the same calls: 21.5 ms median in the build container (profiles/r06_js_apply_latency.txt)"}


def sharded_js(name, scale, gpus, want_sha=None, reps=5, timeout_s=150):
    """ONE document over `gpus` GPUs from the JS host (north_star: host code stays JavaScript): node -> js/sharded.js -> one worker
    process per GPU -> am355_shard_init / am355_sharded_replay (RCCL inside libam355.so: ncclAllGather of the patch-IR fragments) ->
    stitched record tables on rank 0 -> materialize.js. js/bench_sharded.js reports the engine part of a step (max over the workers
    of stage + sharded replay, + rank 0's fetch) and the sha256 of the patch text; with gpus > 1 the same batch also runs through a
    pool of ONE worker, for the ratio. In processes of its own under a hard limit: whatever happens there cannot cost the bench line.
    None when node or the addon is missing."""
    import shutil
    import signal
    import subprocess
    import tempfile
    node = shutil.which("node")
    js = os.path.join(ROOT, "automerge_classic_amd", "js")
    if node is None or not os.path.exists(os.path.join(js, "am355_napi.node")):
        return None
    log = make_log(name, scale, BASE_SEED[name])

    def one(path, g):
        p = subprocess.Popen([node, os.path.join(js, "bench_sharded.js"), path, str(g), str(reps)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             start_new_session=True)
        try:
            out, err = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)   # (the pool's workers are in the group: none is left holding a GPU)
            p.communicate()
            return {"error": f"no answer within {timeout_s} s"}
        if p.returncode != 0:
            return {"error": (err or out)[-300:]}
        return json.loads(out.strip().splitlines()[-1])
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "log.bin")
            log.save(path)
            r = one(path, gpus)
            r1 = one(path, 1) if gpus > 1 and "error" not in r else None
    except Exception as e:   # (an extra: it must not cost the bench line)
        return {"error": str(e)[:300]}
    if "error" in r:
        return {"n_gpus": gpus, "error": r["error"]}
    out = {"workload": f"{name} x{scale}: {log.n_ops} ops, {log.n_changes} changes, ONE document over {gpus} GPU(s) by objectId; node -> js/sharded.js -> worker process per GPU -> am355_sharded_replay (RCCL in the library)",
           "n_gpus": gpus, "engine_ms_per_step": r["engine_ms_per_step"], "ops_per_s": r["ops_per_s"], "per_rank_ms": r["per_rank_ms"], "fragment_bytes": r.get("fragment_bytes"),
           "timed_region": f"per worker: am355_load_changes + am355_sharded_replay, max over the workers, + rank 0's am355_fetch_ir; last of {reps} repetitions"}
    same = []
    if want_sha is not None:
        same.append(r["patch_sha256"] == want_sha)
    if r1 is not None and "error" not in r1:
        out["single_worker_ms_per_step"] = r1["engine_ms_per_step"]
        out["speedup_vs_single_worker"] = r1["engine_ms_per_step"] / r["engine_ms_per_step"]
        same.append(r["patch_sha256"] == r1["patch_sha256"])
    out["parity"] = ("patch text sha256 == the unsharded engine's" if all(same) else "MISMATCH") if same else "not compared"
    return out


def cpu_baseline_document(doc_bytes, n_rows, budget_s=12.0):
    """The CPU oracle's Backend.load + getPatch on the same saved document (1 thread)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.lib()
    best, reps, t_all = None, 0, time.perf_counter()
    while reps < 2 or (time.perf_counter() - t_all < budget_s and reps < 40):
        t0 = time.perf_counter()
        doc = oracle_lib.OracleDoc.load_document(doc_bytes)
        doc.patch_json()
        dt = time.perf_counter() - t0
        doc.close()
        best = dt if best is None else min(best, dt)
        reps += 1
    return {"value": n_rows / best, "unit": "ops/s", "cores": 1, "kind": "port",
            "sample": f"the same {len(doc_bytes)}-byte document, {n_rows} op rows, best of {reps} runs of oracle load+getPatch "
                      f"({time.perf_counter() - t_all:.1f} s of CPU work)"}


PHASES = ("ms_parse", "ms_host_schedule", "ms_decode", "ms_merge", "ms_order", "ms_hash_stream")


class Workload:
    """One staged input (a change log or a saved document) and the two timed regions over it."""

    def __init__(self, eng, name, scale, seed, shuffled=False, deflate=False):
        from automerge_classic_amd import loggen
        self.eng, self.name, self.scale = eng, name, scale
        self.is_doc = WORKLOADS[name][0] == "doc"
        if self.is_doc:
            self.doc_bytes, self.doc_rows = loggen.document_config(scale)
            self.doc_np = np.frombuffer(self.doc_bytes, dtype=np.uint8)   # (the binding hands the buffer over as it is: no 44 MB copy per step)
            self.log = None
        else:
            self.log = make_log(name, scale, seed, deflate)
            if deflate:
                self.log.name = name + "+deflate"
            if shuffled:
                self.log = self.log.reordered(np.random.default_rng(seed & 0xFFFF).permutation(self.log.n_changes))
                self.log.name = name + "+shuffled"

    def stage(self):
        if self.is_doc:
            self.eng.load_document(self.doc_bytes)
        else:
            self.eng.load_changes(self.log)

    def step_replay(self):
        """T_replay: host buffers -> inflate/staging -> H2D -> replay -> patch IR + envelope in host memory."""
        if self.is_doc:   # Backend.load is ONE call of the reference: am355_backend_load (checksum thread beside the device stages)
            self.eng.backend_load(self.doc_np)
        else:
            self.stage()
            self.eng.replay()
        self.eng.fetch_ir()

    def step_device(self):
        """T_device: the replay alone, staged bytes resident in HBM, IR left in HBM."""
        self.eng.replay()

    def describe(self, st):
        if self.is_doc:
            return (f"{self.name} x{self.scale}: Backend.load of a {len(self.doc_bytes)}-byte saved document, {st.n_ops} op rows, {st.n_actors} actors "
                    f"({st.raw_bytes} bytes of inflated op columns), {SHAPE[self.name]}")
        return (f"{self.log.name} x{self.scale}: {st.n_ops} ops, {st.n_changes} changes, {st.n_actors} actors, {st.raw_bytes} encoded bytes, "
                f"{SHAPE[self.name]}")


def timed(fn, steps, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return time.perf_counter() - t0


def algorithmic_bytes(st):
    """A = E + R + P bytes per op: encoded input, one fixed-width op record (53 B here), patch IR."""
    n = max(int(st.n_ops), 1)
    E, R, P = st.raw_bytes / n, 53.0, st.ir_bytes / n
    return {"E_encoded": E, "R_op_record": R, "P_patch_ir": P, "A": E + R + P}


def phase_table(phases, st, n_preds):
    """Live HIP-event brackets (recorded on the engine's stream inside am355_replay) priced against their algorithmic bytes."""
    n, raw = int(st.n_ops), int(st.raw_bytes)
    rows = [
        ("parse + actor intern (k_parse_changes, k_actor_intern, k_actor_check)", phases["ms_parse"], raw + 176 * int(st.n_changes)),
        ("column decode (k_decode_wave<small|large>)", phases["ms_decode"], raw + 53 * n + 8 * n_preds),
        ("merge: resolve + emit + compaction (k_resolve, k_emit, scans)", phases["ms_merge"], 53 * n + 8 * n_preds + 28 * n),
        ("order + patch IR (sibling grouping, list ranking, edits)", phases["ms_order"], 30 * int(st.n_list_elems) + int(st.ir_bytes)),
        ("SHA-256 + dependency resolution (second stream, overlapped)", phases["ms_hash_stream"], raw),
    ]
    out = []
    for name, ms, b in rows:
        if ms > 0:
            gbs = b / (ms * 1e-3) / 1e9
            out.append({"phase": name, "ms": ms, "algorithmic_bytes": b, "GB_per_s": gbs, "frac_of_hbm_peak": gbs / 8000.0})
    return out


def measure_phases(w, steps, sync):
    """Per-phase device times (am355_stats.ms_*): a few replays of the staged input with the HIP events between the phases switched on
    (am355_set_phase_events). They are packets of their own in front of the next kernel, so the timed regions run without them."""
    eng = w.eng
    eng.set_phase_events(True)
    try:
        w.stage()
        for _ in range(3):
            eng.replay()
        steps = max(1, min(steps, 30))
        parts = {k: 0.0 for k in PHASES}
        sync()
        for _ in range(steps):
            eng.replay()
            s = eng.stats()
            for k in parts:
                parts[k] += getattr(s, k)
        sync()
    finally:
        eng.set_phase_events(False)
    return {k: v / steps for k, v in parts.items()}


def run_workload(w, steps, warmup, sync, want_rows=True):
    eng = w.eng
    for _ in range(warmup):
        w.step_replay()
    t_replay = timed(w.step_replay, steps, sync)
    st = eng.stats()
    # T_device: staged once, replayed K times
    w.stage()
    for _ in range(min(warmup, 3)):
        w.step_device()
    t_device = timed(eng.replay, steps, sync)
    st = eng.stats()
    phases = measure_phases(w, steps, sync)
    n_preds = int(eng.rows()["pred_num"].sum()) if want_rows else 0
    return {"t_replay_s": t_replay, "t_device_s": t_device, "stats": st, "phases": phases, "n_preds": n_preds}


def subline(eng, name, scale, seed, steps, warmup, sync, shuffled=False, deflate=False, cpu_budget_s=0.0):
    w = Workload(eng, name, scale, seed, shuffled=shuffled, deflate=deflate)
    r = run_workload(w, steps, warmup, sync, want_rows=False)
    st = r["stats"]
    A = algorithmic_bytes(st)
    whole = st.n_ops * A["A"] / (r["t_device_s"] / steps) / 1e9
    whole_r40 = st.n_ops * (A["A"] - A["R_op_record"] + 40.0) / (r["t_device_s"] / steps) / 1e9   # (R = 40)
    out = {"workload": w.describe(st), "n_changes": int(st.n_changes), "ops_per_s": st.n_ops * steps / r["t_replay_s"], "ms_per_step": r["t_replay_s"] / steps * 1e3,
           "t_device_ops_per_s": st.n_ops * steps / r["t_device_s"], "t_device_ms": r["t_device_s"] / steps * 1e3, "fast_path": int(st.fast_path),
           "phases_ms": r["phases"], "algorithmic_bytes_per_op": A,
           "roofline_whole_path": {"achieved": whole_r40, "unit": "GB/s", "frac": whole_r40 / 8000.0, "R53_rows_as_written": {"achieved": whole, "frac": whole / 8000.0}}}
    if w.is_doc and cpu_budget_s:
        # (Backend.load has a CPU leg of its own: the C port on this box on the same document, and the reference's recorded rate)
        out["cpu_baseline"] = cpu_baseline_document(w.doc_bytes, int(st.n_ops), budget_s=cpu_budget_s)
        rec = reference_js_recorded(name)
        if rec is not None:
            out["cpu_baseline"]["reference_js_recorded"] = rec
    return out


def apply_changes_section(eng, log, sync, reps=7):
    """SURVEY.md §8f-2 / 8f-4 (not part of `value`): Backend.applyChanges with its incremental patch on the engine -- binary changes in
    host memory -> am355_apply_changes (replay of the earlier changes + the batch, delta stage, setupPatches) -> patch record tables in
    host memory -- for the whole headline log onto an empty document and for batches onto the document the rest of the log made; and
    the sync protocol's Bloom filter over all its change hashes, built on the device."""
    from automerge_classic_amd.loggen import ChangeLog
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    changes = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    n = len(changes)
    rows = []
    for k in sorted({n, max(1, n // 10), max(1, n // 100), 1}, reverse=True):   # (the last one: a single change, the resident path of am355_apply_changes)
        # small batches as a host meets them: CALL AFTER CALL onto the growing document (the first call after a bulk load also builds the
        # host's indexes of the applied changes -- reported beside it); large ones: the one call onto the document the rest made
        calls = 8 if k <= max(1, n // 50) and n > 8 * k else 1
        first = n - calls * k
        base = ChangeLog.from_changes(changes[:first]) if first > 0 else None
        batches = [ChangeLog.from_changes(changes[first + j * k:first + (j + 1) * k]) for j in range(calls)]
        best, best_first, ops_before = None, None, 0
        for _ in range(reps if calls == 1 else max(3, reps // 2)):
            eng.reset()
            if base is not None:
                eng.apply_changes(base)
            times = []
            for j, batch in enumerate(batches):
                if j == calls - 1:
                    ops_before = int(eng.stats().n_ops)
                sync()
                t0 = time.perf_counter()
                eng.apply_changes(batch)
                times.append(time.perf_counter() - t0)
            later = sorted(times[1:]) if calls > 1 else times
            dt = later[len(later) // 2]
            best = dt if best is None else min(best, dt)
            best_first = times[0] if best_first is None else min(best_first, times[0])
        st = eng.stats()
        batch_ops = int(st.n_ops) - ops_before
        rows.append({"batch_changes": k, "batch_ops": batch_ops, "document_ops_before": ops_before, "ms": best * 1e3, "batch_ops_per_s": batch_ops / best,
                     "calls_in_a_row": calls, "first_call_ms": best_first * 1e3, "patch_bytes": len(eng.apply_patch_json())})
    idx = np.arange(n, dtype=np.uint32)
    eng.bloom_build(idx)
    t0 = time.perf_counter()
    for _ in range(reps):
        bits = eng.bloom_build(idx)
    bloom_ms = (time.perf_counter() - t0) / reps * 1e3
    return {"timed_region": "host change buffers -> am355_apply_changes -> incremental patch record tables in host memory (best of %d; batches with calls_in_a_row > 1: "
                            "that many consecutive calls onto the growing document, median of the calls after the first, first_call_ms beside it)" % reps,
            "parity": "patch text == oracle session == reference on the captured applyChanges calls (tests/test_apply_engine.py)",
            "batches": rows, "sync_bloom_filter": {"hashes": n, "filter_bytes": int(bits.size), "ms": bloom_ms}}


def sharded_measurement(eng, rank, world, dist, device, steps, warmup, barrier, scale=1.0, sync=None, name="c4_text_multi"):
    """ONE document sharded by objectId over the ranks: T_replay-style region (host buffers on every rank -> stitched
    patch IR on rank 0's host), max over ranks; beside it the same input unsharded on rank 0 alone.
    c4_text_multi: a change log of 64 Text objects -- every rank stages the batch, decodes the changes that touch its objects in
    full and the others as far as the object columns, merges its objects. c5_doc_mixed: Backend.load of a saved document -- a
    document's columns are run-length streams, every rank decodes and checks all rows and emits the records of its objects."""
    import hashlib
    import torch
    from automerge_classic_amd import loggen, shard
    is_doc = WORKLOADS[name][0] == "doc"
    if is_doc:
        doc_bytes, n_ops = loggen.document_config(scale)
        stage = lambda: eng.load_document(doc_bytes)
        what = f"{name} x{scale}: Backend.load of a {len(doc_bytes)}-byte saved document, {n_ops} op rows"
    else:
        log = make_log(name, scale, BASE_SEED[name])  # (the same log on every rank)
        stage = lambda: eng.load_changes(log)
        n_ops = int(log.n_ops)
        what = f"{name} x{scale}: {n_ops} ops, {log.n_changes} changes, 64 Text objects"
    if sync is None:
        sync = torch.cuda.synchronize
    # parity first, outside the timed region: stitched patch == unsharded patch
    eng.set_shard(0, 1)
    stage()
    eng.replay()
    want = hashlib.sha256(eng.patch_json().encode()).hexdigest() if rank == 0 else None
    sr = shard.ShardedReplay(eng, dist, device)
    have = sr.step(stage)
    same = (hashlib.sha256(eng.patch_json().encode()).hexdigest() == want) if have else True
    eng.set_shard(0, 1)
    dt, info = shard.bench_sharded(eng, stage, dist, device, steps, warmup, barrier)
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0])
    # the same input on one GPU (rank 0; the other ranks wait at the barrier)
    t_single = 0.0
    if rank == 0:
        def one():
            stage()
            eng.replay()
            eng.fetch_ir()
        for _ in range(warmup):
            one()
        t_single = timed(one, steps, sync)
    model = None
    if rank == 0 and not is_doc:
        # the critical-path model (sharding_model) from rank 0's own single-GPU phases, next to the measured figure
        eng.set_shard(0, 1)
        sub = subline(eng, name, scale, BASE_SEED[name], max(3, steps // 2), 2, sync)
        m = sharding_model(sub, n_list=(1, world))
        model = {"projected_speedup": m["projected"][-1]["projected_speedup"], "projected_ms_per_step": m["projected"][-1]["ms_per_step"],
                 "measured_on_one_gpu_ms": m["measured_on_one_gpu_ms"], "assumptions": m["assumptions"]}
    barrier()
    if rank != 0:
        return None
    return {"model": model, "workload": what + f", ONE document over {world} GPUs (objectId sharding: owner = (object counter + actor rank) mod N, _root on rank 0; "
                        "all_gather of the patch-IR fragments over RCCL, stitch on rank 0)",
            "scaling": "strong", "n_gpus": world, "steps": steps, "ops_per_s": n_ops * steps / dt, "ms_per_step": dt / steps * 1e3,
            "single_gpu_ops_per_s": n_ops * steps / t_single, "single_gpu_ms_per_step": t_single / steps * 1e3,
            "speedup_vs_single_gpu": t_single / dt, "fragment_bytes": info.get("fragment_bytes"),
            "parity": "stitched patch == unsharded patch (sha256 of the patch text)" if same else "MISMATCH"}


def sharding_model(sub, n_list=(1, 2, 4, 8)):
    """Critical-path model of the objectId-sharded replay of ONE change log over N GPUs, from the phases
    measured in THIS run on one GPU -- printed so that the first real multi-GPU run has something to be checked against (no 8-GPU node
    has been available to the driver so far). What the design replicates on every rank and what it divides:
      replicated  host staging + H2D of the whole batch (every rank has its own PCIe link), stage 1 (parse, actor tables, plan), the hash
                  stream (overlapped), the object columns of foreign changes in the decoder, the IR of the whole document to rank 0's host;
      divided     the full decode of the changes that touch the rank's objects, every merge / order / patch kernel (rows of own objects);
      added       one all_gather of the patch-IR fragments over xGMI (ring: (N-1)/N of the IR per link at ~153 GB/s, + ~40 us launch).
    A kernel does not get shorter than a launch of ~5 us: the ~20 kernels of the merge / order phases bound that part from below."""
    ph = sub["phases_ms"]
    t_dev, t_replay = sub["t_device_ms"], sub["ms_per_step"]
    host_side = max(t_replay - t_dev, 0.0)          # staging + IR to host (what of them the replay does not hide)
    parse, decode, merge_order, hash_stream = ph["ms_parse"], ph["ms_decode"], ph["ms_merge"] + ph["ms_order"], ph["ms_hash_stream"]
    gaps = max(t_dev - parse - decode - merge_order, 0.0)
    f_obj = 0.35                                     # decoder time up to the object columns (5 of 12 columns: action, ids, insert, object)
    floor = 20 * 0.005                               # 20 launches of >= 5 us
    ir_bytes = sub["algorithmic_bytes_per_op"]["P_patch_ir"] * sub["t_device_ops_per_s"] * t_dev * 1e-3
    rows = []
    n_changes = int(sub.get("n_changes") or 4097)
    for n in n_list:
        dec = decode * (1.0 / n + f_obj * (1.0 - 1.0 / n))
        mo = max(merge_order / n, min(floor, merge_order))
        main_chain = parse + dec + mo + gaps
        gather = 0.0 if n == 1 else 0.04 + ir_bytes * (n - 1) / n / 153e9 * 1e3
        t = host_side + max(main_chain, hash_stream) + gather
        # SURVEY.md §8e(i) on top (VERDICT r5 next #6b, modelled, not built): parse + SHA-256 sharded by change index, one ncclAllGather of the
        # ChangeMetas (176 B per change) in the main chain and one of the digests (32 B) in the hash stream, ~25 us of latency each
        ag1 = 0.0 if n == 1 else 0.025 + 176.0 * n_changes * (n - 1) / n / 153e9 * 1e3
        ag2 = 0.0 if n == 1 else 0.025 + 32.0 * n_changes * (n - 1) / n / 153e9 * 1e3
        chain_b = parse / n + ag1 + dec + mo + gaps
        t_b = host_side + max(chain_b, hash_stream / n + ag2) + gather
        rows.append({"n_gpus": n, "ms_per_step": t, "device_chain_ms": main_chain, "all_gather_ms": gather,
                     "with_stage1_sharded": {"ms_per_step": t_b, "device_chain_ms": chain_b, "parse_ms": parse / n, "metas_all_gather_ms": ag1}})
    t1 = rows[0]["ms_per_step"]
    for r in rows:
        r["projected_speedup"] = t1 / r["ms_per_step"]
        r["with_stage1_sharded"]["projected_speedup"] = t1 / r["with_stage1_sharded"]["ms_per_step"]
    return {"workload": sub["workload"], "measured_on_one_gpu_ms": {"t_replay": t_replay, "t_device": t_dev, "parse": parse, "decode": decode, "merge_order": merge_order,
                                                                   "hash_stream": hash_stream, "host_side": host_side},
            "assumptions": {"decoder_fraction_for_foreign_changes": f_obj, "launch_floor_ms": floor, "xgmi_link_GB_per_s": 153, "all_gather_launch_ms": 0.04},
            "projected": rows,
            "stage1_sharding": "modelled, not built: parse / N + an all-gather of the ChangeMetas costs what it saves when a change parses in ~5 ns (21 us for 4097 changes) "
                               "and an all-gather starts at ~25 us; it pays for batches of few fat changes (c3_map_lww: parse 95 us, hash stream 380 us)",
            "note": "strong scaling of ONE document; the deployment shape (one document per GPU, `value` at N > 1) scales with N by construction"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c4_text_single", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--prewarm", type=float, default=2.0, help="seconds of untimed steps before the warm-up steps (clocks, PCIe link, host threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sublines", action="store_true")
    ap.add_argument("--no-live-trace", action="store_true", help="do not run the rocprofv3-traced child that fills roofline.kernels (the committed table is embedded instead)")
    ap.add_argument("--no-shard", action="store_true", help="N > 1: skip the objectId-sharded measurement that follows the replica measurement")
    ap.add_argument("--subline-scale", type=float, default=1.0, help="scale of the sub-workloads (tests run the line builder at a small one)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"), help="where the full record goes (the stdout line is the compact one)")
    return ap.parse_args(argv)


def run(args, eng, rank, world, dist, device, barrier, sync):
    """Everything bench.py measures, as ONE record (rank 0; None on the other ranks). main() prints its compact form (compact_line) on
    stdout and the whole of it to --detail and stderr."""
    import hashlib
    from automerge_classic_amd import dist_util
    w = Workload(eng, args.workload, args.scale, dist_util.rank_seed(BASE_SEED[args.workload], rank))
    # ---- the in-run parity gate (before any timing counts): the engine's getPatch text of the workload about to be
    # timed == the CPU oracle's on the same bytes (sha256 of the text). N = 1 only (the oracle leg runs on rank 0 at N = 1).
    parity_in_run = "skipped (--no-cpu-baseline)" if args.no_cpu_baseline else "skipped (N > 1: the oracle leg runs at N = 1 only)" if world > 1 else None
    if parity_in_run is None:
        w.step_replay()
        got = hashlib.sha256(eng.patch_json().encode()).hexdigest()
        want = oracle_document_patch_sha256(w.doc_bytes) if w.is_doc else oracle_patch_sha256(w.log)
        if got != want:
            raise SystemExit(f"PARITY GATE FAILED: sha256(engine getPatch) {got} != sha256(oracle getPatch) {want} on {args.workload} x{args.scale}: no timing is reported")
        parity_in_run = True
    # untimed: a fresh box needs a moment of load before it runs at its steady rate (on two of five boxes of the pool the first
    # ~0.2 s of steps -- the PCIe-heavy staging part -- ran 30 % slower than everything after); then the W warm-up steps of the contract
    t_pre, pre = time.perf_counter(), []
    while time.perf_counter() - t_pre < args.prewarm:
        t0 = time.perf_counter()
        w.step_replay()
        pre.append(time.perf_counter() - t0)
    if pre and rank == 0:
        k = max(1, min(100, len(pre) // 4))
        print(f"pre-warm: {len(pre)} steps, first {k}: {sum(pre[:k]) / k * 1e3:.3f} ms/step, last {k}: {sum(pre[-k:]) / k * 1e3:.3f} ms/step", file=sys.stderr)
    for _ in range(args.warmup):
        w.step_replay()
    # ---- the timed region of `value`: K steps of T_replay (host buffers in -> patch IR in host memory) ----
    elapsed = timed(w.step_replay, args.steps, barrier)
    st = eng.stats()
    elapsed, total_ops = dist_util.aggregate(elapsed, float(st.n_ops) * args.steps, dist, device)
    # ---- T_device (not `value`): the replay alone, inputs resident in HBM ----
    w.stage()
    for _ in range(3):
        w.step_device()
    t_dev = timed(eng.replay, args.steps, barrier)
    t_dev, _ = dist_util.aggregate(t_dev, 0.0, dist, device)
    phases = measure_phases(w, args.steps, sync)  # (its own context, HIP events between the phases: not timed)
    sharded = sharded_c5 = None
    if world > 1 and not args.no_shard:
        sharded = sharded_measurement(eng, rank, world, dist, device, max(5, min(args.steps // 2, 30)), 3, barrier)
        # BASELINE config 5 (the other 8-GPU configuration): one saved document over the N ranks
        sharded_c5 = sharded_measurement(eng, rank, world, dist, device, 3, 1, barrier, name="c5_doc_mixed")
    if rank != 0:
        return None
    st = eng.stats()
    value = total_ops / elapsed
    t_device_ms = t_dev / args.steps * 1e3
    n_preds = int(eng.rows()["pred_num"].sum())
    A = algorithmic_bytes(st)
    whole = st.n_ops * A["A"] / (t_device_ms * 1e-3) / 1e9  # (N_ops x A / T_device), GB/s
    # SURVEY.md §8d prices the op record at R = 40 B (10 x u32); the rows this engine writes are 53 B (13 SoA fields + the insert flag)
    whole_r40 = st.n_ops * (A["A"] - A["R_op_record"] + 40.0) / (t_device_ms * 1e-3) / 1e9
    # (`achieved` / `frac`: SURVEY.md §8d's definition, R = 40 B; the 53 B this engine's rows take are in record_bytes beside it)
    roofline = {"bound": "hbm", "achieved": whole_r40, "peak": 8000.0, "unit": "GB/s", "frac": whole_r40 / 8000.0, "traffic": None,
                "record_bytes": {"R53_rows_as_written": {"achieved": whole, "frac": whole / 8000.0},
                                 "R40_survey_figure": {"achieved": whole_r40, "frac": whole_r40 / 8000.0}},
                "kernel": "whole path (SURVEY.md §8d): N_ops x (E + R + P) / T_device, R = 40", "algorithmic_bytes_per_launch": st.n_ops * (A["A"] - A["R_op_record"] + 40.0),
                "launch_ms": t_device_ms, "n_preds": n_preds, "phases": phase_table(phases, st, n_preds)}
    import glob
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_table.json")))  # (the latest round's summary)
    table = tables[-1] if tables else os.path.join(ROOT, "profiles", "r02_kernel_table.json")
    headline = args.workload == "c4_text_single" and args.scale == 1.0
    live = live_kernel_table() if (headline and world == 1 and not args.no_live_trace) else None
    if live is not None:
        roofline["kernels"] = live.get("kernels")
        roofline["traffic"] = live.get("traffic_bytes_per_replay") or None
        roofline["kernels_source"] = live.get("source")
        roofline["kernels_live"] = True
        roofline["kernels_trace_seconds"] = live.get("seconds")
        roofline["t_device_ms_under_trace"] = live.get("t_device_ms_under_trace")
    elif os.path.exists(table) and headline:
        roofline["kernels_live"] = False
        # rocprofv3 kernel-trace + PMC passes of this same command (committed summary; counters cannot be read in-process):
        # per kernel: calls per replay, average us, algorithmic bytes, PMC HBM bytes, fraction of the 8 TB/s peak
        with open(table) as f:
            t = json.load(f)
        roofline["kernels"] = t.get("kernels")
        roofline["traffic"] = t.get("traffic_bytes_per_replay")
        roofline["kernels_source"] = t.get("source")
    save_info = None
    if not w.is_doc:
        eng.save()
        t0 = time.perf_counter()
        saved = eng.save()
        save_info = {"ms": (time.perf_counter() - t0) * 1e3, "doc_bytes": len(saved)}
    out = {
        "metric": "CRDT ops/sec applied (bulk replay)", "value": value, "unit": "ops/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": w.describe(st) + "; one document per GPU",
                   "timed_region": "T_replay (SURVEY.md §8d): binary changes in host memory -> host inflate/staging -> H2D -> replay -> patch IR + envelope in host memory",
                   "parity": PARITY, "parity_checked_in_run": parity_in_run, "fast_path": int(st.fast_path),
                   "untimed_before_the_K_steps": f"{args.prewarm} s of steps + {args.warmup} warm-up steps"},
        "t_device_ops_per_s": st.n_ops / (t_device_ms * 1e-3), "t_device_ms": t_device_ms,
        "phases_ms": phases, "algorithmic_bytes_per_op": A, "n_list_elems": int(st.n_list_elems), "save": save_info, "roofline": roofline,
    }
    if sharded is not None:
        out["sharded"] = sharded
    if sharded_c5 is not None:
        out["sharded_c5"] = sharded_c5
    if world > 1 and not args.no_shard:
        # (rank 0 only; the other ranks have left run() and idle in main() until rank 0 is through: their GPUs are free for the workers)
        eng.set_shard(0, 1)
        mlog = make_log("c4_text_multi", 1.0, BASE_SEED["c4_text_multi"])
        eng.load_changes(mlog)
        eng.replay()
        sj = sharded_js("c4_text_multi", 1.0, world, want_sha=hashlib.sha256(eng.patch_json().encode()).hexdigest())
        if sj is not None:
            out["sharded_js"] = sj
    if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
        if w.is_doc:
            out["cpu_baseline"] = cpu_baseline_document(w.doc_bytes, int(st.n_ops))
            rec = reference_js_recorded(args.workload)
            if rec is not None:
                out["cpu_baseline"]["reference_js_recorded"] = rec
        else:
            port = cpu_baseline(w.log)
            ref_js = reference_js_baseline(args.workload, args.scale, BASE_SEED[args.workload])
            if ref_js is not None:   # the reference itself ran beside the engine in this run: that is the baseline; the C port rides along
                ref_js["port"] = {k: port[k] for k in ("value", "unit", "cores", "sample")}
                ref_js["leg"] = "reference JS backend (AUTOMERGE_REF resolved)"
                out["cpu_baseline"] = ref_js
            else:
                port["leg"] = "C port of the reference's algorithm (oracle/): the reference tree is not on this box; its recorded rate is beside it"
                rec = reference_js_recorded(args.workload)
                if rec is not None:
                    port["reference_js_recorded"] = rec
                out["cpu_baseline"] = port
    if world == 1 and not w.is_doc and not args.no_sublines:
        e2e = js_end_to_end(w.log)
        if e2e is not None:
            out["js_end_to_end"] = e2e
            if "t_e2e_ms" in e2e:
                out["t_e2e_ms"] = e2e["t_e2e_ms"]
        ja = js_apply_latency(w.log)
        if ja is not None:
            out["js_apply_changes"] = ja
    if not args.no_sublines and world == 1:
        ss = args.subline_scale
        subs, k, wu = [], max(5, min(args.steps // 3, 30)), 3
        for name in ("c4_text_multi", "c3_map_lww", "c2_text_typing"):
            if name != args.workload:
                subs.append(subline(eng, name, ss, BASE_SEED[name], k, wu, barrier))
        subs.append(subline(eng, "c4_text_single", ss, BASE_SEED["c4_text_single"], k, wu, barrier, shuffled=True))
        # the same log with every change DEFLATEd as the reference's encodeChange does for changes >= 256 bytes (columnar.js:798-811):
        # T_replay then includes the host inflate (zlib, on the engine's host threads)
        subs.append(subline(eng, "c4_text_single", ss, BASE_SEED["c4_text_single"], k, wu, barrier, deflate=True))
        # a larger batch of the headline shape (4 x: 4.1 M ops, 54 MB of changes): the same kernels with more work per launch
        subs.append(subline(eng, "c4_text_single", 4.0 * ss, BASE_SEED["c4_text_single"], min(8, k), 2, barrier))
        if args.workload != "c5_doc_mixed":
            subs.append(subline(eng, "c5_doc_mixed", ss, BASE_SEED["c5_doc_mixed"], min(5, k), 2, barrier, cpu_budget_s=0.0 if args.no_cpu_baseline else 8.0))
        out["workloads"] = subs
        multi = next((x for x in subs if x["workload"].startswith("c4_text_multi")), None)
        if multi is not None:
            out["sharding_model"] = sharding_model(multi)
        if save_info is not None:
            # the history of the saved headline document after Backend.load (binary changes + hashes rebuilt:
            # op columns decoded on the GPU, regroup / re-encode / hash chain on the host threads); not part of `value`
            # (best of 5 after one untimed call, like every other timing of this line: the first call allocates the stage's buffers)
            def history_ms(deflate):
                best, off = None, None
                for i in range(6):
                    eng.load_document(saved)
                    eng.replay()
                    t0 = time.perf_counter()
                    _, off, _ = eng.doc_changes(deflate=deflate)
                    dt = (time.perf_counter() - t0) * 1e3
                    if i:
                        best = dt if best is None else min(best, dt)
                return best, off
            ms_plain, h_off = history_ms(False)
            ms_deflate, _ = history_ms(True)
            out["history_after_load"] = {"n_changes": int(len(h_off) - 1), "change_bytes": int(h_off[-1]), "ms": ms_plain, "ms_with_deflate": ms_deflate,
                                         "ops_per_s": st.n_ops / (ms_plain * 1e-3),
                                         "timed_region": "am355_doc_changes of the loaded document: binary changes + hashes in host memory (best of 5 after one untimed call)"}
        if not w.is_doc:
            out["apply_changes"] = apply_changes_section(eng, w.log, barrier)
        # the sharded path as the JS host reaches it, on this box's one GPU: a communicator of one rank through the real librccl
        eng.set_shard(0, 1)
        mlog = make_log("c4_text_multi", args.subline_scale, BASE_SEED["c4_text_multi"])
        eng.load_changes(mlog)
        eng.replay()
        sj = sharded_js("c4_text_multi", args.subline_scale, 1, want_sha=hashlib.sha256(eng.patch_json().encode()).hexdigest())
        if sj is not None:
            out["sharded_js"] = sj
    return out


def _r(x, digits=4):
    """Numbers of the compact line: 4 significant digits are what a reader compares; NaN / inf never reach the line."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    x = float(x)
    if x != x or x in (float("inf"), float("-inf")):
        return None
    if x == 0.0:
        return 0.0
    from math import floor, log10
    return round(x, digits - 1 - int(floor(log10(abs(x)))))


def compact_line(d):
    """The ONE stdout line (<= LINE_LIMIT bytes): the contract keys, the whole-path roofline with the dominant kernel, the CPU baseline,
    one short row per sub-workload. Everything else of run()'s record stays in --detail."""
    cfg = d["config"]
    line = {k: d[k] for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value"] = _r(d["value"], 6)
    line["ms_per_step"] = _r(d["ms_per_step"], 5)
    line["config"] = {"workload": cfg["workload"][:300], "timed_region": cfg["timed_region"][:200], "parity": cfg["parity"][:200],
                      "parity_checked_in_run": cfg["parity_checked_in_run"], "fast_path": cfg["fast_path"]}
    line["t_device_ms"] = _r(d["t_device_ms"], 5)
    line["algorithmic_bytes_per_op"] = {k: _r(v) for k, v in d["algorithmic_bytes_per_op"].items()}
    rf = d["roofline"]
    roof = {"bound": rf["bound"], "achieved": _r(rf["achieved"], 5), "peak": rf["peak"], "unit": rf["unit"], "frac": _r(rf["frac"], 4),
            "traffic": rf.get("traffic"), "kernel": rf["kernel"], "launch_ms": _r(rf["launch_ms"], 5),
            "algorithmic_bytes_per_launch": int(rf["algorithmic_bytes_per_launch"])}
    ks = [k for k in (rf.get("kernels") or []) if k.get("avg_us")]
    if ks:
        dom = max(ks, key=lambda k: k.get("pct_of_kernel_time") or 0.0)
        roof["dominant_kernel"] = {"name": str(dom.get("kernel"))[:60], "avg_us": _r(dom.get("avg_us")), "pct_of_kernel_time": _r(dom.get("pct_of_kernel_time")),
                                   "algorithmic_bytes": dom.get("algorithmic_bytes"), "frac": _r(dom.get("frac_of_hbm_peak")),
                                   "pmc_traffic_bytes": dom.get("pmc_traffic_bytes"), "live": bool(rf.get("kernels_live"))}
        if rf.get("traffic"):
            roof["traffic_over_algorithmic"] = _r(rf["traffic"] / max(rf["algorithmic_bytes_per_launch"], 1.0))
    line["roofline"] = roof
    cb = d.get("cpu_baseline")
    if cb:
        c = {k: (_r(cb[k], 5) if k == "value" else cb[k]) for k in ("value", "unit", "cores", "kind") if k in cb}
        c["sample"] = str(cb.get("sample", ""))[:220]
        if "port" in cb:
            c["port"] = {"value": _r(cb["port"]["value"], 5), "cores": cb["port"]["cores"]}
        rec = cb.get("reference_js_recorded")
        if rec:
            c["reference_js_recorded"] = {"value": _r(rec["value"], 5), "file": rec["file"], "recorded": rec.get("recorded"),
                                          "why": "the reference tree cannot travel to the GPU box: its rate recorded in the build container by tools/record_reference_js.py"}
        line["cpu_baseline"] = c
    rows = []
    for x in d.get("workloads") or []:
        row = {"name": x["workload"].split(":")[0], "ops_per_s": _r(x["ops_per_s"], 5), "ms_per_step": _r(x["ms_per_step"], 5),
               "t_device_ms": _r(x["t_device_ms"], 5), "frac": _r(x["roofline_whole_path"]["frac"], 4)}
        rows.append(row)
    if rows:
        line["workloads"] = rows
    if d.get("js_apply_changes") and "ms_per_call" in d["js_apply_changes"]:
        line["js_apply_changes_ms_per_call"] = _r(d["js_apply_changes"]["ms_per_call"])
    if d.get("apply_changes"):
        line["apply_changes_ms"] = [[b["batch_changes"], _r(b["ms"])] for b in d["apply_changes"]["batches"]]
    for k in ("t_e2e_ms",):
        if d.get(k) is not None:
            line[k] = _r(d[k])
    if d.get("save"):
        line["save_ms"] = _r(d["save"]["ms"])
    if d.get("history_after_load"):
        line["history_after_load_ms"] = _r(d["history_after_load"]["ms"])
    for k in ("sharded", "sharded_c5"):
        if d.get(k):
            x = d[k]
            line[k] = {"n_gpus": x["n_gpus"], "scaling": x["scaling"], "ops_per_s": _r(x["ops_per_s"], 5), "ms_per_step": _r(x["ms_per_step"], 5),
                       "single_gpu_ms_per_step": _r(x["single_gpu_ms_per_step"], 5), "speedup_vs_single_gpu": _r(x["speedup_vs_single_gpu"]),
                       "parity": x["parity"][:80]}
    if d.get("sharded_js"):
        x = d["sharded_js"]
        line["sharded_js"] = {k: (_r(x[k]) if isinstance(x[k], float) else x[k]) for k in ("n_gpus", "engine_ms_per_step", "single_worker_ms_per_step", "speedup_vs_single_worker", "parity", "error") if k in x}
        if "error" in line["sharded_js"]:
            line["sharded_js"]["error"] = str(line["sharded_js"]["error"])[:120]
    if d.get("sharding_model"):
        line["sharding_model_projected_speedup"] = {str(r["n_gpus"]): [_r(r["projected_speedup"]), _r(r["with_stage1_sharded"]["projected_speedup"])] for r in d["sharding_model"]["projected"]}
    line["detail"] = d.get("detail_file", "gpurun_out/bench_detail.json")
    # the line must fit whole in the driver's record: drop the optional parts, last first, until it does
    for k in ("sharding_model_projected_speedup", "history_after_load_ms", "save_ms", "js_apply_changes_ms_per_call", "apply_changes_ms", "sharded_js", "sharded_c5", "workloads"):
        if len(json.dumps(line, allow_nan=False)) < LINE_LIMIT - 64:
            break
        line.pop(k, None)
    return line


def _finite(x):
    """run()'s record with every NaN / inf replaced by None (strict JSON on both outputs)."""
    if isinstance(x, float):
        return x if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, np.generic):
        return _finite(x.item())
    return x


def emit(detail, path):
    """stderr + --detail: the whole record. stdout: the compact line, the LAST thing printed."""
    detail = _finite(detail)
    detail["detail_file"] = os.path.relpath(path, ROOT) if path else None
    text = json.dumps(detail, allow_nan=False)
    if path:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(text + "\n")
        except OSError as e:   # (a read-only checkout must not cost the line)
            print(f"bench detail not written: {e}", file=sys.stderr)
    print("bench detail: " + text, file=sys.stderr, flush=True)
    line = json.dumps(compact_line(detail), allow_nan=False)
    assert len(line) < LINE_LIMIT, len(line)
    print(line, flush=True)
    return line


def main():
    args = parse_args()
    import torch
    from automerge_classic_amd import dist_util, engine
    rank, world, local_rank = dist_util.rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    eng = engine.Engine(local_rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    detail = run(args, eng, rank, world, dist, torch.device("cuda", local_rank), barrier, torch.cuda.synchronize)
    if detail is not None:
        emit(detail, args.detail)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
