"""Parity tests proper: the HIP engine (through the C ABI, on a real MI355X) against the CPU oracle and the
golden patches of the unmodified reference. Bit-exact: all of this is integer / byte work."""
import hashlib
import json
import os

import numpy as np
import pytest

import golden_util
import mutation_util
import oracle_lib
from automerge_classic_amd import engine, loggen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(0)
    yield e
    e.close()


def gpu_patch(eng, log):
    eng.load_changes(log)
    eng.replay()
    return eng.patch_json()


@pytest.mark.parametrize("name", golden_util.fixture_names())
def test_golden_reference_patches(eng, name):
    fx = golden_util.load_fixture(name)
    assert gpu_patch(eng, fx["log"]) == fx["expected"]


@pytest.mark.parametrize("name", golden_util.doc_fixture_names())
def test_document_load_matches_reference(eng, name):
    """Backend.load(bytes) + getPatch against the unmodified reference's save()/load()."""
    fx = golden_util.load_fixture(name)
    if "doc_bytes" not in fx:
        pytest.skip("no document fixture")
    eng.load_document(fx["doc_bytes"])
    eng.replay()
    assert eng.patch_json() == fx["expected_load"]
    eng.backend_load(fx["doc_bytes"])   # (Backend.load in one call: am355_backend_load)
    assert eng.patch_json() == fx["expected_load"]
    assert oracle_lib.OracleDoc.load_document(fx["doc_bytes"]).patch_json() == fx["expected_load"]


def test_counters_and_valueless_rows_inside_lists(eng):
    """new.js:937-965, 1010-1018, 1026-1033 in whole-document patches (refused through round 4): patch, save, load, history of every case
    of tests/golden/list_quirks.json equal the unmodified reference's; exactly one case is left to the JS path."""
    served, refused = golden_util.check_list_quirk_cases(eng, engine)
    assert set(refused) == golden_util.LIST_QUIRK_REFUSED and served >= 47


def test_defect_fixture_both_delivery_orders(eng):
    """Inputs on which the STOCK reference's block-boundary defect fires (its two patches for the two delivery orders differ
    from each other): the engine gives the block-size-patched reference's document for both orders, also on the
    12,801-op slice of the headline workload (reference digests committed by oracle/make_defect_fixture.py)."""
    fx = golden_util.defect_fixture()
    assert json.loads(fx["patch"])["diffs"] != json.loads(fx["patch_reversed"])["diffs"]
    assert gpu_patch(eng, fx["log"]) == fx["patch_bigblock"] != fx["patch"]
    assert gpu_patch(eng, fx["log_reversed"]) == fx["patch_bigblock_reversed"]
    big = fx["larger"]
    log = loggen.config(big["workload"], big["scale"])
    assert hashlib.sha256(gpu_patch(eng, log).encode()).hexdigest() == big["patch_sha256"]["bigblock"] != big["patch_sha256"]["stock"]
    rlog = log.reordered([0] + list(range(64, 0, -1)) + list(range(65, log.n_changes)))
    assert hashlib.sha256(gpu_patch(eng, rlog).encode()).hexdigest() == big["patch_sha256"]["bigblock_reversed"]


def test_device_primitives(eng):
    rng = np.random.default_rng(7)
    for n in (1, 63, 2048, 2049, 8192, 8193, 100_003, 1_500_000, 5_000_011, 41_000_003):  # one launch / two launches / three launches of the scan
        vals = rng.integers(0, 9, n, dtype=np.uint32)
        out, total = eng.test_scan(vals)
        ref = np.concatenate(([0], np.cumsum(vals.astype(np.uint64))[:-1])).astype(np.uint32)
        assert np.array_equal(out, ref) and total == int(vals.sum())
        if n > 2_200_000:   # (the single-pass form with decoupled look-back: off by default, kept correct)
            os.environ["AM355_SCAN_LOOKBACK"] = "1"
            try:
                out, total = eng.test_scan(vals)
            finally:
                del os.environ["AM355_SCAN_LOOKBACK"]
            assert np.array_equal(out, ref) and total == int(vals.sum())
        keys = rng.integers(0, 1 << 40, n, dtype=np.uint64)
        k, v = eng.test_sort(keys, np.arange(n, dtype=np.uint32), 40)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(k, keys[order]) and np.array_equal(v, order.astype(np.uint32))


def test_change_hashes_match_sha256(eng):
    log = loggen.config("c4_text_multi", 0.05)
    eng.load_changes(log)
    eng.replay()
    h = eng.hashes()
    for i in range(0, log.n_changes, 7):
        c = log.change(i)
        assert bytes(h[i]) == hashlib.sha256(c[8:]).digest()
        assert bytes(h[i][:4]) == c[4:8]


CASES = [
    ("c2_text_typing", 0.2, False),
    ("c2_text_typing", 0.05, True),
    ("c3_map_lww", 0.5, False),
    ("c4_text_single", 0.1, False),
    ("c4_text_single", 0.1, True),
    ("c4_text_multi", 0.1, False),
]


def test_many_children_of_one_element_use_the_radix_path(eng):
    # 1000 actors all inserting at the head in the same round: one parent with 1000 children
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=1000, n_rounds=2, ins_per_change=3, del_per_change=1, n_objects=1, seed=5)
    assert gpu_patch(eng, log) == oracle_lib.OracleDoc(log).patch_json()


def test_oversized_columns_use_the_serial_decoder(eng):
    # 2000 deletes per change: pred / key columns exceed the wave decoder's LDS staging and take the lane-serial kernel
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=4, n_rounds=3, ins_per_change=3000, del_per_change=2000, n_objects=1, seed=9)
    assert gpu_patch(eng, log) == oracle_lib.OracleDoc(log).patch_json()


@pytest.mark.parametrize("name,scale,deflate", CASES)
def test_generated_workloads_match_oracle(eng, name, scale, deflate):
    log = loggen.config(name, scale, deflate=deflate)
    got = gpu_patch(eng, log)
    want = oracle_lib.OracleDoc(log).patch_json()
    assert got == want
    st = eng.stats()
    assert st.n_ops == log.n_ops and st.n_pending == 0 and st.n_applied == log.n_changes


def test_delivery_order_only_changes_clock_order(eng):
    log = loggen.config("c4_text_multi", 0.05)
    base = json.loads(gpu_patch(eng, log))
    perm = np.random.default_rng(3).permutation(log.n_changes)
    shuf = log.reordered(perm)
    got = gpu_patch(eng, shuf)
    assert got == oracle_lib.OracleDoc(shuf).patch_json()
    p = json.loads(got)
    assert p["diffs"] == base["diffs"] and p["deps"] == base["deps"] and p["maxOp"] == base["maxOp"] and p["clock"] == base["clock"]


def _scheduler_cases():
    from test_engine_emu import scheduler_cases
    return list(scheduler_cases())


@pytest.mark.parametrize("name,log", _scheduler_cases(), ids=[n for n, _ in _scheduler_cases()])
def test_general_scheduler_on_the_device_equals_host_and_oracle(eng, name, log):
    """SURVEY 8 row a12 on the GPU: ks_pass (register / LDS form and global-memory form), the host's restatement and the oracle agree on
    shuffled, reversed, duplicated and incomplete deliveries; two sweeps are not enough for reversed chains and the host takes over."""
    from test_engine_emu import check_scheduler_variants
    seen = check_scheduler_variants(eng, name, log)
    if name in ("reversed", "chain_reversed"):
        assert seen["device_out_of_sweeps"][0] == 0


def test_full_size_headline_log_in_shuffled_delivery(eng):
    """The 1,020,801-op headline log delivered in random order: scheduled on the device (fast_path 2), same patch as the oracle, same
    document as the in-order delivery."""
    log = loggen.config("c4_text_single", 1.0)
    shuf = log.reordered(np.random.default_rng(4).permutation(log.n_changes))
    got = gpu_patch(eng, shuf)
    st = eng.stats()
    assert st.fast_path == 2 and st.n_pending == 0 and st.n_applied == log.n_changes
    assert got == oracle_lib.OracleDoc(shuf).patch_json()
    assert json.loads(got)["diffs"] == json.loads(gpu_patch(eng, log))["diffs"]


def test_missing_dependency_leaves_changes_pending(eng):
    log = loggen.config("c4_text_single", 0.05)
    keep = [i for i in range(log.n_changes) if i != 1]  # drop one change of round 0: everything after it must wait
    part = log.reordered(keep)
    got = gpu_patch(eng, part)
    assert got == oracle_lib.OracleDoc(part).patch_json()
    assert json.loads(got)["pendingChanges"] > 0


def test_corrupt_input_is_rejected(eng):
    log = loggen.config("c2_text_typing", 0.01)
    arena = log.arena.copy()
    arena[int(log.offsets[1]) + 20] ^= 0x55  # flip a byte inside the second change
    bad = loggen.ChangeLog(arena, log.offsets, log.n_ops)
    eng.load_changes(bad)
    with pytest.raises(engine.InvalidChanges) as ei:
        eng.replay()
    assert "BAD_CHECKSUM" in ei.value.flag_names
    with pytest.raises(oracle_lib.OracleError):
        oracle_lib.OracleDoc(bad)


@pytest.mark.parametrize("name,scale", [("c2_text_typing", 1.0), ("c3_map_lww", 1.0), ("c4_text_multi", 1.0)])
def test_full_size_other_configs(eng, name, scale):
    """BASELINE configs 2 and 3 and the shardable variant of config 4 at full size, bit-exact against the oracle."""
    log = loggen.config(name, scale)
    got = gpu_patch(eng, log)
    want = oracle_lib.OracleDoc(log).patch_json()
    assert hashlib.sha256(got.encode()).hexdigest() == hashlib.sha256(want.encode()).hexdigest()
    assert eng.stats().n_ops == log.n_ops


def test_full_size_headline_workload(eng):
    """BASELINE config 4 at full size (1M ops, 64 actors, one Text object): bit-exact against the oracle, plus
    size-independent properties of the patch."""
    log = loggen.config("c4_text_single", 1.0)
    got = gpu_patch(eng, log)
    st = eng.stats()
    assert st.n_ops == log.n_ops > 1_000_000
    want = oracle_lib.OracleDoc(log).patch_json()
    assert hashlib.sha256(got.encode()).hexdigest() == hashlib.sha256(want.encode()).hexdigest()
    p = json.loads(got)
    text = next(iter(p["diffs"]["props"]["text"].values()))
    n_vis, idx = 0, 0
    for e in text["edits"]:
        assert e["index"] == idx  # edits tile the index space with no gaps: visible elements are numbered densely
        k = len(e["values"]) if e["action"] == "multi-insert" else 1
        idx += k
        n_vis += k
    # every insert is either visible or deleted; the generator only deletes visible elements, but concurrent actors may delete
    # the same element twice in one round, so the deletes bound the removed elements from above
    n_ins, n_del = int(st.n_list_elems), log.n_ops - 1 - int(st.n_list_elems)
    assert n_ins - n_del <= n_vis <= n_ins and n_vis > 0


def test_long_tour_uses_the_pointer_jumping_rounds(eng):
    """25,600 single-character inserts at random positions by 64 actors: every element is its own typing run, so the Euler tour
    (2 x runs + 1 entries) does not fit the single-workgroup LDS list ranking and takes the rounds in HBM."""
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=64, n_rounds=400, ins_per_change=1, del_per_change=0, n_objects=1, seed=21)
    assert gpu_patch(eng, log) == oracle_lib.OracleDoc(log).patch_json()
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=64, n_rounds=200, ins_per_change=2, del_per_change=1, n_objects=7, seed=22)
    assert gpu_patch(eng, log) == oracle_lib.OracleDoc(log).patch_json()


@pytest.mark.parametrize("deflate", [False, True])
def test_generated_document_matches_oracle(eng, deflate):
    """Config-5 shaped saved document (text / nested maps / lists), ~120 k rows, against the oracle's Backend.load."""
    doc, rows = loggen.document_config(0.01, deflate=deflate)
    eng.load_document(doc)
    eng.replay()
    assert eng.stats().n_ops == rows
    assert eng.patch_json() == oracle_lib.OracleDoc.load_document(doc).patch_json()


def test_document_columns_parallel_equals_serial(eng, monkeypatch):
    """The parallel big-column decoder and the lane-serial decoders produce identical op rows."""
    doc, rows = loggen.generate_document(n_actors=9, n_texts=5, text_len=2500, n_maps=4, keys_per_map=400, n_submaps=3, n_lists=5, list_len=1500,
                                         deflate=False, seed=0xD0C7)
    got = []
    for serial in ("0", "1"):
        monkeypatch.setenv("AM355_DOC_SERIAL", serial)
        eng.load_document(doc)
        eng.replay()
        got.append((eng.rows(), eng.patch_json()))
    monkeypatch.delenv("AM355_DOC_SERIAL")
    assert got[0][1] == got[1][1]
    for k in got[0][0]:
        assert np.array_equal(got[0][0][k], got[1][0][k]), k


@pytest.mark.parametrize("scale", [0.1, 1.0])
def test_full_size_document(eng, scale):
    """BASELINE config 5 at 1/10 scale (1.2 M rows) and at the benchmarked size (12.0 M rows, 101 MB of inflated op columns):
    Backend.load + getPatch bit-exact against the oracle (digest of the patch text at full size)."""
    doc, rows = loggen.document_config(scale)
    eng.load_document(doc)
    eng.replay()
    assert eng.stats().n_ops == rows
    got, want = eng.patch_json(), oracle_lib.OracleDoc.load_document(doc).patch_json()
    assert len(got) == len(want) and hashlib.sha256(got.encode()).hexdigest() == hashlib.sha256(want.encode()).hexdigest()
    if scale >= 1.0:
        assert rows > 10_000_000


@pytest.mark.parametrize("name", golden_util.fixture_names())
def test_save_after_replay_is_byte_identical_to_the_reference(eng, name):
    """Backend.save(Backend.loadChanges(Backend.init(), changes)) against the unmodified reference's bytes."""
    fx = golden_util.load_fixture(name)
    eng.load_changes(fx["log"])
    eng.replay()
    if "doc_bytes" not in fx:
        with pytest.raises(engine.UnsupportedChanges):
            eng.save()
        return
    assert eng.save() == fx["doc_bytes"]


@pytest.mark.parametrize("name", golden_util.doc_fixture_names())
def test_reencoding_a_loaded_document_reproduces_it(eng, name):
    fx = golden_util.load_fixture(name)
    eng.load_document(fx["doc_bytes"])
    eng.replay()
    assert eng.save() == fx["doc_bytes"]
    if "nodeflate" not in name:
        assert eng.save(reencode=True) == fx["doc_bytes"]


@pytest.mark.parametrize("case", golden_util.save_digest_cases(), ids=lambda c: c["workload"])
def test_save_of_generated_logs_matches_the_reference_digest(eng, case):
    """(digests of the block-size-patched reference: the stock reference is delivery-order dependent on these workloads)"""
    import hashlib
    log = loggen.config(case["workload"], case["scale"], False)
    eng.load_changes(log)
    eng.replay()
    doc = eng.save()
    assert len(doc) == case["doc_len"] and hashlib.sha256(doc).hexdigest() == case["doc_sha256"]


@pytest.mark.parametrize("workload", ["c3_map_lww", "c4_text_multi", "c4_text_single"])
def test_full_size_save_round_trip(eng, workload):
    """save() of the full-size logs: the oracle and the engine load the saved document to the same patch, whose content
    equals the replay's; re-encoding the loaded document reproduces the saved bytes."""
    log = loggen.config(workload, 1.0, False)
    eng.load_changes(log)
    eng.replay()
    want = json.loads(eng.patch_json())
    doc = eng.save()
    eng.load_document(doc)
    eng.replay()
    got = eng.patch_json()
    assert got == oracle_lib.OracleDoc.load_document(doc).patch_json()
    got = json.loads(got)
    assert got["diffs"] == want["diffs"] and got["maxOp"] == want["maxOp"] and got["deps"] == want["deps"] and got["clock"] == want["clock"]
    assert eng.save(reencode=True) == doc


@pytest.mark.parametrize("name", sorted(golden_util.history_golden()["fixtures"]))
def test_history_of_a_loaded_document_matches_the_reference(eng, name):
    """Backend.getAllChanges(Backend.load(doc)) against the unmodified reference: binary changes and hashes byte for
    byte; where the reference throws, the engine refuses (the JS path then raises the reference's error)."""
    want = golden_util.history_golden()["fixtures"][name]
    fx = golden_util.load_fixture(name)
    eng.load_document(fx["doc_bytes"])
    eng.replay()
    if "error" in want:
        with pytest.raises((engine.InvalidChanges, engine.UnsupportedChanges)):
            eng.doc_changes()
        return
    arena, offsets, hashes = eng.doc_changes()
    assert len(offsets) - 1 == want["n_changes"] and int(offsets[-1]) == want["bytes"]
    assert golden_util.history_digests(arena, offsets, hashes) == (want["changes_sha256"], want["hashes_sha256"])
    assert eng.patch_json() == fx["expected_load"]

@pytest.mark.parametrize("name", sorted(golden_util.longkey_history_golden()))
def test_history_with_long_keys_that_many_changes_overwrite(eng, name):
    """The rebuilt changes repeat a key the document's RLE column holds once: the device key column of the history is many times the
    document's (its output was sized from the document's column). Reference-made document and digests."""
    want = golden_util.longkey_history_golden()[name]
    eng.load_document(want["doc_bytes"])
    eng.replay()
    assert json.loads(eng.patch_json()) == want["patch"]
    arena, offsets, hashes = eng.doc_changes()
    assert len(offsets) - 1 == want["n_changes"] and int(offsets[-1]) == want["bytes"] > 20 * len(want["doc_bytes"])
    assert golden_util.history_digests(arena, offsets, hashes) == (want["changes_sha256"], want["hashes_sha256"])


@pytest.mark.parametrize("case", golden_util.history_golden()["generated"], ids=lambda c: "%s-%s-%s" % (c["workload"], c["scale"], c["deflate"]))
def test_history_after_save_and_load_of_generated_logs(eng, case):
    log = loggen.config(case["workload"], case["scale"], case["deflate"])
    eng.load_changes(log)
    eng.replay()
    in_hashes = eng.hashes()[eng.applied()]
    doc = eng.save()
    eng.load_document(doc)
    eng.replay()
    arena, offsets, hashes = eng.doc_changes()
    assert golden_util.history_digests(arena, offsets, hashes) == (case["changes_sha256"], case["hashes_sha256"])
    assert (hashes == in_hashes).all()


@pytest.mark.parametrize("workload", ["c2_text_typing", "c3_map_lww", "c4_text_multi"])
def test_full_size_history_round_trip(eng, workload):
    """Full-size logs: replay -> save -> load -> history gives back the very containers that went in (uncompressed form, byte for
    byte, in application order) with their hashes; replaying that history saves to the same document."""
    import time
    log = loggen.config(workload, 1.0, False)
    eng.load_changes(log)
    eng.replay()
    order = eng.applied()
    in_hashes = eng.hashes()[order]
    doc = eng.save()
    eng.load_document(doc)
    eng.replay()
    t0 = time.perf_counter()
    arena, offsets, hashes = eng.doc_changes(deflate=False)
    dt = time.perf_counter() - t0
    print(f"\n{workload}: history of {len(offsets) - 1} changes / {log.n_ops} ops rebuilt in {dt * 1e3:.1f} ms")
    assert (hashes == in_hashes).all()
    la, lo = np.asarray(log.arena), np.asarray(log.offsets)
    if (order == np.arange(len(order))).all():
        assert int(offsets[-1]) == int(lo[-1]) and (arena == la[:int(lo[-1])]).all()
    else:
        assert bytes(arena) == b"".join(bytes(la[int(lo[i]):int(lo[i + 1])]) for i in order)
    again = loggen.ChangeLog.from_changes([bytes(arena[int(offsets[i]):int(offsets[i + 1])]) for i in range(len(offsets) - 1)], name="history")
    eng.load_changes(again)
    eng.replay()
    assert eng.save() == doc


def test_empty_and_tiny_inputs(eng):
    """No changes at all; one change with a single op: patch against the oracle, save against the reference's bytes."""
    empty = loggen.ChangeLog.from_changes([])
    eng.load_changes(empty)
    eng.replay()
    assert eng.patch_json() == oracle_lib.OracleDoc(empty).patch_json()
    doc = eng.save()
    assert doc.hex() == "856f4a83b81a9544000400000000"  # Backend.save(Backend.init()) of the unmodified reference
    eng.load_document(doc)
    eng.replay()
    assert eng.patch_json() == '{"maxOp":0,"clock":{},"deps":[],"pendingChanges":0,"diffs":{"objectId":"_root","type":"map","props":{}}}'
    one = loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=1, ops_per_change=1, seed=5)
    eng.load_changes(one)
    eng.replay()
    assert eng.patch_json() == oracle_lib.OracleDoc(one).patch_json()
    want = eng.patch_json()
    assert oracle_lib.OracleDoc.load_document(eng.save()).patch_json() == want


def test_document_with_a_long_key_literal(eng):
    """200 k keys written once each: the saved document's key column is one literal of 200 k strings, which the key index
    resolves in its second stage (more doubling rounds than the first stage runs)."""
    log = loggen.generate(loggen.KIND_MAP_LWW, n_actors=1, n_rounds=1, n_keys=200_000, seed=9)
    eng.load_changes(log)
    eng.replay()
    want = json.loads(eng.patch_json())
    doc = eng.save()
    eng.load_document(doc)
    eng.replay()
    got = eng.patch_json()
    assert got == oracle_lib.OracleDoc.load_document(doc).patch_json()
    assert json.loads(got)["diffs"] == want["diffs"]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_objectid_shards_stitch_to_the_unsharded_patch(eng, world):
    """objectId sharding on one GPU: `world` contexts, each merging the objects it owns, their patch-IR fragments
    exported to host memory and stitched: byte-identical to the unsharded patch and to the oracle's -- 64 Text objects at full
    size, and a real-frontend document with nested objects."""
    logs = [loggen.config("c4_text_multi", 1.0), golden_util.load_fixture("campaign_mixed_1004")["log"], golden_util.load_fixture("frontend_mixed_6actors")["log"]]
    ranks = [engine.Engine(0) for _ in range(world)]
    try:
        for log in logs:
            if log is None:
                continue
            want = gpu_patch(eng, log)
            frags, offsets = [], [0]
            for r, e in enumerate(ranks):
                e.set_shard(r, world)
                e.load_changes(log)
                e.replay()
                buf = np.zeros(e.fragment_size(), dtype=np.uint8)
                assert e.export_fragment(buf.ctypes.data, buf.size, False) == buf.size
                frags.append(buf)
                offsets.append(offsets[-1] + buf.size)
            ranks[0].import_fragments(np.concatenate(frags), np.array(offsets, dtype=np.uint64))
            got = ranks[0].patch_json()
            assert hashlib.sha256(got.encode()).hexdigest() == hashlib.sha256(want.encode()).hexdigest()
            if log.n_ops < 100_000:
                assert got == oracle_lib.OracleDoc(log).patch_json()
            else:
                # every rank holds a real share of the work: no fragment carries (nearly) all edit records
                sizes = [f.size for f in frags]
                assert max(sizes) < 0.6 * sum(sizes)
        # saved documents (Backend.load; BASELINE config 5 is the other 8-GPU configuration): every rank decodes and checks all rows
        # and emits the records of the objects it owns
        docs = [golden_util.load_fixture("campaign_mixed_1008")["doc_bytes"], loggen.document_config(0.05)[0]]
        for doc in docs:
            eng.load_document(doc)
            eng.replay()
            want = eng.patch_json()
            frags, offsets = [], [0]
            for r, e in enumerate(ranks):
                e.set_shard(r, world)
                e.load_document(doc)
                e.replay()
                buf = np.zeros(e.fragment_size(), dtype=np.uint8)
                assert e.export_fragment(buf.ctypes.data, buf.size, False) == buf.size
                frags.append(buf)
                offsets.append(offsets[-1] + buf.size)
            ranks[0].import_fragments(np.concatenate(frags), np.array(offsets, dtype=np.uint64))
            assert hashlib.sha256(ranks[0].patch_json().encode()).hexdigest() == hashlib.sha256(want.encode()).hexdigest()
            assert want == oracle_lib.OracleDoc.load_document(doc).patch_json()
    finally:
        for e in ranks:
            e.close()


RCCL_WORKER = r'''
import hashlib, os, sys
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist   # (torch first: its bundled HIP runtime then serves the engine too, as in bench.py)
from automerge_classic_amd import engine, loggen, shard
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
log = loggen.config("c4_text_multi", 0.1)
e = engine.Engine(0)
e.load_changes(log); e.replay()
want = e.patch_json()
sr = shard.ShardedReplay(e, dist, torch.device("cuda", 0))
assert sr.step(lambda: e.load_changes(log))
assert e.patch_json() == want
dist.destroy_process_group()
print("rccl-ok", hashlib.sha256(want.encode()).hexdigest())
'''


def test_sharded_replay_through_rccl_world_1(tmp_path):
    """The host binding of the sharded path (automerge_classic_amd/shard.py) with the RCCL backend on the one GPU of the test box:
    fragment exported into a device tensor, all_gather_into_tensor, stitch. (World sizes > 1 over RCCL need more GPUs: the
    driver's multi-GPU bench; the stitching itself is covered for 2, 3 and 8 shards above and over gloo on CPU.)"""
    import subprocess
    import sys
    script = tmp_path / "rccl_worker.py"
    script.write_text(f"ROOT = {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}\n" + RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "rccl-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_headline_shape_at_125k_ops_equals_the_patched_reference(eng):
    """The engine on the GPU == the block-size-patched reference at 124,801 ops of the headline shape (tests/golden/headline_pin.json,
    oracle/make_headline_pin.py): getPatch text and Backend.save bytes, generator order (in-order fast path) and bench.py's shuffled
    order (device scheduler). The same digests pin the oracle (tests/test_oracle_golden.py), which checks the 1 M-op size."""
    for case, log in golden_util.headline_pin_cases():
        text = gpu_patch(eng, log)
        assert hashlib.sha256(text.encode()).hexdigest() == case["patch_sha256"], case["order"]
        assert eng.stats().fast_path == (1 if case["order"] == "in_order" else 2)
        doc = eng.save()
        assert len(doc) == case["save_len"] and hashlib.sha256(doc).hexdigest() == case["save_sha256"], case["order"]


LIB_RCCL_WORKER = r'''
import hashlib, os, sys
sys.path.insert(0, ROOT)
from automerge_classic_amd import engine, loggen   # (no torch in this process: the library opens librccl.so.1 itself)
log = loggen.config("c4_text_multi", 0.1)
doc = loggen.document_config(0.02)[0]
e = engine.Engine(0)
e.load_changes(log); e.replay(); want = e.patch_json()
e.load_document(doc); e.replay(); want_doc = e.patch_json()
e.shard_init(e.shard_unique_id(), 0, 1)            # ncclGetUniqueId + ncclCommInitRank
e.load_changes(log); e.sharded_replay()            # replay + ncclAllGather (sizes) + ncclAllGather (fragment, HBM -> HBM) + stitch
assert e.patch_json() == want
assert int(e.shard_fragment_bytes(1)[0]) > 1000
e.load_document(doc); e.sharded_replay(True)
assert e.patch_json() == want_doc
bad = loggen.ChangeLog(log.arena.copy(), log.offsets, log.n_ops); bad.arena[int(bad.offsets[1]) + 20] ^= 0x55
e.load_changes(bad)
try:
    e.sharded_replay(); raise SystemExit("corrupt batch accepted")
except engine.EngineError:
    pass
e.shard_finalize()                                  # ncclCommDestroy: the context is unsharded again
e.load_changes(log); e.replay(); assert e.patch_json() == want
print("lib-rccl-ok", hashlib.sha256(want.encode()).hexdigest())
'''


def test_sharded_replay_with_the_collective_inside_the_library_world_1(tmp_path):
    """am355_shard_init / am355_sharded_replay (include/am355.h): the library itself calls RCCL (librccl.so.1, opened on first use) on
    the context's stream -- communicator of one rank on the one GPU of the test box; world 2 runs between processes of the emulation
    over a stand-in for librccl (tests/test_multiproc.py, tests/test_js_host.py)."""
    import subprocess
    import sys
    script = tmp_path / "lib_rccl_worker.py"
    script.write_text(f"ROOT = {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}\n" + LIB_RCCL_WORKER)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "lib-rccl-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_mutated_changes_and_headers_never_disagree_with_the_oracle(eng):
    """tests/mutation_util.py on the GPU: single-byte damage in the op columns and in the header of a change (checksum repaired) is refused
    or gives the oracle's patch -- the wave-parallel header parse of k_parse_changes and its fallback to the lane-serial parser included."""
    import mutation_util
    equal, refused = mutation_util.column_mutations(lambda log: gpu_patch(eng, log))
    assert (equal, refused) == mutation_util.COLUMN_CAMPAIGN, (equal, refused)   # (deterministic campaign: the exact split, not a lower bound)
    equal, refused = mutation_util.header_mutations(lambda log: gpu_patch(eng, log))
    assert (equal, refused) == mutation_util.HEADER_CAMPAIGN, (equal, refused)


def test_more_list_objects_than_the_fused_list_order_holds(eng):
    """1100 Text objects: the three-launch form of the list order (k_obj_n, prefix sum, k_list_order) instead of k_list_order_objs."""
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=6, n_rounds=4, ins_per_change=12, del_per_change=3, n_objects=1100, seed=77)
    assert gpu_patch(eng, log) == oracle_lib.OracleDoc(log).patch_json()


def test_change_hashes_at_every_length_modulo_the_sha256_block(eng):
    """columnar.js:693-705: hash = SHA-256 of the chunk, checksum = its first four bytes. The padding of the last one or two blocks
    depends on the length modulo 64 (0x80 in the last data word, in a word of its own, in a block of its own; the bit length in the
    same block or the next): 140 first changes of 140 actors whose commit messages differ in length by one byte each."""
    import hashlib
    base = loggen.generate(loggen.KIND_MAP_LWW, n_actors=1, n_rounds=1, n_keys=3, seed=9).change(0)
    changes = [mutation_util.with_message_and_actor(base, bytes(97 + (i + j) % 26 for j in range(i)), hashlib.md5(b"actor%d" % i).digest()) for i in range(140)]
    assert len({len(c) % 64 for c in changes}) == 64
    log = loggen.ChangeLog.from_changes(changes, name="hash lengths")
    eng.load_changes(log)
    eng.replay()
    assert eng.patch_json() == oracle_lib.OracleDoc(log).patch_json()
    h = eng.hashes()
    for i, c in enumerate(changes):
        assert bytes(h[i]) == hashlib.sha256(c[8:]).digest(), i
    # one flipped message byte in one change: the checksum no longer matches
    bad = bytearray(changes[77])
    bad[-20] ^= 1
    with pytest.raises(engine.InvalidChanges):
        eng.load_changes(loggen.ChangeLog.from_changes(changes[:77] + [bytes(bad)] + changes[78:], name="bad checksum"))
        eng.replay()


def test_fat_changes_counted_by_the_wavefront(eng, monkeypatch):
    """Changes of more than 4 KB on average go through k_parse_changes<true>: the rows of the action column and the sum of the predNum
    column by the whole wavefront (orbit of the record headers by pointer doubling) instead of two lanes walking the records. Same
    rows, same patch as the lean kernel and as the oracle; damaged columns (checksum repaired) are refused or give the oracle's patch."""
    log = loggen.config("c3_map_lww", 0.3)
    assert log.raw_bytes / log.n_changes > 4096
    want = oracle_lib.OracleDoc(log).patch_json()
    fat = gpu_patch(eng, log)
    assert fat == want
    monkeypatch.setenv("AM355_PARSE_LEAN", "1")
    fat = gpu_patch(eng, log)
    monkeypatch.delenv("AM355_PARSE_LEAN")
    assert fat == want
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    changes = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    equal, refused = mutation_util.column_mutations(lambda l: gpu_patch(eng, l), rounds=60, changes=changes, seed=5)
    assert equal + refused == 60 and refused > 10
