"""Kernel-logic and host-logic tests that run WITHOUT a GPU: the engine's .hip sources compiled by g++ against the
HIP-runtime emulation in tests/emu (every kernel thread an OS thread). This checks the algorithms, the host
scheduler (fast and general path), the C ABI and the JSON renderer against the oracle and the reference goldens.
It says nothing about the GPU build: the parity tests proper are in test_engine_gpu.py (-m gpu)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_util
import mutation_util
import oracle_lib
from automerge_classic_amd import engine, loggen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libam355_emu.so")


@pytest.fixture(scope="module")
def eng():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    e = engine.Engine(0, EMU_LIB)
    yield e
    e.close()


def emu_patch(eng, log):
    eng.load_changes(log)
    eng.replay()
    return eng.patch_json()


@pytest.mark.parametrize("name", golden_util.fixture_names())
def test_golden_reference_patches(eng, name):
    fx = golden_util.load_fixture(name)
    assert emu_patch(eng, fx["log"]) == fx["expected"]
    general = any(k in name for k in ("shuffled", "pending", "dups"))
    assert eng.stats().fast_path == (2 if general else 1)   # (2: the general path, scheduled on the device)


def test_counters_and_valueless_rows_inside_lists(eng):
    """new.js:937-965, 1010-1018, 1026-1033 in whole-document patches (refused through round 4): patch, save, load, history of every case
    of tests/golden/list_quirks.json equal the unmodified reference's; exactly one case is left to the JS path."""
    served, refused = golden_util.check_list_quirk_cases(eng, engine)
    assert set(refused) == golden_util.LIST_QUIRK_REFUSED and served >= 47


def test_headline_shape_at_125k_ops_equals_the_patched_reference(eng):
    """getPatch text and Backend.save bytes of the (emulated) engine == the block-size-patched reference's digests on c4_text_single
    x0.125, in both delivery orders (tests/golden/headline_pin.json; in-order fast path and general scheduler)."""
    import hashlib
    for case, log in golden_util.headline_pin_cases():
        text = emu_patch(eng, log)
        assert hashlib.sha256(text.encode()).hexdigest() == case["patch_sha256"], case["order"]
        assert eng.stats().fast_path == (1 if case["order"] == "in_order" else 2)
        doc = eng.save()
        assert len(doc) == case["save_len"] and hashlib.sha256(doc).hexdigest() == case["save_sha256"], case["order"]


def test_defect_fixture_both_delivery_orders(eng):
    """Inputs on which the stock reference diverges with delivery order: the engine gives the block-size-patched
    reference's document for both orders."""
    fx = golden_util.defect_fixture()
    assert emu_patch(eng, fx["log"]) == fx["patch_bigblock"] != fx["patch"]
    assert emu_patch(eng, fx["log_reversed"]) == fx["patch_bigblock_reversed"]


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
    # a corrupted byte must be caught by the chunk checksum
    bad = bytearray(fx["doc_bytes"])
    bad[len(bad) // 2] ^= 0x40
    with pytest.raises(engine.InvalidChanges):
        eng.load_document(bytes(bad))


@pytest.mark.parametrize("kind,kw", [
    (loggen.KIND_TEXT_TYPING, dict(n_ops=1500, ops_per_change=40)),
    (loggen.KIND_TEXT_TYPING, dict(n_ops=300, ops_per_change=1)),
    (loggen.KIND_MAP_LWW, dict(n_actors=8, n_rounds=4, n_keys=200)),
    (loggen.KIND_TEXT_CONCURRENT, dict(n_actors=7, n_rounds=3, ins_per_change=100, del_per_change=10, n_objects=1)),
    (loggen.KIND_TEXT_CONCURRENT, dict(n_actors=3, n_rounds=2, ins_per_change=40, del_per_change=700, n_objects=1)),  # columns > 1 KiB: lane-serial decoder
    (loggen.KIND_MAP_LWW, dict(n_actors=3, n_rounds=3, n_keys=1500)),  # 500 literal keys/values per change
    (loggen.KIND_MAP_LWW, dict(n_actors=700, n_rounds=1, n_keys=2)),   # ~350 concurrent values per key: beyond k_map_group_rank's walk, the trigger passes
    (loggen.KIND_MAP_LWW, dict(n_actors=700, n_rounds=1, n_keys=1)),   # every emission on ONE key: no radix pass runs at all before the walk gives up (perm_out was left unwritten)
    (loggen.KIND_TEXT_CONCURRENT, dict(n_actors=300, n_rounds=2, ins_per_change=2, del_per_change=1, n_objects=1)),  # 300 children of _head: radix fallback
    (loggen.KIND_TEXT_CONCURRENT, dict(n_actors=16, n_rounds=3, ins_per_change=30, del_per_change=8, n_objects=5)),
    (loggen.KIND_TEXT_CONCURRENT, dict(n_actors=40, n_rounds=210, ins_per_change=1, del_per_change=0, n_objects=1)),  # 8400 typing runs: tour beyond the LDS list ranking
])
def test_generated_workloads_match_oracle(eng, kind, kw):
    log = loggen.generate(kind, seed=11, deflate=True, **kw)
    assert emu_patch(eng, log) == oracle_lib.OracleDoc(log).patch_json()
    assert eng.stats().fast_path == 1
    # the same changes in a random delivery order: general scheduler, same document
    perm = np.random.default_rng(5).permutation(log.n_changes)
    shuf = log.reordered(perm)
    assert emu_patch(eng, shuf) == oracle_lib.OracleDoc(shuf).patch_json()


def scheduler_cases():
    """Delivery orders the general scheduler (new.js:1550-1597, 1822-1841) has to get right: (name, log)."""
    rng = np.random.default_rng(17)
    text = loggen.generate(loggen.KIND_TEXT_CONCURRENT, seed=21, n_actors=12, n_rounds=9, ins_per_change=7, del_per_change=2, n_objects=3)
    n = text.n_changes
    yield "shuffled", text.reordered(rng.permutation(n))
    yield "reversed", text.reordered(list(range(n - 1, -1, -1)))                       # every change waits for the ones behind it: one pass per round
    yield "duplicates", text.reordered(list(rng.permutation(n)) + [3, 3, 0, n - 1])    # later copies are dropped (new.js:1566)
    yield "duplicates_first", text.reordered([5, 5] + list(range(n)))                  # a copy in front of the original's dependencies
    keep = [i for i in rng.permutation(n) if i not in (2, 17)]
    yield "missing_deps", text.reordered(keep)                                         # dependents of the missing changes stay queued
    typing = loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=1200, ops_per_change=3, seed=4)   # one actor: a chain of 400 changes
    yield "chain_reversed", typing.reordered(list(range(typing.n_changes - 1, -1, -1)))      # 400 passes
    yield "chain_shuffled", typing.reordered(np.random.default_rng(2).permutation(typing.n_changes))
    maps = loggen.generate(loggen.KIND_MAP_LWW, seed=9, n_actors=8, n_keys=40, n_rounds=6) if hasattr(loggen, "KIND_MAP_LWW") else None
    if maps is not None:
        yield "map_shuffled", maps.reordered(np.random.default_rng(8).permutation(maps.n_changes))


def check_scheduler_variants(eng, name, log):
    """The device scheduler (am355_sched.hip) in its register/LDS form and in its global-memory form, and the host's restatement: the
    same patch text (which carries clock, heads, pendingChanges and the document), equal to the oracle's."""
    want = oracle_lib.OracleDoc(log).patch_json()
    seen = {}
    for variant, env in (("device", {}), ("device_global_memory", {"AM355_SCHED_BIG": "1"}), ("host", {"AM355_HOST_SCHEDULE": "1"}),
                         ("device_out_of_sweeps", {"AM355_SCHED_SWEEPS": "2"})):
        for k in ("AM355_SCHED_BIG", "AM355_HOST_SCHEDULE", "AM355_SCHED_SWEEPS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            eng.load_changes(log)
            eng.replay()
            got = eng.patch_json()
            st = eng.stats()
            seen[variant] = (st.fast_path, st.n_applied, st.n_pending)
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert got == want, f"{name} / {variant}"
    assert seen["device"][0] == 2 and seen["device_global_memory"][0] == 2 and seen["host"][0] == 0, (name, seen)
    assert len({v[1:] for v in seen.values()}) == 1, (name, seen)
    return seen


@pytest.mark.parametrize("name,log", list(scheduler_cases()), ids=[n for n, _ in scheduler_cases()])
def test_general_scheduler_on_the_device_equals_host_and_oracle(eng, name, log):
    seen = check_scheduler_variants(eng, name, log)
    if name in ("reversed", "chain_reversed"):
        assert seen["device_out_of_sweeps"][0] == 0    # two sweeps do not settle these: the device says so and the host schedules
    if name == "missing_deps":
        assert seen["device"][2] > 0


def test_empty_batch_and_single_change(eng):
    empty = loggen.ChangeLog.from_changes([])
    assert emu_patch(eng, empty) == oracle_lib.OracleDoc(empty).patch_json()
    log = loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=0, ops_per_change=10, seed=1)
    assert emu_patch(eng, log) == oracle_lib.OracleDoc(log).patch_json()


def test_hashes_and_rows(eng):
    import hashlib
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=4, n_rounds=2, ins_per_change=20, del_per_change=4, n_objects=1, seed=3)
    eng.load_changes(log)
    eng.replay()
    h = eng.hashes()
    for i in range(log.n_changes):
        assert bytes(h[i]) == hashlib.sha256(log.change(i)[8:]).digest()
    rows = eng.rows()
    ref = oracle_lib.OracleDoc(log)
    assert len(rows["action"]) == ref.n_ops
    # succ counts: every delete adds one successor to the element it removes
    assert int(rows["succ_cnt"].sum()) == int((rows["action"] == 3).sum())

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


def test_invalid_and_unsupported_inputs_are_reported(eng):
    log = loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=50, ops_per_change=10, seed=2)
    arena = log.arena.copy()
    arena[int(log.offsets[2]) + 30] ^= 0x1
    bad = loggen.ChangeLog(arena, log.offsets, log.n_ops)
    eng.load_changes(bad)
    with pytest.raises(engine.InvalidChanges) as ei:
        eng.replay()
    assert "BAD_CHECKSUM" in ei.value.flag_names
    # a sequence gap: drop the second change of a single-actor history but keep the third (its dep is then missing -> pending)
    part = log.reordered([0, 1, 3, 4, 5])
    assert json.loads(emu_patch(eng, part))["pendingChanges"] == 3
    assert emu_patch(eng, part) == oracle_lib.OracleDoc(part).patch_json()


def _uleb(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def test_truncated_deflate_streams_are_rejected_not_retried(eng):
    """A DEFLATE stream cut short must end in AM355_E_INVALID / BAD_DEFLATE (the reference throws a catchable error), for a
    DEFLATEd change (chunk type 2, columnar.js:813-823) and for a DEFLATEd document column (columnar.js:1062-1067)."""
    import hashlib
    import zlib
    log = loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=4000, ops_per_change=2000, seed=4, deflate=True)
    c = log.change(1)
    assert c[8] == 2
    off, clen, shift = 9, 0, 0
    while True:
        b = c[off]
        off += 1
        clen |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            break
    for cut in (1, 5, clen // 2):
        data = c[off:off + clen - cut]
        bad = c[:9] + _uleb(len(data)) + data
        eng.load_changes(loggen.ChangeLog.from_changes([log.change(0)]))  # (context stays usable around a rejected batch)
        with pytest.raises(engine.InvalidChanges) as ei:
            eng.load_changes(loggen.ChangeLog.from_changes([log.change(0), bad]))
        assert "BAD_DEFLATE" in ei.value.flag_names
    # a document whose one op column is a truncated raw-DEFLATE stream
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    stream = comp.compress(bytes(range(256)) * 40) + comp.flush()
    for cut in (1, len(stream) // 2):
        col = stream[:-cut]
        body = _uleb(0) + _uleb(0) + _uleb(0) + _uleb(1) + _uleb(0x42 | 8) + _uleb(len(col)) + col
        chunk = bytes([0]) + _uleb(len(body)) + body
        doc = bytes([0x85, 0x6F, 0x4A, 0x83]) + hashlib.sha256(chunk).digest()[:4] + chunk
        with pytest.raises(engine.InvalidChanges) as ei:
            eng.load_document(doc)
        assert "BAD_DEFLATE" in ei.value.flag_names
    # offsets that are not ascending are an argument error, not a crash
    arena = np.zeros(16, dtype=np.uint8)
    with pytest.raises(engine.EngineError) as ei:
        eng.load_changes(loggen.ChangeLog(arena, np.array([0, 12, 4], dtype=np.uint64), 0))
    assert ei.value.code == engine.AM355_E_ARG


@pytest.mark.parametrize("deflate", [False, True])
def test_sliced_staging_gathers_the_same_arena(eng, monkeypatch, deflate):
    """am355_load_changes gathers large batches in slices on host threads (gather / inflate + H2D per slice): with the slice
    size lowered the sliced path runs on a small log; the staged raw arena, offsets and the patch equal the one-slice result."""
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, seed=5, n_actors=8, n_rounds=6, ins_per_change=60, del_per_change=15, n_objects=2, deflate=deflate)
    eng.load_changes(log)
    eng.replay()
    want_raw, want_patch = eng.raw(), eng.patch_json()
    monkeypatch.setenv("AM355_SLICE_BYTES", "600")
    eng.load_changes(log)
    eng.replay()
    got_raw = eng.raw()
    assert np.array_equal(got_raw[0], want_raw[0]) and np.array_equal(got_raw[1], want_raw[1])
    assert eng.patch_json() == want_patch == oracle_lib.OracleDoc(log).patch_json()
    if not deflate:
        # plain changes of a large batch are gathered in small units, the H2D copies enqueued per group of units
        monkeypatch.delenv("AM355_SLICE_BYTES")
        monkeypatch.setenv("AM355_GATHER_UNIT", "97")
        eng.load_changes(log)
        eng.replay()
        got_raw = eng.raw()
        assert np.array_equal(got_raw[0], want_raw[0]) and np.array_equal(got_raw[1], want_raw[1])
        assert eng.patch_json() == want_patch


def test_device_primitives(eng):
    rng = np.random.default_rng(1)
    for n in (1, 64, 2049, 8192, 9000, 70_001, 135_001):  # (the last: more than 64 sort tiles -- the scan of the histogram table as its own launch)
        vals = rng.integers(0, 5, n, dtype=np.uint32)
        out, total = eng.test_scan(vals)
        assert np.array_equal(out, np.concatenate(([0], np.cumsum(vals)[:-1])).astype(np.uint32)) and total == int(vals.sum())
        keys = rng.integers(0, 1 << 20, n, dtype=np.uint64)
        k, v = eng.test_sort(keys, np.arange(n, dtype=np.uint32), 20)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(k, keys[order]) and np.array_equal(v, order.astype(np.uint32))
    # three-launch scan (more than 1024 tiles: k_scan_sums between the tile sums and the apply kernel), aligned and unaligned length
    for n in (2_200_003, 2_300_000):
        vals = rng.integers(0, 3, n, dtype=np.uint32)
        for lookback in ("0", "1"):   # (1: the single-pass form with decoupled look-back, off by default -- measured slower)
            os.environ["AM355_SCAN_LOOKBACK"] = lookback
            try:
                out, total = eng.test_scan(vals)
            finally:
                del os.environ["AM355_SCAN_LOOKBACK"]
            assert np.array_equal(out, np.concatenate(([0], np.cumsum(vals)[:-1])).astype(np.uint32)) and total == int(vals.sum())


@pytest.mark.parametrize("serial", [False, True])
@pytest.mark.parametrize("deflate", [False, True])
def test_generated_document_matches_oracle(eng, serial, deflate, monkeypatch):
    """Synthetic config-5 shaped document: parallel big-column decoder and the lane-serial one against the oracle."""
    doc, rows = loggen.generate_document(n_actors=7, n_texts=3, text_len=180, n_maps=3, keys_per_map=40, n_submaps=2, n_lists=2, list_len=90,
                                         deflate=deflate, seed=0xD0C5)
    if serial:
        monkeypatch.setenv("AM355_DOC_SERIAL", "1")
    else:
        monkeypatch.delenv("AM355_DOC_SERIAL", raising=False)
    eng.load_document(doc)
    eng.replay()
    assert eng.stats().n_ops == rows
    assert eng.patch_json() == oracle_lib.OracleDoc.load_document(doc).patch_json()


def test_document_columns_parallel_equals_serial(eng, monkeypatch):
    doc, rows = loggen.generate_document(n_actors=5, n_texts=2, text_len=700, n_maps=2, keys_per_map=90, n_submaps=3, n_lists=3, list_len=200,
                                         deflate=False, seed=0xD0C6)
    got = []
    for serial in ("0", "1"):
        monkeypatch.setenv("AM355_DOC_SERIAL", serial)
        eng.load_document(doc)
        eng.replay()
        got.append(eng.rows())
    for k in got[0]:
        assert np.array_equal(got[0][k], got[1][k]), k


def _mutated_documents(n, seed):
    """Single-byte mutations in the op columns of an uncompressed document, container checksum repaired."""
    import hashlib
    import random
    doc, _ = loggen.generate_document(n_actors=4, n_texts=2, text_len=60, n_maps=2, keys_per_map=25, n_submaps=2, n_lists=2, list_len=40,
                                      deflate=False, seed=77)
    rng = random.Random(seed)
    for _ in range(n):
        d = bytearray(doc)
        pos = rng.randrange(len(d) // 3, len(d))
        d[pos] = rng.randrange(256)
        d[4:8] = hashlib.sha256(bytes(d[8:])).digest()[:4]
        yield pos, bytes(d)


def test_mutated_documents_never_disagree_with_the_oracle(eng):
    """Whatever a damaged document decodes to, the engine either rejects it (the JS host then runs the reference path)
    or produces exactly the oracle's patch; it never accepts what the oracle rejects."""
    agree = 0
    for pos, d in _mutated_documents(200, 5):
        try:
            want = oracle_lib.OracleDoc.load_document(d).patch_json()
        except oracle_lib.OracleError:
            want = None
        try:
            eng.load_document(d)
            eng.replay()
            got = eng.patch_json()
        except engine.EngineError:
            got = None
        if got is not None:
            assert want is not None, f"engine accepted a document the oracle rejects (byte {pos})"
            assert got == want, f"different patch for mutation at byte {pos}"
            agree += 1
    assert agree > 20


@pytest.mark.parametrize("name", golden_util.fixture_names())
def test_save_after_replay_is_byte_identical_to_the_reference(eng, name):
    """Backend.save(Backend.loadChanges(Backend.init(), changes)) -- the golden holds the unmodified reference's bytes."""
    fx = golden_util.load_fixture(name)
    eng.load_changes(fx["log"])
    eng.replay()
    if "doc_bytes" not in fx:  # changes left in the queue: that document is saved by the JS path
        with pytest.raises(engine.UnsupportedChanges):
            eng.save()
        return
    assert eng.save() == fx["doc_bytes"]


@pytest.mark.parametrize("name", golden_util.doc_fixture_names())
def test_reencoding_a_loaded_document_reproduces_it(eng, name):
    """Column encoders alone: decode the op columns of a reference-written document, encode them again."""
    fx = golden_util.load_fixture(name)
    eng.load_document(fx["doc_bytes"])
    eng.replay()
    assert eng.save() == fx["doc_bytes"]  # unchanged document: the bytes it was given (new.js:2034)
    if "nodeflate" not in name:            # (that fixture was written without column compression; save() always compresses)
        assert eng.save(reencode=True) == fx["doc_bytes"]


@pytest.mark.parametrize("workload,scale", [("c2_text_typing", 0.02), ("c3_map_lww", 0.25), ("c4_text_multi", 0.03)])
def test_saved_document_loads_to_the_same_patch(eng, workload, scale):
    """save() of a generated log, loaded again (engine and oracle): same document."""
    log = loggen.config(workload, scale, False)
    want = emu_patch(eng, log)
    doc = eng.save()
    eng.load_document(doc)
    eng.replay()
    got = eng.patch_json()
    oracle = oracle_lib.OracleDoc.load_document(doc).patch_json()
    assert got == oracle
    # the loaded document has no pending queue and the same content; `clock` key order is first-appearance in both
    assert json.loads(got)["diffs"] == json.loads(want)["diffs"]
    assert json.loads(got)["maxOp"] == json.loads(want)["maxOp"] and json.loads(got)["deps"] == json.loads(want)["deps"]


@pytest.mark.parametrize("case", golden_util.save_digest_cases(), ids=lambda c: c["workload"])
def test_save_of_generated_logs_matches_the_reference_digest(eng, case):
    """Same bytes as the unmodified reference's Backend.save on the generated workloads (digests: oracle/make_save_golden.py)."""
    import hashlib
    log = loggen.config(case["workload"], case["scale"], False)
    assert log.n_ops == case["n_ops"] and log.n_changes == case["n_changes"]
    eng.load_changes(log)
    eng.replay()
    doc = eng.save()
    assert len(doc) == case["doc_len"] and hashlib.sha256(doc).hexdigest() == case["doc_sha256"]


@pytest.mark.parametrize("name", sorted(golden_util.history_golden()["fixtures"]))
def test_history_of_a_loaded_document_matches_the_reference(eng, name):
    """Backend.getAllChanges(Backend.load(doc)): the rebuilt binary changes and their hashes, byte for byte; where
    the reference throws (documents whose rows contradict their change metadata, byte-array values it re-encodes wrongly), the
    engine refuses too and the JS path raises the reference's error."""
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
    # the patch is untouched by the history query
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
    """log -> replay -> save -> load -> history: the reference's digests, and the very changes that went in."""
    log = loggen.config(case["workload"], case["scale"], case["deflate"])
    assert log.n_ops == case["n_ops"]
    eng.load_changes(log)
    eng.replay()
    order = eng.applied()
    in_hashes = eng.hashes()[order]
    doc = eng.save()
    eng.load_document(doc)
    eng.replay()
    arena, offsets, hashes = eng.doc_changes()
    assert len(offsets) - 1 == case["n_changes"] and int(offsets[-1]) == case["bytes"]
    assert golden_util.history_digests(arena, offsets, hashes) == (case["changes_sha256"], case["hashes_sha256"])
    assert (hashes == in_hashes).all()
    # without compression: the uncompressed containers of the input, byte for byte
    arena, offsets, _ = eng.doc_changes(deflate=False)
    plain = loggen.config(case["workload"], case["scale"], False)
    pa, po = np.asarray(plain.arena), np.asarray(plain.offsets)
    want = b"".join(bytes(pa[int(po[i]):int(po[i + 1])]) for i in order)
    assert bytes(arena) == want
    # and they replay to the same document
    from automerge_classic_amd.loggen import ChangeLog
    again = ChangeLog.from_changes([bytes(arena[int(offsets[i]):int(offsets[i + 1])]) for i in range(len(offsets) - 1)], name="history")
    eng.load_changes(again)
    eng.replay()
    assert eng.save() == doc


def test_map_key_with_a_byte_order_mark_goes_to_the_js_path(eng):
    """The reference decodes strings with TextDecoder, which drops a leading U+FEFF (encoding.js:9-17): values are rendered without
    it (golden hand_bom_values); a KEY that loses its first character can collide with another key -- the reference's own decoder
    throws on such documents -- so the engine leaves those to the JS path."""
    import base64
    change = base64.b64decode("hW9Kgw1pqGMBPwACqqoBAeXyxtUGDkluaXRpYWxpemF0aW9uAAYVDjQBQgJWAlcCcAJ+Bu+7v2tleQVvdGhlcgICAQIWdncCAA==")
    log = loggen.ChangeLog.from_changes([change], name="bom key")
    eng.load_changes(log)
    eng.replay()
    with pytest.raises(engine.UnsupportedChanges):
        eng.patch_json()
    with pytest.raises(oracle_lib.OracleError):
        oracle_lib.OracleDoc(log).patch_json()


def test_history_call_sequence_and_refusals(eng, monkeypatch):
    """am355_doc_changes: only after load_document + replay; the serial document decoder (diagnostic) does not feed it; the result is
    cached per flag and dropped by the next load."""
    fx = golden_util.load_fixture("frontend_text_4actors")
    eng.load_changes(fx["log"])
    eng.replay()
    with pytest.raises(engine.EngineError):   # a replayed change log is not a loaded document
        eng.doc_changes()
    eng.load_document(fx["doc_bytes"])
    with pytest.raises(engine.EngineError):   # not replayed yet
        eng.doc_changes()
    eng.replay()
    a1, o1, h1 = eng.doc_changes()
    a2, o2, h2 = eng.doc_changes()            # served from the context
    assert bytes(a1) == bytes(a2) and (o1 == o2).all() and (h1 == h2).all()
    a3, o3, h3 = eng.doc_changes(deflate=False)
    assert (h3 == h1).all() and len(o3) == len(o1)   # (same changes; compression may even enlarge small ones)
    monkeypatch.setenv("AM355_DOC_SERIAL", "1")
    eng.load_document(fx["doc_bytes"])
    eng.replay()
    with pytest.raises(engine.UnsupportedChanges):
        eng.doc_changes()
    monkeypatch.delenv("AM355_DOC_SERIAL")
    eng.load_document(fx["doc_bytes"])
    eng.replay()
    assert bytes(eng.doc_changes()[0]) == bytes(a1)


def test_inflate_without_libdeflate_gives_the_same_result(tmp_path):
    """The host inflate goes through libdeflate when the system has it; AM355_NO_LIBDEFLATE=1 (read once per process) forces zlib."""
    script = tmp_path / "w.py"
    script.write_text(f"""
import sys, hashlib
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, "tests")!r})
from automerge_classic_amd import engine, loggen
eng = engine.Engine(0, {EMU_LIB!r})
log = loggen.config("c4_text_multi", 0.03, True)
eng.load_changes(log); eng.replay()
h = hashlib.sha256(eng.patch_json().encode())
doc, rows = loggen.generate_document(n_actors=6, n_texts=3, text_len=900, n_maps=2, keys_per_map=300, n_submaps=2, n_lists=2, list_len=300, deflate=True, seed=5)
eng.load_document(doc); eng.replay()
h.update(eng.patch_json().encode())
print(h.hexdigest())
""")
    outs = []
    for env in ({}, {"AM355_NO_LIBDEFLATE": "1"}):
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 64


@pytest.mark.parametrize("name", ["frontend_text_8actors", "frontend_mixed_6actors", "campaign_text_2003"])
def test_threaded_rendering_produces_the_same_text(eng, name, monkeypatch):
    """Long edit lists are rendered by several host threads; with a tiny chunk size the goldens go through that path."""
    monkeypatch.setenv("AM355_RENDER_CHUNK", "3")
    fx = golden_util.load_fixture(name)
    assert emu_patch(eng, fx["log"]) == fx["expected"]


def test_mutated_changes_never_disagree_with_the_oracle(eng):
    """Single-byte damage in the op columns of one change (checksum repaired): the engine may refuse more than the oracle does
    (the JS host then runs the reference path), but it never accepts what the oracle rejects and never produces another patch."""
    equal, refused = mutation_util.column_mutations(lambda log: emu_patch(eng, log))
    assert (equal, refused) == mutation_util.COLUMN_CAMPAIGN, (equal, refused)   # (deterministic campaign: the exact split, not a lower bound)


def test_mutated_change_headers_never_disagree_with_the_oracle(eng):
    """The same for the HEADER of a change -- dependency count and hashes, actor, seq, startOp, time, message, the table of other actors,
    the column directory -- which k_parse_changes reads by the whole wavefront (five ballot-tokenised windows) and hands to the
    lane-serial parser whenever anything is irregular: single-byte damage (checksum repaired) must be accepted with the oracle's patch
    or refused, over several fixtures (short and long actor tables, one and many dependencies)."""
    equal, refused = mutation_util.header_mutations(lambda log: emu_patch(eng, log))
    assert (equal, refused) == mutation_util.HEADER_CAMPAIGN, (equal, refused)


@pytest.mark.parametrize("tile", ["16", "64", "1024"])
def test_key_column_literals_across_tiles(eng, tile, monkeypatch):
    """The key-string index follows the strings of a literal inside LDS tiles and finishes the literals that leave their tile from the
    tiles' (hops, exit) pairs: with tiles of a few bytes every literal of these documents crosses tiles, many of them several."""
    monkeypatch.setenv("AM355_KEY_TILE", tile)
    if tile == "64":
        monkeypatch.setenv("AM355_KEY_CONT_SMALL", "0")   # every last stretch through the LDS walker (kk_kth_big)
    if tile == "16":
        monkeypatch.setenv("AM355_KEY_JUMPS", "1")        # the walker gives up after one window: the true literals it cut off make the load repeat without the bound
    for name in ("synthetic_doc_medium", "frontend_mixed_6actors", "campaign_mixed_1008"):
        fx = golden_util.load_fixture(name)
        eng.load_document(fx["doc_bytes"])
        eng.replay()
        assert eng.patch_json() == fx["expected_load"], name
    doc, rows = loggen.generate_document(n_actors=5, n_texts=2, text_len=300, n_maps=3, keys_per_map=120, n_submaps=2, n_lists=2, list_len=100,
                                         deflate=False, seed=0xD0C8)
    eng.load_document(doc)
    eng.replay()
    assert eng.patch_json() == oracle_lib.OracleDoc.load_document(doc).patch_json()


def test_document_with_a_long_key_literal(eng):
    """6000 keys written once each: one key-column literal longer than the first stage of the key index resolves."""
    log = loggen.generate(loggen.KIND_MAP_LWW, n_actors=1, n_rounds=1, n_keys=6000, seed=9)
    emu_patch(eng, log)
    doc = eng.save()
    eng.load_document(doc)
    eng.replay()
    assert eng.patch_json() == oracle_lib.OracleDoc.load_document(doc).patch_json()


def test_more_list_objects_than_the_fused_list_order_holds(eng):
    """1100 Text objects: beyond the 1023 whose element counts k_list_order_objs works out in LDS, so the three-launch form (k_obj_n,
    prefix sum, k_list_order) and the standalone map kernels run; several paths in one tour (64-bit list ranking)."""
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=6, n_rounds=4, ins_per_change=12, del_per_change=3, n_objects=1100, seed=77)
    assert emu_patch(eng, log) == oracle_lib.OracleDoc(log).patch_json()
    assert eng.stats().n_objects == 1101


def test_full_size_headline_workload_emulated(eng):
    """BASELINE.json's headline configuration at FULL size (1,020,801 ops, 4097 changes, 64 actors, one Text: a tour of 16 k entries, the
    single-path list ranking, the wave-per-change decoder on every change) through the CPU emulation of the kernels, against the oracle."""
    log = loggen.config("c4_text_single", 1.0, False)
    assert emu_patch(eng, log) == oracle_lib.OracleDoc(log).patch_json()
    st = eng.stats()
    assert st.n_ops == 1020801 and st.n_changes == 4097 and st.fast_path == 1


def test_applied_order(eng):
    """am355_get_applied: input order on the fast path, duplicates dropped, queued changes absent."""
    fx = golden_util.load_fixture("frontend_text_4actors")
    emu_patch(eng, fx["log"])
    n = len(fx["log"].offsets) - 1
    assert list(eng.applied()) == list(range(n))
    dups = golden_util.load_fixture("frontend_text_4actors_dups")
    emu_patch(eng, dups["log"])
    assert list(eng.applied()) == list(range(n))          # the appended copies are not applied again
    pend = golden_util.load_fixture("hand_conflicts_pending")
    emu_patch(eng, pend["log"])
    st = eng.stats()
    assert len(eng.applied()) == st.n_applied and st.n_pending > 0
    shuf = golden_util.load_fixture("frontend_mixed_3actors_shuffled")
    emu_patch(eng, shuf["log"])
    order = list(eng.applied())
    assert sorted(order) == list(range(len(shuf["log"].offsets) - 1)) and order != sorted(order)


def test_changes_without_ops_are_part_of_the_history(eng):
    """A change with no ops is applied like any other (found by the reference-suite vectors): it appears in the application
    order and in the saved document."""
    import base64
    import gzip
    with open(os.path.join(golden_util.GOLDEN_DIR, "ref_suite_vectors.json.gz"), "rb") as f:
        d = json.loads(gzip.decompress(f.read()))
    pool = [base64.b64decode(x) for x in d["pool"]]
    v = d["vectors"][817]
    blobs = [pool[k] for k in v["changes"]]
    log = loggen.ChangeLog.from_changes(blobs)
    assert json.loads(emu_patch(eng, log)) == json.loads(v["patch"])
    assert list(eng.applied()) == [0, 1] and eng.stats().n_ops < 2   # at least one of the two changes has no ops
    import hashlib
    assert hashlib.sha256(eng.save()).hexdigest() == v["doc_sha256"]


def test_call_sequence_and_argument_errors():
    """C-ABI error behaviour (include/am355.h): state errors instead of crashes, results owned by the context."""
    import ctypes
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    e = engine.Engine(0, EMU_LIB)
    try:
        with pytest.raises(engine.EngineError) as ei:
            e.replay()                      # nothing staged
        assert ei.value.code == engine.AM355_E_STATE
        with pytest.raises(engine.EngineError) as ei:
            e.save()                        # nothing replayed
        assert ei.value.code == engine.AM355_E_STATE
        with pytest.raises(engine.EngineError) as ei:
            e.patch_json()
        assert ei.value.code == engine.AM355_E_STATE
        L = e._L
        assert L.am355_save(e._h, 0, None, None) == engine.AM355_E_ARG
        assert L.am355_load_document(e._h, None, 0) == engine.AM355_E_ARG
        n = ctypes.c_uint32()
        assert L.am355_get_applied(e._h, None, ctypes.byref(n)) == engine.AM355_E_STATE
        # a rejected batch leaves the context usable
        fx = golden_util.load_fixture("hand_conflicts")
        bad = bytearray(fx["log"].arena.tobytes())
        bad[6] ^= 0xFF                      # checksum byte of the first change
        log = loggen.ChangeLog.from_changes([bytes(bad[int(fx["log"].offsets[i]):int(fx["log"].offsets[i + 1])]) for i in range(len(fx["log"].offsets) - 1)])
        with pytest.raises(engine.InvalidChanges):
            e.load_changes(log)
            e.replay()
        assert emu_patch(e, fx["log"]) == fx["expected"]
        # a document after changes and changes after a document
        doc = e.save()
        e.load_document(doc)
        e.replay()
        with pytest.raises(engine.EngineError):
            e.hashes()                      # a loaded document has no per-change hashes
        assert emu_patch(e, fx["log"]) == fx["expected"]
    finally:
        e.close()


@pytest.mark.parametrize("world", [2, 3])
def test_objectid_shards_of_changes_and_documents_stitch_to_the_unsharded_patch(world):
    """objectId sharding under the emulation, `world` contexts on one machine: every rank merges (change logs) or emits
    (saved documents) the objects it owns, the fragments stitch by object index to the unsharded patch. For change logs a change none
    of whose rows this rank owns is decoded only as far as k_resolve looks at foreign rows (am355_decode.hip)."""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    single = engine.Engine(0, EMU_LIB)
    ranks = [engine.Engine(0, EMU_LIB) for _ in range(world)]
    try:
        logs = [loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=6, n_rounds=3, ins_per_change=25, del_per_change=6, n_objects=7, seed=41),
                golden_util.load_fixture("campaign_mixed_1004")["log"], golden_util.load_fixture("frontend_mixed_6actors")["log"]]
        docs = [golden_util.load_fixture(n)["doc_bytes"] for n in ("campaign_mixed_1008", "frontend_mixed_6actors", "synthetic_doc_medium")]
        docs.append(loggen.generate_document(n_actors=5, n_texts=6, text_len=200, n_maps=4, keys_per_map=60, n_submaps=3, n_lists=4, list_len=80, deflate=True, seed=0xD0C9)[0])

        def stitched(stage):
            frags, offsets = [], [0]
            for r, e in enumerate(ranks):
                e.set_shard(r, world)
                stage(e)
                e.replay()
                buf = np.zeros(e.fragment_size(), dtype=np.uint8)
                assert e.export_fragment(buf.ctypes.data, buf.size, False) == buf.size
                frags.append(buf)
                offsets.append(offsets[-1] + buf.size)
            ranks[0].import_fragments(np.concatenate(frags), np.array(offsets, dtype=np.uint64))
            return ranks[0].patch_json(), [f.size for f in frags]

        for log in logs:
            single.load_changes(log); single.replay()
            got, _ = stitched(lambda e: e.load_changes(log))
            assert got == single.patch_json() == oracle_lib.OracleDoc(log).patch_json()
        # foreign rows are decoded as far as their object columns: calls that read whole rows refuse a sharded context
        for call in (ranks[-1].save, ranks[-1].rows):
            with pytest.raises(engine.UnsupportedChanges):
                call()
        for doc in docs:
            single.load_document(doc); single.replay()
            got, sizes = stitched(lambda e: e.load_document(doc))
            assert got == single.patch_json() == oracle_lib.OracleDoc.load_document(doc).patch_json()
        assert max(sizes) < 0.9 * sum(sizes)   # (the generated document: no rank holds nearly all records)
    finally:
        single.close()
        for e in ranks:
            e.close()


def test_fat_changes_counted_by_the_wavefront(eng, monkeypatch):
    """Changes of more than 4 KB on average go through k_parse_changes<true>: the rows of the action column and the sum of the predNum
    column by the whole wavefront (orbit of the record headers by pointer doubling) instead of two lanes walking the records. Same
    rows, same patch as the lean kernel and as the oracle; damaged columns (checksum repaired) are refused or give the oracle's patch."""
    log = loggen.config("c3_map_lww", 0.3)
    assert log.raw_bytes / log.n_changes > 4096
    want = oracle_lib.OracleDoc(log).patch_json()
    fat = emu_patch(eng, log)
    assert fat == want
    monkeypatch.setenv("AM355_PARSE_LEAN", "1")
    fat = emu_patch(eng, log)
    monkeypatch.delenv("AM355_PARSE_LEAN")
    assert fat == want
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    changes = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    equal, refused = mutation_util.column_mutations(lambda l: emu_patch(eng, l), rounds=60, changes=changes, seed=5)
    assert equal + refused == 60 and refused > 10
