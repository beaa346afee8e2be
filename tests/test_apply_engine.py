"""Backend.applyChanges with its incremental patch on the engine (include/am355.h am355_apply_changes).

Checkers: the patches the unmodified reference returned (tests/golden/ref_apply_vectors.json.gz: every applyChanges call of the
reference's own suites; tests/golden/apply_campaign.json.gz: sessions on random multi-actor documents made with the real
frontend) and the sequential oracle (oracle/am_oracle_apply.c) on generated logs split into batches.  The engine must return the
reference's patch text (JS property order included; `clock` may order its keys differently) or refuse the call
(AM355_E_UNSUPPORTED: the JS path serves it) -- never a different patch, and it must reject what the reference rejects.

CPU suite (`-m "not gpu"`): the kernels under the HIP-runtime emulation of tests/emu on a slice of the vectors.  GPU suite
(`-m gpu`): everything, full-size logs included."""
import base64
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from automerge_classic_amd import engine, loggen
from automerge_classic_amd.loggen import ChangeLog
from test_apply_vectors import chains, load_vectors, same_patch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libam355_emu.so")


def _ordered(text):
    return json.loads(text, object_pairs_hook=lambda pairs: tuple(pairs))


def _same(got, want, local=False):
    if not local:
        return same_patch(got, want)
    g, w = dict(_ordered(got)), dict((k, v) for k, v in _ordered(want) if k not in ("actor", "seq"))
    return list(g) == list(w) and all(dict(g[k]) == dict(w[k]) if k == "clock" else g[k] == w[k] for k in g)


# The applyChanges calls of the reference's own suites (sessions that start from an empty document or from a loaded one): exactly
# which ones the engine serves, refuses (JS path) and rejects like the reference -- by vector id, as the whole-document twin does
# (tests/test_ref_suite_vectors.py). Refused: none any more. (Through round 4: 15, 28, 507, 508 = two ops on one list element in one
# merge call -- now one event per run of a call, kd_events --, and 501, 619 = an increment of a counter inside a list -- now the
# whole-document patch of such lists, k_quirk_rows, and the counter rules of the delta stage, quirk_elem_state.)
# All 1582 captured calls are accounted for: 1578 served and equal, none refused, 4 rejected (GPU: every chain, the 600-call one
# included; emulation: chains of <= 50 calls, 978 served).
SUITE_EQUAL, SUITE_EQUAL_SHORT, SUITE_REFUSED, SUITE_REJECTED = 1578, 978, [], 4


def run_vector_chains(make_engine, max_chain, max_chains=None):
    """Every call of every chain (sessions that start from an empty document) through a fresh engine context.
    Returns (equal, refused ids, rejected-like-the-reference)."""
    vectors, pool = load_vectors()
    checked, refused = set(), set()
    equal = rejected = 0
    done = 0
    for chain in chains(vectors):
        if len(chain) > max_chain:
            continue
        if max_chains is not None and done >= max_chains:
            break
        done += 1
        eng = make_engine()
        try:
            if "doc" in vectors[chain[0]]:  # the session starts from Backend.load(doc): the batch goes onto the loaded document
                eng.load_document(pool[vectors[chain[0]]["doc"]])
                eng.replay()
            for j in chain:
                v = vectors[j]
                try:
                    eng.apply_changes(ChangeLog.from_changes([pool[k] for k in v["changes"]]))
                    got = eng.apply_patch_json()
                except engine.UnsupportedChanges:
                    if j not in checked:
                        checked.add(j)
                        refused.add(j)
                    break
                except engine.InvalidChanges:
                    if j not in checked:
                        checked.add(j)
                        assert "error" in v, f"vector {j}: the engine rejects what the reference accepts"
                        rejected += 1
                    break
                if j in checked:
                    continue
                checked.add(j)
                assert "patch" in v, f"vector {j}: the engine accepts what the reference rejects ({v.get('error')})"
                assert _same(got, v["patch"], v["local"]), f"vector {j}:\n{got}\n{v['patch']}"
                equal += 1
        finally:
            eng.close()
    return equal, refused, rejected


def load_campaign(name="apply_campaign.json.gz"):
    with open(os.path.join(HERE, "golden", name), "rb") as f:
        d = json.loads(gzip.decompress(f.read()))
    return d["sessions"], [base64.b64decode(x) for x in d["pool"]]


def run_campaign(make_engine, names=None, fixture="apply_campaign.json.gz"):
    sessions, pool = load_campaign(fixture)
    equal = refused = 0
    for s in sessions:
        if names is not None and s["name"] not in names:
            continue
        eng = make_engine()
        given = []
        try:
            if "doc" in s:  # the session goes onto a LOADED document (Backend.load(doc), then applyChanges)
                eng.load_document(base64.b64decode(s["doc"]))
                eng.replay()
                if s.get("graph"):  # the reference had been asked for the document's changes before the first call (new.js:1922)
                    assert eng.hash_graph_known(True)
            for ci, (call, want) in enumerate(zip(s["calls"], s["patches"])):
                batch = [pool[k] for k in call]
                given += batch
                try:
                    eng.apply_changes(ChangeLog.from_changes(batch))
                    got = eng.apply_patch_json()
                except engine.UnsupportedChanges:
                    # the host serves this call -- and the rest of the session -- on the JS path: what the reference's objectMeta holds
                    # after a call the engine did not follow is history the engine has no record of
                    refused += 1
                    break
                assert not isinstance(want, dict), f"{s['name']} call {ci}: the reference rejects this batch"
                assert same_patch(got, want), f"{s['name']} call {ci}:\n{got}\n{want}"
                equal += 1
        finally:
            eng.close()
    return equal, refused


def split_log(log, n_batches):
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    changes = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    per = (len(changes) + n_batches - 1) // n_batches
    return [changes[i:i + per] for i in range(0, len(changes), per)]


def check_against_oracle_session(eng, batches):
    """The engine's patch of every batch == the oracle's, and the state afterwards == the bulk replay."""
    session = oracle_lib.OracleSession()
    for k, batch in enumerate(batches):
        want = session.apply(batch)
        eng.apply_changes(ChangeLog.from_changes(batch))
        got = eng.apply_patch_json()
        assert same_patch(got, want), f"batch {k}:\n{got[:3000]}\n{want[:3000]}"
    assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"]


def interpret_edits(patch_text, texts):
    """Applies the list edits of an incremental patch to `texts` ({objectId: list}) like the frontend does (apply_patch.js)."""
    def visit(node):
        if "edits" in node:
            cur = texts.setdefault(node["objectId"], [])
            for e in node["edits"]:
                if e["action"] == "insert":
                    cur.insert(e["index"], e["value"].get("value"))
                elif e["action"] == "multi-insert":
                    cur[e["index"]:e["index"]] = e["values"]
                elif e["action"] == "remove":
                    del cur[e["index"]:e["index"] + e["count"]]
                elif e["action"] == "update":
                    cur[e["index"]] = e["value"].get("value")
        for vals in node.get("props", {}).values():
            for v in vals.values():
                if isinstance(v, dict) and "objectId" in v:
                    visit(v)
    visit(json.loads(patch_text)["diffs"])


# ---------------------------------------------------------------------------------------------------------------------------
# CPU: kernels under emulation
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return EMU_LIB


def test_reference_suite_calls_emulated(emu_lib):
    """The same chains and the same bar as test_reference_suite_calls_gpu, through the CPU emulation of the kernels."""
    equal, refused, rejected = run_vector_chains(lambda: engine.Engine(0, emu_lib), max_chain=50)
    assert (equal, sorted(refused), rejected) == (SUITE_EQUAL_SHORT, SUITE_REFUSED, SUITE_REJECTED)


def test_campaign_sessions_emulated(emu_lib):
    """Sessions of tests/golden/apply_campaign.json.gz (24 sessions / 447 calls; every second one here, all on the GPU) through the CPU
    emulation of the kernels: every call served, every patch the live reference's. (tests/golden/apply_campaign_lists.json.gz: a slice
    in test_list_assignment_sessions_emulated, all 412 calls on the GPU.)"""
    sessions, _ = load_campaign()
    equal, refused = run_campaign(lambda: engine.Engine(0, emu_lib), names={s["name"] for s in sessions[::2]})   # (every second session; all of them on the GPU)
    assert equal == 180 and refused == 0


@pytest.mark.parametrize("kind,kw,n_batches", [
    (loggen.KIND_TEXT_CONCURRENT, dict(n_actors=6, n_rounds=4, ins_per_change=30, del_per_change=9, n_objects=1), 3),
    (loggen.KIND_TEXT_CONCURRENT, dict(n_actors=5, n_rounds=3, ins_per_change=12, del_per_change=5, n_objects=4), 5),
    (loggen.KIND_MAP_LWW, dict(n_actors=6, n_rounds=3, n_keys=60), 4),
    (loggen.KIND_TEXT_TYPING, dict(n_ops=500, ops_per_change=20), 6),
    (loggen.KIND_TEXT_CONCURRENT, dict(n_actors=8, n_rounds=4, ins_per_change=60, del_per_change=15, n_objects=2), 2),  # > 1024 edit items per batch: level-by-level partitions
])
def test_generated_logs_in_batches_match_the_oracle_emulated(emu_lib, kind, kw, n_batches):
    log = loggen.generate(kind, seed=23, **kw)
    eng = engine.Engine(0, emu_lib)
    try:
        check_against_oracle_session(eng, split_log(log, n_batches))
    finally:
        eng.close()


def test_partition_variants_agree_emulated(emu_lib, monkeypatch):
    """The dominance counts of the list edits by their four versions -- all partition levels in LDS in a workgroup of 256 (<= 1024 items),
    pairs of 256-item tiles over the whole device (kd_dom_tiles / kd_dom_cross, <= 65536 items), all levels in LDS in a workgroup of 1024
    with the items held once (<= 11264 items, kd_partition_lds_big), three launches per level -- give the oracle's patches, on batches of a
    few hundred, ~1200 and ~4800 items (two list objects, deletions)."""
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=8, n_rounds=4, ins_per_change=60, del_per_change=15, n_objects=2, seed=29)
    big = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=16, n_rounds=4, ins_per_change=120, del_per_change=30, n_objects=2, seed=31)
    texts = []
    # (+ the second half of the stage by one workgroup without a host round trip, kd_edit_small, against the launches it stands for, and
    #  the stage with its map kernels launched for batches the in-place list merge knows to be free of map rows)
    for env in ({}, {"AM355_DELTA_NO_TILES": "1"}, {"AM355_DELTA_NO_TILES": "1", "AM355_DELTA_NO_BIG_LDS": "1"}, {"AM355_DELTA_NO_LDS": "1"},
                {"AM355_DELTA_NO_SMALL": "1", "AM355_DELTA_ALL_KERNELS": "1"}):
        for k in ("AM355_DELTA_NO_TILES", "AM355_DELTA_NO_BIG_LDS", "AM355_DELTA_NO_LDS", "AM355_DELTA_NO_SMALL", "AM355_DELTA_ALL_KERNELS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        engs = [engine.Engine(0, emu_lib) for _ in range(3)]
        try:
            check_against_oracle_session(engs[0], split_log(log, 2))
            check_against_oracle_session(engs[1], split_log(log, 16))
            check_against_oracle_session(engs[2], split_log(big, 2))
            texts.append((engs[0].patch_json(), engs[2].patch_json()))
        finally:
            for e in engs:
                e.close()
    assert texts[0] == texts[1] == texts[2] == texts[3] == texts[4]


def test_batch_of_exactly_1024_edit_items_emulated(emu_lib):
    """A batch whose list edits are EXACTLY the 1024 items the single-workgroup partition holds (partition_lds_block, four items per thread
    of 256): the prefix entry behind the last item was nobody's to write, and the group's zero count came from stale LDS
    (tools/soak_resident.py, seed 7019: every edit of the batch at one index, in position order)."""
    log = loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=1229, ops_per_change=41, seed=7019)
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    ch = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    eng = engine.Engine(0, emu_lib)
    try:
        check_against_oracle_session(eng, [ch[:6], ch[6:31]])
    finally:
        eng.close()


def _one_by_one(log, head):
    """`head` changes as the first batch, then every following change as a batch of its own."""
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    changes = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    return [changes[:head]] + [[c] for c in changes[head:]]


def test_resident_state_small_batches_emulated(emu_lib, monkeypatch):
    """Backend.applyChanges change by change onto the state the context holds (am355_replay.hip replay_resident: only the batch is
    parsed, hashed, scheduled, decoded and resolved): every incremental patch and the final getPatch equal the oracle session's, the
    resident path really served the calls, and a batch it cannot take -- a new actor, a dependency that is not applied yet -- goes
    through the full replay with the same result. Text with deletions, two list objects, and a map with conflicts."""
    cases = [
        (loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=5, n_rounds=6, ins_per_change=14, del_per_change=5, n_objects=2, seed=61), 5),
        (loggen.generate(loggen.KIND_MAP_LWW, n_actors=4, n_rounds=5, n_keys=40, seed=62), 4),
        (loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=400, ops_per_change=10, seed=63), 3),
    ]
    for log, head in cases:
        batches = _one_by_one(log, head)
        eng = engine.Engine(0, emu_lib)
        try:
            check_against_oracle_session(eng, batches)
            served, fell_back, in_place = eng.resident_counters()
            assert served >= len(batches) - 3 and fell_back <= 1, (served, fell_back, len(batches))
        finally:
            eng.close()
    # the same calls with the resident path switched off give the same document
    log, head = cases[0]
    monkeypatch.setenv("AM355_NO_RESIDENT", "1")
    eng = engine.Engine(0, emu_lib)
    try:
        check_against_oracle_session(eng, _one_by_one(log, head))
        assert eng.resident_counters() == (0, 0, 0)
    finally:
        eng.close()
    monkeypatch.delenv("AM355_NO_RESIDENT")
    # new actors arrive one by one (the ranks of the kept rows change: full replay each time), then steady state again
    late = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=6, n_rounds=3, ins_per_change=9, del_per_change=3, n_objects=1, seed=64)
    eng = engine.Engine(0, emu_lib)
    try:
        check_against_oracle_session(eng, _one_by_one(late, 1))
        served, fell_back, _ = eng.resident_counters()
        assert fell_back >= 4 and served >= 8, (served, fell_back)
    finally:
        eng.close()
    # a change delivered before its dependency: queued by the full path, applied by a later call; the calls after that are resident again
    arena, offs = bytes(late.arena), [int(x) for x in late.offsets]
    ch = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    batches = [ch[:8], [ch[9]], [ch[8]], [ch[10]], [ch[11]]] + [[c] for c in ch[12:]]
    eng = engine.Engine(0, emu_lib)
    try:
        check_against_oracle_session(eng, batches)
        assert eng.resident_counters()[0] >= len(ch) - 14
    finally:
        eng.close()


def _changes_of(log):
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    return [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]


def mixed_document_batches(seed, text_kw, map_kw, held_text):
    """A document that holds a Text AND root-map keys: the changes of a concurrent-text log and of a map log (two sets of actors, two
    histories side by side -- neither depends on the other). The first batch holds the text log but its last `held_text` changes and the
    first round of the map log (every actor known from then on); then map changes, text changes and pairs of both take turns."""
    text = _changes_of(loggen.generate(loggen.KIND_TEXT_CONCURRENT, seed=seed, **text_kw))
    maps = _changes_of(loggen.generate(loggen.KIND_MAP_LWW, seed=seed + 1, **map_kw))
    na = map_kw["n_actors"]
    batches = [text[:len(text) - held_text] + maps[:na]]
    t, m = len(text) - held_text, na
    k = 0
    while t < len(text) or m < len(maps):
        turn = k % 4
        if turn in (0, 1) and m < len(maps):
            n = 1 if turn == 0 else min(3, len(maps) - m)
            batches.append(maps[m:m + n]); m += n                       # map rows only
        elif turn == 2 and t < len(text):
            batches.append([text[t]]); t += 1                           # list rows only
        elif t < len(text) and m < len(maps):
            batches.append([text[t], maps[m]]); t += 1; m += 1          # both: the full merge
        elif m < len(maps):
            batches.append([maps[m]]); m += 1
        else:
            batches.append([text[t]]); t += 1
        k += 1
    return batches


def test_resident_map_batches_leave_the_lists_alone_emulated(emu_lib, monkeypatch):
    """A batch of plain map rows onto a kept state (replay_resident, merge_run_maps): the map half of the merge alone -- no list kernel
    runs, the stored order stays --, in turn with list-only batches merged in place and batches of both (list rows merged in place, then
    the map half; with the path switched off: the whole merge). Every
    incremental patch, getPatch in between and at the end (which rebuilds the stale edit tables and, AM355_RESORDER_VERIFY, compares the
    order) equal the oracle session's; and the same calls with the path switched off."""
    monkeypatch.setenv("AM355_RESORDER_VERIFY", "1")
    batches = mixed_document_batches(81, dict(n_actors=4, n_rounds=6, ins_per_change=15, del_per_change=4, n_objects=2),
                                     dict(n_actors=3, n_rounds=12, n_keys=25), held_text=12)
    for off in (False, True):
        if off:
            monkeypatch.setenv("AM355_NO_MAPS_ONLY", "1")
        eng = engine.Engine(0, emu_lib)
        session = oracle_lib.OracleSession()
        try:
            for i, batch in enumerate(batches):
                want = session.apply(batch)
                eng.apply_changes(ChangeLog.from_changes(batch))
                assert same_patch(eng.apply_patch_json(), want), f"batch {i}"
                if i % 7 == 3:
                    assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"], f"getPatch after batch {i}"
            assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"]
            served, fell_back, in_place = eng.resident_counters()
            maps_only = eng.resident_maps_only_calls()
            if off:
                assert maps_only == 0
            else:
                assert maps_only >= 10 and in_place >= 5 and served >= len(batches) - 3, (served, fell_back, in_place, maps_only, len(batches))
            doc = bytes(eng.save())
            back = oracle_lib.OracleSession(doc)
            assert dict(_ordered(back.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"]
            back.close()
        finally:
            eng.close()
            session.close()


def test_resident_batches_that_fail_behind_the_device_work_emulated(emu_lib, monkeypatch):
    """The batch's hashes, the duplicate check and the dependency check run on the host BEHIND the enqueue of its decode / resolution /
    list merge (replay_resident, hashes_and_dependencies): a batch that fails there has already changed the kept arrays, and the full
    replay that follows must start from the staged bytes. Text merged in place: a batch that repeats an applied change among new ones,
    a batch whose second change depends on a change not delivered yet, a batch that holds one change twice -- each followed by calls
    the resident path serves again; every incremental patch and the final document equal the oracle session's. And a change whose
    checksum is wrong is rejected like the reference rejects it (the context then starts over, as after any rejected call)."""
    monkeypatch.setenv("AM355_RESORDER_VERIFY", "1")
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=4, n_rounds=8, ins_per_change=12, del_per_change=3, n_objects=1, seed=71)
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    ch = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    assert len(ch) >= 30
    batches = [ch[:10], [ch[10]], [ch[11]],
               [ch[3], ch[12], ch[13]],          # an applied change again, then two new ones
               [ch[14]],
               [ch[15], ch[17]],                 # (same actor's next change first: 17 depends on 16) -> queued by the full path
               [ch[16]],
               [ch[18]], [ch[19]],
               [ch[20], ch[20], ch[21]],         # one change twice in a batch
               [ch[22]], [ch[23]]] + [[c] for c in ch[24:]]
    eng = engine.Engine(0, emu_lib)
    try:
        check_against_oracle_session(eng, batches)
        served, fell_back, in_place = eng.resident_counters()
        assert fell_back >= 3 and served >= len(batches) - 8 and in_place >= 5, (served, fell_back, in_place, len(batches))
    finally:
        eng.close()
    # a wrong checksum in the middle of a batch
    bad = bytearray(ch[12]); bad[5] ^= 0x40
    eng = engine.Engine(0, emu_lib)
    session = oracle_lib.OracleSession()
    try:
        for batch in (ch[:10], [ch[10]], [ch[11]]):
            want = session.apply(batch)
            eng.apply_changes(ChangeLog.from_changes(batch))
            assert same_patch(eng.apply_patch_json(), want)
        with pytest.raises(Exception):
            eng.apply_changes(ChangeLog.from_changes([bytes(bad), ch[13]]))
        # (a rejected call leaves the context without a state, whichever path rejected it -- include/am355.h: the caller delivers the
        #  document's changes again, as js/index.js does)
        fresh = oracle_lib.OracleSession()
        want = fresh.apply(ch[:14])
        eng.apply_changes(ChangeLog.from_changes(ch[:14]))
        assert same_patch(eng.apply_patch_json(), want)
        for batch in ([ch[14]], [ch[15]]):
            want = fresh.apply(batch)
            eng.apply_changes(ChangeLog.from_changes(batch))
            assert same_patch(eng.apply_patch_json(), want)
        assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(fresh.patch_json()))["diffs"]
        assert eng.resident_counters()[0] >= 4
    finally:
        eng.close()


@pytest.mark.parametrize("chunk", [0, 5])
def test_resident_list_order_merged_in_place_emulated(emu_lib, monkeypatch, chunk):
    """(chunk = 5: AM355_RESORDER_CHUNK -- a batch is merged five rows at a time, each chunk against the order the chunks in front left,
    the order ping-ponging between its two arrays: what a batch of more than 4096 rows goes through.)
    am355_resorder.hip: the new elements of a small list-only batch are ranked against the STORED order (forward scan for the first
    smaller id behind the reference element, roots of one gap by descending id, typing runs behind their roots). Change by change on
    concurrent text edits -- insertions at the same spots by several actors, deletions, two objects --; after every third call the
    whole-document patch is asked for, which rebuilds the tables from scratch and (AM355_RESORDER_VERIFY) compares the order computed
    from scratch with the one the in-place merges left; every incremental patch and every getPatch equal the oracle session's."""
    monkeypatch.setenv("AM355_RESORDER_VERIFY", "1")
    if chunk:
        monkeypatch.setenv("AM355_RESORDER_CHUNK", str(chunk))
    logs = [
        loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=6, n_rounds=5, ins_per_change=9, del_per_change=3, n_objects=2, seed=71),
        loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=4, n_rounds=8, ins_per_change=3, del_per_change=1, n_objects=1, seed=72),   # short runs: many roots per object
        loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=300, ops_per_change=7, seed=73),
    ]
    for log in logs:
        arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
        ch = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
        head = max(2, len(ch) // 4)
        batches = [ch[:head]]
        k, size = head, 1
        while k < len(ch):
            batches.append(ch[k:k + size])
            k += size
            size = size % 3 + 1
        eng = engine.Engine(0, emu_lib)
        session = oracle_lib.OracleSession()
        try:
            for i, batch in enumerate(batches):
                want = session.apply(batch)
                eng.apply_changes(ChangeLog.from_changes(batch))
                assert same_patch(eng.apply_patch_json(), want), f"batch {i}"
                if i % 3 == 2:
                    assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"], f"getPatch after batch {i}"
            assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"]
            doc = eng.save()
            eng2 = engine.Engine(0, emu_lib)
            eng2.load_changes(log)
            eng2.replay()
            assert bytes(doc) == bytes(eng2.save())    # Backend.save of the state the in-place merges built == of the bulk replay
            eng2.close()
            served, fell_back, in_place = eng.resident_counters()
            assert in_place >= (len(batches) - 1) // 2, (served, fell_back, in_place, len(batches))
        finally:
            eng.close()


def test_resident_new_elements_of_two_objects_sharing_a_gap_emulated(emu_lib, monkeypatch):
    """The end of one list object is the first position of the next: a batch that appends to the first and inserts at the head of the
    second has new elements with the SAME gap in two objects -- the object in front first, whatever the ids (found by
    tools/soak_resident.py, seed 2118: kr_order ranked them by id alone and the merged order interleaved the objects)."""
    monkeypatch.setenv("AM355_RESORDER_VERIFY", "1")
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=5, n_rounds=6, ins_per_change=1, del_per_change=2, n_objects=2, seed=2118)
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    ch = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    batches, k = [], 0
    for size in (1, 1, 1, 2, 3, 8, 13, 2):
        batches.append(ch[k:k + size])
        k += size
    assert k >= len(ch)
    eng = engine.Engine(0, emu_lib)
    session = oracle_lib.OracleSession()
    try:
        for i, batch in enumerate(b for b in batches if b):
            want = session.apply(batch)
            eng.apply_changes(ChangeLog.from_changes(batch))
            assert same_patch(eng.apply_patch_json(), want), f"batch {i}"
            assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"], f"getPatch after batch {i}"
        assert eng.resident_counters()[2] >= 2
    finally:
        eng.close()


def test_batches_behind_the_staged_changes_or_restaged_emulated(emu_lib, monkeypatch):
    """A batch onto a state whose changes are all applied is staged behind them (only the batch is copied); AM355_APPLY_RESTAGE=1
    rebuilds the whole queue instead, as a call with queued changes does. Same patches either way, deflated batches included."""
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=5, n_rounds=4, ins_per_change=14, del_per_change=4, n_objects=2, seed=41)
    zlog = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=5, n_rounds=4, ins_per_change=14, del_per_change=4, n_objects=2, seed=41, deflate=True)
    plain, packed = split_log(log, 5), split_log(zlog, 5)
    batches = [packed[k] if k % 2 else plain[k] for k in range(len(plain))]
    texts = []
    for restage in (False, True):
        if restage:
            monkeypatch.setenv("AM355_APPLY_RESTAGE", "1")
        eng = engine.Engine(0, emu_lib)
        try:
            check_against_oracle_session(eng, batches)
            texts.append(eng.patch_json())
        finally:
            eng.close()
    assert texts[0] == texts[1]


def test_apply_after_load_changes_and_queue_emulated(emu_lib):
    """loadChanges + replay, then applyChanges on top; a batch delivered out of order waits in the queue (new.js:1822-1841)."""
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=4, n_rounds=4, ins_per_change=10, del_per_change=3, n_objects=1, seed=5)
    batches = split_log(log, 4)
    eng = engine.Engine(0, emu_lib)
    session = oracle_lib.OracleSession()
    try:
        eng.load_changes(ChangeLog.from_changes(batches[0]))
        eng.replay()
        session.apply(batches[0])
        # batch 2 arrives before batch 1: everything in it waits; batch 1 then releases it
        for batch in (batches[2], batches[1], batches[3]):
            want = session.apply(batch)
            eng.apply_changes(ChangeLog.from_changes(batch))
            assert same_patch(eng.apply_patch_json(), want)
        assert json.loads(eng.apply_patch_json())["pendingChanges"] == 0
    finally:
        eng.close()


def test_unserved_batch_is_refused_emulated(emu_lib):
    """A batch outside the served subset ends in a refusal that names the reason, not in a patch: an increment of a counter inside a
    list that another actor has deleted meanwhile (the reference then goes on counting the element as visible while its patch says
    `remove`: DESIGN.md 5, tests/golden/list_quirks.json hand_counter_deleted_and_incremented). (Through round 4 this test used vector
    15 of the reference's suites -- two assignments to one list element in one change --, which is served since round 5.)"""
    import golden_util
    blobs = next(c[1] for c in golden_util.list_quirk_cases() if c[0] == "hand_counter_deleted_and_incremented")
    eng = engine.Engine(0, emu_lib)
    try:
        eng.apply_changes(ChangeLog.from_changes(blobs[:2]))   # the list with its counter; the deletion
        with pytest.raises(engine.UnsupportedChanges, match="neither a value nor"):
            eng.apply_changes(ChangeLog.from_changes(blobs[2:]))   # the concurrent increment
    finally:
        eng.close()


def test_several_ops_of_one_merge_call_on_one_list_element_emulated(emu_lib):
    """Vectors 15, 28, 507, 508 of the reference's suites (two assignments to one list element in one change: one merge call, one visit of
    the element, new.js:1092-1118) -- refused through round 4, served since the runs of a call report as one event (kd_events)."""
    vectors, pool = load_vectors()
    for j in (15, 28):   # (sessions of one call; 507 / 508 are later calls of sessions: test_reference_suite_calls_emulated)
        eng = engine.Engine(0, emu_lib)
        try:
            eng.apply_changes(ChangeLog.from_changes([pool[k] for k in vectors[j]["changes"]]))
            assert _same(eng.apply_patch_json(), vectors[j]["patch"], vectors[j]["local"])
        finally:
            eng.close()


def test_edits_inside_objects_that_are_no_longer_visible_emulated(emu_lib):
    """Nested documents whose objects are overwritten and deleted while other actors still edit them: setupPatches needs what the
    reference's objectMeta.children holds for the parent property, which the device replays from the history of the rows on it
    (delta_key_history). Three sessions of the campaign that this decides, served to the end."""
    equal, refused = run_campaign(lambda: engine.Engine(0, emu_lib), names={"m:12:4:160:3#1", "m:16:6:140:2#2", "21:3:70:2#0"})
    assert equal == 10 + 16 + 37 and refused == 0


def test_state_replayed_in_one_go_is_served_or_refused_emulated(emu_lib):
    """The JS host replays the retained changes of a state into a fresh context when the old one has moved on: one call where the
    reference had many. The engine is told (am355_forget_call_history) and must then still return the reference's patch -- or refuse
    the call where the patch depends on where the reference's calls ended -- never a different one."""
    sessions, pool = load_campaign()
    equal = refused = 0
    for name in ("m:16:6:140:2#2", "21:3:70:2#0", "m:15:2:100:3#1"):
        s = next(x for x in sessions if x["name"] == name)
        given = []
        eng = None
        try:
            for ci, (call, want) in enumerate(zip(s["calls"], s["patches"])):
                batch = [pool[k] for k in call]
                if ci % 7 == 3:  # the context "moved on": everything given so far again, in one go
                    if eng is not None:
                        eng.close()
                    eng = engine.Engine(0, emu_lib)
                    eng.load_changes(ChangeLog.from_changes(given))
                    eng.replay()
                    eng.forget_call_history()
                elif eng is None:
                    eng = engine.Engine(0, emu_lib)
                given += batch
                try:
                    eng.apply_changes(ChangeLog.from_changes(batch))
                    got = eng.apply_patch_json()
                except engine.UnsupportedChanges:
                    refused += 1
                    break
                assert same_patch(got, want), f"{name} call {ci}:\n{got}\n{want}"
                equal += 1
        finally:
            if eng is not None:
                eng.close()
    assert equal >= 30


def test_loaded_lineage_replayed_in_one_go_emulated(emu_lib):
    """A state whose lineage began with Backend.load, replayed into a fresh context from its retained changes (the JS host does that
    when the context has moved on): the rebuilt changes of the document + what the calls since applied + what is queued, in one go,
    with am355_forget_call_history(ctx, number of document changes). Told whether the reference has rebuilt the hash graph by then
    (am355_hash_graph_known, which the host reads after every call) the engine goes on exactly like the context that made the calls;
    not told, it serves a call only when both answers give the same schedule -- never a different patch."""
    sessions, pool = load_campaign("apply_campaign_loaded.json.gz")
    told_equal = untold_equal = untold_refused = 0
    for name in ("m:51:3:120:2#2", "m:52:4:160:3#1", "m:55:2:100:3#1", "l:57:4:120:10#0", "m:56:6:140:2#2+g", "61:3:70:2#1"):
        s = next(x for x in sessions if x["name"] == name)
        doc = base64.b64decode(s["doc"])
        for tell in (True, False):
            eng = engine.Engine(0, emu_lib)
            try:
                eng.load_document(doc)
                eng.replay()
                if s.get("graph"):
                    eng.hash_graph_known(True)
                arena, offs, _ = eng.doc_changes(deflate=False)
                history = [bytes(arena[int(offs[i]):int(offs[i + 1])]) for i in range(len(offs) - 1)]
                listed, n_applied = list(history), len(history)  # the engine's list of changes, as the JS host keeps it: applied ++ queued
                for ci, (call, want) in enumerate(zip(s["calls"], s["patches"])):
                    batch = [pool[k] for k in call]
                    if ci % 3 == 1:  # the context "moved on"
                        known = eng.hash_graph_known()
                        eng.close()
                        eng = engine.Engine(0, emu_lib)
                        # (the APPLIED changes only: a change queued because the reference forgot a hash while it rebuilt the graph must
                        # not be applied before the next call -- it is handed over behind that call's batch, new.js:1822)
                        eng.load_changes(ChangeLog.from_changes(listed[:n_applied]))
                        eng.replay()
                        eng.forget_call_history(len(history))
                        if tell:
                            assert eng.hash_graph_known(known) == known
                        batch = batch + listed[n_applied:]
                        listed = listed[:n_applied]
                    try:
                        eng.apply_changes(ChangeLog.from_changes(batch))
                        got = eng.apply_patch_json()
                    except engine.UnsupportedChanges:
                        assert not tell, f"{name} call {ci}: refused although the host told what it knows"
                        untold_refused += 1
                        break
                    assert not isinstance(want, dict) and same_patch(got, want), f"{name} call {ci} (told: {tell}):\n{got}\n{want}"
                    told_equal += tell
                    untold_equal += not tell
                    applied, pending = eng.applied(), eng.pending()
                    full = listed[:n_applied] + batch + listed[n_applied:]  # (the engine's list: applied before ++ batch ++ queued before)
                    listed, n_applied = [full[i] for i in applied] + [full[i] for i in pending], len(applied)
            finally:
                eng.close()
    assert told_equal >= 60 and untold_equal >= 20, (told_equal, untold_equal, untold_refused)


def test_list_assignment_sessions_emulated(emu_lib):
    """Lists whose elements are assigned to (`list[i] = v`), concurrently, against deletions, inside one merge call with the
    reference's index lag (oracle/js/apply_campaign.js listScenario): a slice of tests/golden/apply_campaign_lists.json.gz."""
    equal, refused = run_campaign(lambda: engine.Engine(0, emu_lib), names={"l:33:4:120:0#2", "l:34:3:100:10#1", "l:36:2:80:25#2"},
                                  fixture="apply_campaign_lists.json.gz")
    assert equal == 19 + 9 + 8 and refused == 0


def test_wide_conflict_sessions_emulated(emu_lib):
    """Every actor assigns the SAME list elements (and map keys) in one change each, delivered call by call or in one call
    (oracle/js/apply_campaign.js conflictScenario -> tests/golden/apply_campaign_conflicts.json.gz): an update edit holds one record
    per visible value, so a call yields up to 5 edit records per op row -- more than the per-row edit table of the delta stage held
    (device buffer overflow in kd_edit_pack; the table now grows from the record bound kd_events publishes)."""
    equal, refused = run_campaign(lambda: engine.Engine(0, emu_lib), fixture="apply_campaign_conflicts.json.gz")
    assert equal == 35 and refused == 0


def test_sessions_onto_loaded_documents_emulated(emu_lib):
    """Backend.applyChanges onto Backend.load(doc) (tests/golden/apply_campaign_loaded.json.gz, oracle/make_apply_campaign.py LOADED_SPECS:
    the first calls of a campaign session are saved and loaded again by the live reference, the recorded calls go onto the loaded
    document). The engine rebuilds the document's changes on the device, schedules the batch the way the reference does while it has
    not rebuilt the hash graph (the document's heads are all it knows; a round that applies nothing makes it rebuild the graph and
    forget what the call applied so far, new.js:1822-1841 -- visible as `pendingChanges` in these patches) and takes objectMeta from
    one pass over the document's rows. Every session also with the graph rebuilt by a query before the first call ("+g":
    am355_hash_graph_known). Every call served, every patch the live reference's (48 sessions / 681 calls on the GPU, half of them here)."""
    sessions, _ = load_campaign("apply_campaign_loaded.json.gz")
    half = {s["name"] for s in sessions[::2]}   # (every second session here: with and without "+g" of all eight generators; the GPU test runs all)
    equal, refused = run_campaign(lambda: engine.Engine(0, emu_lib), names=half, fixture="apply_campaign_loaded.json.gz")
    assert equal >= 300 and refused == 0


QUIRK_EQUAL, QUIRK_REFUSED = 316, 19


def test_sessions_with_counters_inside_lists_emulated(emu_lib):
    """Lists that hold counters and rows without a value (tests/golden/apply_campaign_quirks.json.gz, oracle/js/make_list_quirk_golden.js:
    24 sessions from empty documents, 18 onto documents the reference saved and loaded with such lists in them; patches of the live
    reference). The whole-document patch serves these lists since round 5; the INCREMENTAL patch of a call that increments a counter
    inside a list, or assigns to an element that holds one, stays with the JS path (the session ends there) -- everything else is
    served, the calls onto the loaded documents included, and no served call differs."""
    equal, refused = run_campaign(lambda: engine.Engine(0, emu_lib), fixture="apply_campaign_quirks.json.gz")
    assert (equal, refused) == (QUIRK_EQUAL, QUIRK_REFUSED)


# ---------------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_sessions_with_counters_inside_lists_gpu():
    equal, refused = run_campaign(lambda: engine.Engine(0), fixture="apply_campaign_quirks.json.gz")
    assert (equal, refused) == (QUIRK_EQUAL, QUIRK_REFUSED)


@pytest.mark.gpu
def test_sessions_onto_loaded_documents_gpu():
    equal, refused = run_campaign(lambda: engine.Engine(0), fixture="apply_campaign_loaded.json.gz")
    assert equal == 681 and refused == 0


@pytest.mark.gpu
def test_wide_conflict_sessions_gpu():
    equal, refused = run_campaign(lambda: engine.Engine(0), fixture="apply_campaign_conflicts.json.gz")
    assert equal == 35 and refused == 0


@pytest.mark.gpu
def test_reference_suite_calls_gpu():
    equal, refused, rejected = run_vector_chains(lambda: engine.Engine(0), max_chain=1 << 30)
    assert (equal, sorted(refused), rejected) == (SUITE_EQUAL, SUITE_REFUSED, SUITE_REJECTED)


@pytest.mark.gpu
def test_campaign_sessions_gpu():
    equal, refused = run_campaign(lambda: engine.Engine(0))
    assert equal == 447 and refused == 0


@pytest.mark.gpu
def test_list_assignment_sessions_gpu():
    """All 18 sessions / 412 calls of tests/golden/apply_campaign_lists.json.gz: every call served, every patch the reference's."""
    equal, refused = run_campaign(lambda: engine.Engine(0), fixture="apply_campaign_lists.json.gz")
    assert equal == 412 and refused == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,scale,n_batches", [
    ("c4_text_single", 0.1, 5), ("c4_text_multi", 0.1, 3), ("c3_map_lww", 1.0, 4), ("c2_text_typing", 1.0, 7), ("c4_text_single", 1.0, 4),
])
def test_bench_workloads_in_batches_match_the_oracle_gpu(name, scale, n_batches):
    log = loggen.config(name, scale)
    eng = engine.Engine(0)
    try:
        check_against_oracle_session(eng, split_log(log, n_batches))
    finally:
        eng.close()


@pytest.mark.gpu
def test_resident_map_batches_leave_the_lists_alone_gpu(monkeypatch):
    """test_resident_map_batches_leave_the_lists_alone_emulated's calls on the GPU, on a larger document (a 60 k-op text, 2 objects, and
    150 map changes): map-only batches run the map half of the merge alone, list-only batches merge in place, mixed ones take the whole
    merge; every incremental patch and the whole-document patch every 25 calls equal the oracle session's."""
    monkeypatch.setenv("AM355_RESORDER_VERIFY", "1")
    batches = mixed_document_batches(83, dict(n_actors=8, n_rounds=10, ins_per_change=600, del_per_change=150, n_objects=2),
                                     dict(n_actors=6, n_rounds=25, n_keys=300), held_text=30)
    eng = engine.Engine(0)
    session = oracle_lib.OracleSession()
    try:
        for i, batch in enumerate(batches):
            want = session.apply(batch)
            eng.apply_changes(ChangeLog.from_changes(batch))
            assert same_patch(eng.apply_patch_json(), want), f"batch {i}"
            if i % 25 == 24:
                assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"], f"getPatch after batch {i}"
        assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"]
        served, fell_back, in_place = eng.resident_counters()
        assert eng.resident_maps_only_calls() >= 40 and in_place >= 10 and fell_back <= 2, (served, fell_back, in_place, eng.resident_maps_only_calls(), len(batches))
    finally:
        eng.close()
        session.close()


@pytest.mark.gpu
def test_resident_state_200_small_batches_gpu():
    """200 small batches (1-3 changes) onto a 100 k-op Text document of 64 actors, interleaved with getPatch calls:
    every incremental patch AND the whole-document patch after every 25th batch equal the oracle session's; the resident path served
    them (am355_replay.hip replay_resident)."""
    log = loggen.config("c4_text_single", 0.125)
    arena, offs = bytes(log.arena), [int(x) for x in log.offsets]
    ch = [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    head = len(ch) - 400
    batches, k, size = [ch[:head]], head, 1
    while len(batches) < 201:
        batches.append(ch[k:k + size])
        k += size
        size = size % 3 + 1
    eng = engine.Engine(0)
    session = oracle_lib.OracleSession()
    try:
        for i, batch in enumerate(batches):
            want = session.apply(batch)
            eng.apply_changes(ChangeLog.from_changes(batch))
            assert same_patch(eng.apply_patch_json(), want), f"batch {i}"
            if i % 25 == 0:
                assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"], f"getPatch after batch {i}"
        assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"]
        served, fell_back, in_place = eng.resident_counters()
        assert served >= 195 and fell_back <= 2 and in_place >= 150, (served, fell_back, in_place)
    finally:
        eng.close()


@pytest.mark.gpu
def test_full_size_edits_rebuild_the_text_gpu():
    """Size-independent property: the edits of the incremental patches, applied one batch after the other like the frontend applies
    them, build the text the whole-document patch of the final state holds (1 M-op headline log in 8 batches)."""
    log = loggen.config("c4_text_single", 1.0)
    eng = engine.Engine(0)
    try:
        texts = {}
        for batch in split_log(log, 8):
            eng.apply_changes(ChangeLog.from_changes(batch))
            interpret_edits(eng.apply_patch_json(), texts)
        final = {}
        interpret_edits(eng.patch_json(), final)
        assert texts == final and sum(len(v) for v in final.values()) > 500_000
    finally:
        eng.close()


# ---------------------------------------------------------------------------------------------------------------------------
# sync protocol, bulk side: dependency graph and Bloom filters over the resident hashes
# ---------------------------------------------------------------------------------------------------------------------------
def _bloom_probes(h, n_bits, num_probes):
    """BloomFilter.getProbes (backend/sync.js:85-100) restated."""
    x = int.from_bytes(h[0:4], "little") % n_bits
    y = int.from_bytes(h[4:8], "little") % n_bits
    z = int.from_bytes(h[8:12], "little") % n_bits
    probes = [x]
    for _ in range(1, num_probes):
        x = (x + y) % n_bits
        y = (y + z) % n_bits
        probes.append(x)
    return probes


def _bloom_bits(hashes):
    """`new BloomFilter(hashes).bits` (sync.js:41-47, 105-109)."""
    n_bytes = (len(hashes) * 10 + 7) // 8
    bits = bytearray(n_bytes)
    for h in hashes:
        for p in _bloom_probes(h, 8 * n_bytes, 7):
            bits[p >> 3] |= 1 << (p & 7)
    return bytes(bits)


def _parse_deps(change):
    """Dependency hashes of an uncompressed binary change (columnar.js:635-640)."""
    off, length, shift = 9, 0, 0
    while True:
        b = change[off]
        off += 1
        length |= (b & 0x7f) << shift
        shift += 7
        if not b & 0x80:
            break
    n = change[off]   # (fewer than 128 dependencies in these logs: one LEB128 byte)
    assert n < 0x80
    off += 1
    return [bytes(change[off + 32 * k: off + 32 * k + 32]) for k in range(n)]


def check_sync_pieces(eng):
    log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=9, n_rounds=5, ins_per_change=6, del_per_change=2, n_objects=2, seed=77)
    eng.load_changes(log)
    eng.replay()
    hashes = [bytes(h) for h in eng.hashes()]
    arena, offs = eng.raw()
    arena = bytes(arena)
    n = len(offs) - 1
    first, index = eng.dep_graph()
    by_hash = {}
    for i, h in enumerate(hashes):
        by_hash.setdefault(h, i)
    for i in range(n):
        deps = _parse_deps(arena[int(offs[i]):int(offs[i + 1])])
        assert [by_hash.get(d, 0xffffffff) for d in deps] == [int(x) for x in index[first[i]:first[i + 1]]]
    # Bloom filters: the BYTES the unmodified reference builds over these hashes and what its containsHash answers for every hash
    # (tests/golden/bloom_filters.json, oracle/make_bloom_golden.py) -- the log is the fixture's, its hashes are checked first
    with open(os.path.join(HERE, "golden", "bloom_filters.json")) as f:
        gold = json.load(f)
    assert gold["n_changes"] == n and [h.hex() for h in hashes] == gold["hashes"]
    everything = np.arange(n, dtype=np.uint32)
    for idx, filter_hex, contains in zip(gold["sets"], gold["filters"], gold["contains"]):
        want = bytes.fromhex(filter_hex)
        bits = eng.bloom_build(np.asarray(idx, dtype=np.uint32))
        if not idx:
            assert want == b"" and len(bits) == 0
            continue
        # header of the reference's encoding: numEntries, BITS_PER_ENTRY = 10, NUM_PROBES = 7 as LEB128 (sync.js:66-74)
        head = bytearray()
        v = len(idx)
        while True:
            head.append((v & 0x7f) | (0x80 if v >> 7 else 0))
            v >>= 7
            if not v:
                break
        head += bytes([10, 7])
        assert want[:len(head)] == bytes(head) and bytes(bits) == want[len(head):], f"filter over {len(idx)} hashes differs from the reference's bytes"
        got = eng.bloom_probe(everything, len(idx), 10, 7, np.frombuffer(want[len(head):], dtype=np.uint8))
        assert [int(x) for x in got] == contains and all(got[i] for i in idx)
    # sizes the fixture does not hold: the restatement of sync.js:87-109 above (already pinned by the fixture's sets)
    rng = np.random.default_rng(5)
    for size in (3, n // 3):
        idx = rng.choice(n, size=size, replace=False).astype(np.uint32)
        assert bytes(eng.bloom_build(idx)) == _bloom_bits([hashes[i] for i in idx])
    assert not eng.bloom_probe(np.arange(n, dtype=np.uint32), 0, 0, 0, np.zeros(0, np.uint8)).any()   # an empty filter contains nothing


def test_sync_dep_graph_and_bloom_filters_emulated(emu_lib):
    eng = engine.Engine(0, emu_lib)
    try:
        check_sync_pieces(eng)
    finally:
        eng.close()


@pytest.mark.gpu
def test_sync_dep_graph_and_bloom_filters_gpu():
    eng = engine.Engine(0)
    try:
        check_sync_pieces(eng)
    finally:
        eng.close()
