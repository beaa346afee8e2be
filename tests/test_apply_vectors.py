"""Incremental patches of Backend.applyChanges: every applyChanges call the reference's own suites make
(new_backend_test, backend_test, test, text_test, table_test, sync_test, proxies_test, frontend_test: 1582 calls, captured from the
unmodified reference by oracle/make_apply_vectors.py) with the patch that call returned.  The oracle (oracle/am_oracle_apply.c,
a line-by-line restatement of mergeDocChangeOps / updatePatchProperty / setupPatches) is advanced call by call like the reference
was and must return the same patch text, JS property order included (only `clock` may list its keys differently); calls the
reference rejects must be rejected with the same message."""
import base64
import gzip
import json
import os

import pytest

import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))

# (Vector 475 gives a LOADED document a change it already holds, which the reference recognises after rebuilding the document's hash
# graph, new.js:1836-1840, test.js:1291-1302: decided by the oracle since round 4, with the hashes of the document's changes from
# tests/golden/ref_apply_vector_doc_hashes.json -- the oracle restates the scheduling, not the reconstruction of changes.)
ORACLE_LEAVES_OUT = set()


def _ordered(text):
    return json.loads(text, object_pairs_hook=lambda pairs: tuple(pairs))


def same_patch(got_text, want_text):
    got, want = dict(_ordered(got_text)), dict(_ordered(want_text))
    if list(got) != list(want):
        return False
    return all(dict(got[k]) == dict(want[k]) if k == "clock" else got[k] == want[k] for k in got)


def load_vectors():
    with open(os.path.join(HERE, "golden", "ref_apply_vectors.json.gz"), "rb") as f:
        d = json.loads(gzip.decompress(f.read()))
    return d["vectors"], [base64.b64decode(x) for x in d["pool"]]


def chains(vectors):
    """Every vector is a call that continues the session of its parent: yield root-to-leaf chains that cover every vector."""
    has_child = {v["parent"] for v in vectors}
    for leaf in range(len(vectors)):
        if leaf in has_child:
            continue
        chain, j = [], leaf
        while j != -1:
            chain.append(j)
            j = vectors[j]["parent"]
        yield chain[::-1]


def test_oracle_reproduces_every_applychanges_call_of_the_reference_suites():
    vectors, pool = load_vectors()
    checked, refused = set(), set()
    n_patches = n_errors = 0
    with open(os.path.join(HERE, "golden", "ref_apply_vector_doc_hashes.json")) as f:
        doc_hashes = {int(k): base64.b64decode(v) for k, v in json.load(f)["doc_hashes"].items()}
    for chain in chains(vectors):
        first = vectors[chain[0]]
        session = oracle_lib.OracleSession(pool[first["doc"]], doc_hashes[first["doc"]]) if "doc" in first else oracle_lib.OracleSession()
        for j in chain:
            v = vectors[j]
            batch = [pool[k] for k in v["changes"]]
            try:
                got = session.apply(batch, v["local"])
            except oracle_lib.OracleError as e:
                if j not in checked:
                    checked.add(j)
                    if str(e).startswith("unsupported"):
                        refused.add(j)
                    else:
                        assert "error" in v, f"vector {j}: the oracle rejects what the reference accepts: {e}"
                        assert str(e) == v["error"], f"vector {j}"
                        n_errors += 1
                break
            if j in checked:
                continue
            checked.add(j)
            assert "patch" in v, f"vector {j}: the oracle accepts what the reference rejects ({v.get('error')})"
            assert same_patch(got, v["patch"]), f"vector {j}:\n{got}\n{v['patch']}"
            n_patches += 1
    assert len(checked) == len(vectors)
    assert refused == ORACLE_LEAVES_OUT
    assert n_patches > 1500 and n_errors >= 4


def test_whole_document_patch_after_a_session_equals_the_bulk_replay():
    """Backend.getPatch of a document advanced call by call == the bulk replay of the same changes (the oracle's two paths)."""
    from automerge_classic_amd.loggen import ChangeLog
    vectors, pool = load_vectors()
    done = 0
    for chain in chains(vectors):
        if "doc" in vectors[chain[0]] or any("error" in vectors[j] for j in chain) or len(chain) > 40:
            continue
        session = oracle_lib.OracleSession()
        blobs = []
        for j in chain:
            batch = [pool[k] for k in vectors[j]["changes"]]
            session.apply(batch, vectors[j]["local"])
            blobs += batch
        try:
            want = oracle_lib.OracleDoc(ChangeLog.from_changes(blobs)).patch_json()
        except oracle_lib.OracleError:
            continue
        got = session.patch_json()
        assert dict(_ordered(got))["diffs"] == dict(_ordered(want))["diffs"]
        done += 1
        if done >= 300:
            break
    assert done >= 200


def test_oracle_follows_the_reference_through_sessions_with_counters_inside_lists():
    """tests/golden/apply_campaign_quirks.json.gz (oracle/js/make_list_quirk_golden.js): applyChanges sessions of the live reference on lists
    that hold counters, increments and deleted counters -- from empty documents and onto documents saved and loaded with such lists."""
    with open(os.path.join(HERE, "golden", "apply_campaign_quirks.json.gz"), "rb") as f:
        d = json.loads(gzip.decompress(f.read()))
    pool = [base64.b64decode(x) for x in d["pool"]]
    equal = 0
    for s in d["sessions"]:
        session = oracle_lib.OracleSession(base64.b64decode(s["doc"]), base64.b64decode(s["doc_hashes"])) if "doc" in s else oracle_lib.OracleSession()
        for ci, (call, want) in enumerate(zip(s["calls"], s["patches"])):
            assert not isinstance(want, dict), f"{s['name']} call {ci}: the reference rejects this batch"
            got = session.apply([pool[k] for k in call])
            assert same_patch(got, want), f"{s['name']} call {ci}:\n{got}\n{want}"
            equal += 1
    assert equal == sum(len(s["calls"]) for s in d["sessions"]) >= 450


def test_oracle_follows_the_reference_onto_loaded_documents():
    """Backend.load + applyChanges sessions recorded from the live reference (tests/golden/apply_campaign_loaded.json.gz: 48 sessions,
    every one also with the hash graph rebuilt by a query before the first call). A BackendDoc made by load schedules against the
    document's heads until a round applies nothing, then rebuilds the hash graph into an index that lacks what the running call has
    applied so far (new.js:1822-1841, 1887-1912): the oracle restates that loop; the hashes of the document's changes, which it does
    not rebuild itself, come from the fixture (`doc_hashes`, the reference's getAllChanges(load(doc))). Every patch must be the
    reference's -- `pendingChanges`, `clock` and `deps` of the calls around the rebuild included."""
    with open(os.path.join(HERE, "golden", "apply_campaign_loaded.json.gz"), "rb") as f:
        d = json.loads(gzip.decompress(f.read()))
    pool = [base64.b64decode(x) for x in d["pool"]]
    equal = 0
    for s in d["sessions"]:
        session = oracle_lib.OracleSession(base64.b64decode(s["doc"]), base64.b64decode(s["doc_hashes"]), bool(s.get("graph")))
        for ci, (call, want) in enumerate(zip(s["calls"], s["patches"])):
            try:
                got = session.apply([pool[k] for k in call])
            except oracle_lib.OracleError as e:
                assert isinstance(want, dict), f"{s['name']} call {ci}: the oracle rejects what the reference accepts: {e}"
                break
            assert not isinstance(want, dict), f"{s['name']} call {ci}: the oracle accepts what the reference rejects"
            assert same_patch(got, want), f"{s['name']} call {ci}:\n{got}\n{want}"
            equal += 1
    assert equal == 681
    # without the hashes the oracle says so instead of guessing
    s = next(x for x in d["sessions"] if x["name"] == "m:52:4:160:3#1")
    session = oracle_lib.OracleSession(base64.b64decode(s["doc"]))
    with pytest.raises(oracle_lib.OracleError, match="unsupported"):
        for call in s["calls"]:
            session.apply([pool[k] for k in call])
