"""The reference's own test suites as parity vectors: every sequence of binary changes a reference test feeds a BackendDoc, with
the patch the unmodified reference then reports (1477 change vectors, 18 saved documents, 4 rejected batches; captured by
oracle/make_ref_suite_vectors.py from new_backend_test, backend_test, test, text_test, table_test, sync_test, proxies_test,
frontend_test). The reference applied the changes in several calls, the bulk replay applies them in one: the patches are compared
with their JS property order (`JSON.stringify`-exact `diffs`, key order of every object included -- SURVEY hard part 2); only
the key order of `clock` may differ (it records application order, which differs between one batch and several calls).
The vectors the engine does not serve are named one by one: nothing else may be refused. 1471 vectors also carry length + SHA-256 of the
reference's Backend.save of the same changes applied to a fresh document in one batch: am355_save must produce those bytes."""
import base64
import gzip
import hashlib
import json
import os
import subprocess

import pytest

import oracle_lib
from automerge_classic_amd import engine
from automerge_classic_amd.loggen import ChangeLog

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU_DIR = os.path.join(HERE, "emu")


# vector ids (position in the file) the engine must not serve, and why
REJECTED_BY_REFERENCE = {14: "BAD_PRED", 16: "BAD_PRED", 25: "BAD_ELEM", 33: "BAD_ELEM"}   # the reference throws on these batches too
LEFT_TO_JS_PATH = {}   # (rounds 1-4: 56 and 911 -- a counter inside a list; served since round 5)
SAVE_LEFT_TO_JS_PATH = {82}                                 # replay served, save() refused: a change carries columns the engine does not model


def _ordered(text):
    """JSON text -> nested tuples that keep the property order of every object (dict equality would ignore it)."""
    return json.loads(text, object_pairs_hook=lambda pairs: tuple(pairs))


def _same_patch(got_text, want_text):
    got, want = dict(_ordered(got_text)), dict(_ordered(want_text))
    if [k for k in got] != [k for k in want]:
        return False
    for k in got:
        if k == "clock":
            if dict(got[k]) != dict(want[k]):
                return False
        elif got[k] != want[k]:
            return False
    return True


def _vectors():
    with open(os.path.join(HERE, "golden", "ref_suite_vectors.json.gz"), "rb") as f:
        d = json.loads(gzip.decompress(f.read()))
    pool = [base64.b64decode(x) for x in d["pool"]]
    return [(v, [pool[k] for k in v["changes"]]) for v in d["vectors"]]


def test_oracle_reproduces_every_reference_suite_vector():
    n = {"changes": 0, "doc": 0, "reject": 0}
    for v, blobs in _vectors():
        if v["kind"] == "reject":
            with pytest.raises(oracle_lib.OracleError):
                oracle_lib.OracleDoc(ChangeLog.from_changes(blobs)).patch_json()
        elif v["kind"] == "doc":
            assert _same_patch(oracle_lib.OracleDoc.load_document(blobs[0]).patch_json(), v["patch"])
        else:
            assert _same_patch(oracle_lib.OracleDoc(ChangeLog.from_changes(blobs)).patch_json(), v["patch"])
        n[v["kind"]] += 1
    assert n["changes"] > 1400 and n["doc"] >= 18 and n["reject"] >= 4


def _run_engine(eng, vectors):
    """vectors: [(id, vector, blobs)]. Returns (ids equal to the reference, {refused id: flag names}, ids whose save() was refused, loaded count);
    raises on a differing patch or on an accepted batch the reference rejects."""
    equal, refused, save_refused = [], {}, []
    loaded = 0
    for i, v, blobs in vectors:
        try:
            if v["kind"] == "doc":
                eng.load_document(blobs[0])
            else:
                eng.load_changes(ChangeLog.from_changes(blobs))
            eng.replay()
            got = eng.patch_json()
        except engine.EngineError as e:
            refused[i] = (type(e).__name__, e.flag_names)   # reference rejects it too, or legal input left to the JS path
            continue
        assert v["kind"] != "reject", f"vector {i}: the engine accepted a batch the reference rejects ({v['error']})"
        assert _same_patch(got, v["patch"]), f"vector {i} ({v['kind']}, {len(blobs)} blobs): patch differs from the reference"
        if "doc_sha256" in v:
            # Backend.save(Backend.loadChanges(Backend.init(), changes)) of the reference, by digest
            try:
                doc = eng.save()
            except engine.UnsupportedChanges:
                doc = None   # e.g. changes with columns the engine does not model: their document is saved by the JS path
                save_refused.append(i)
            if doc is not None:
                assert len(doc) == v["doc_len"] and hashlib.sha256(doc).hexdigest() == v["doc_sha256"], f"vector {i}: saved document differs"
                # ... and those bytes (= the reference's document) loaded again are the same document
                try:
                    eng.load_document(doc)
                    eng.replay()
                    again = eng.patch_json()
                except engine.UnsupportedChanges:
                    again = None   # (a document the load path leaves to the JS backend; `loaded` below bounds how many)
                if again is not None:
                    assert _same_patch(again, v["patch"]), f"vector {i}: Backend.load of the saved document gives another patch"
                    loaded += 1
        equal.append(i)
    return equal, refused, save_refused, loaded


def _check_refusals(ids, refused, save_refused):
    ids = set(ids)
    want = {i: f for i, f in {**REJECTED_BY_REFERENCE, **LEFT_TO_JS_PATH}.items() if i in ids}
    assert set(refused) == set(want), f"refused vectors {sorted(refused)} != expected {sorted(want)}: {refused}"
    for i, (cls, flags) in refused.items():
        assert want[i] in flags, (i, cls, flags)
        assert cls == ("InvalidChanges" if i in REJECTED_BY_REFERENCE else "UnsupportedChanges"), (i, cls)
    assert set(save_refused) == SAVE_LEFT_TO_JS_PATH & ids, save_refused


def test_engine_emulation_on_a_sample_of_reference_suite_vectors():
    """Every fifth vector (plus all documents, all rejects and every vector the engine is expected to refuse) through the CPU
    emulation of the kernels; the GPU suite runs them all."""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    special = set(REJECTED_BY_REFERENCE) | set(LEFT_TO_JS_PATH) | SAVE_LEFT_TO_JS_PATH | {56, 911}   # (56, 911: a counter inside a list)
    sample = [(i, v, b) for i, (v, b) in enumerate(_vectors()) if i % 5 == 0 or v["kind"] != "changes" or i in special]
    eng = engine.Engine(0, os.path.join(EMU_DIR, "libam355_emu.so"))
    try:
        equal, refused, save_refused, loaded = _run_engine(eng, sample)
    finally:
        eng.close()
    _check_refusals([i for i, _, _ in sample], refused, save_refused)
    assert len(equal) == len(sample) - len(refused) and loaded >= len(sample) - 40


@pytest.mark.gpu
def test_engine_on_every_reference_suite_vector():
    vs = [(i, v, b) for i, (v, b) in enumerate(_vectors())]
    eng = engine.Engine(0)
    try:
        equal, refused, save_refused, loaded = _run_engine(eng, vs)
    finally:
        eng.close()
    # exactly: the 4 batches the reference rejects, no legal input left to the JS path; one save() left to the JS path
    _check_refusals(range(len(vs)), refused, save_refused)
    assert len(equal) == len(vs) - 4 and loaded >= 1400
