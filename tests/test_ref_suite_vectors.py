"""The reference's own test suites as parity vectors: every sequence of binary changes a reference test feeds a BackendDoc, with
the patch the unmodified reference then reports (1477 change vectors, 18 saved documents, 4 rejected batches; captured by
oracle/make_ref_suite_vectors.py from new_backend_test, backend_test, test, text_test, table_test, sync_test, proxies_test,
frontend_test). The reference applied the changes in several calls, the bulk replay applies them in one: the patch objects are
compared as objects (`clock` key order is the one thing allowed to differ). 1471 vectors also carry length + SHA-256 of the
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
            assert json.loads(oracle_lib.OracleDoc.load_document(blobs[0]).patch_json()) == json.loads(v["patch"])
        else:
            assert json.loads(oracle_lib.OracleDoc(ChangeLog.from_changes(blobs)).patch_json()) == json.loads(v["patch"])
        n[v["kind"]] += 1
    assert n["changes"] > 1400 and n["doc"] >= 18 and n["reject"] >= 4


def _run_engine(eng, vectors):
    """Returns (equal, refused); raises on a differing patch or on an accepted batch the reference rejects."""
    equal = refused = 0
    loaded = [0]
    for i, (v, blobs) in enumerate(vectors):
        try:
            if v["kind"] == "doc":
                eng.load_document(blobs[0])
            else:
                eng.load_changes(ChangeLog.from_changes(blobs))
            eng.replay()
            got = json.loads(eng.patch_json())
        except engine.EngineError:
            refused += 1   # reference rejects it too, or legal input left to the JS path
            continue
        assert v["kind"] != "reject", f"vector {i}: the engine accepted a batch the reference rejects ({v['error']})"
        assert got == json.loads(v["patch"]), f"vector {i} ({v['kind']}, {len(blobs)} blobs): patch differs from the reference"
        if "doc_sha256" in v:
            # Backend.save(Backend.loadChanges(Backend.init(), changes)) of the reference, by digest
            try:
                doc = eng.save()
            except engine.UnsupportedChanges:
                doc = None   # e.g. changes with columns the engine does not model: their document is saved by the JS path
            if doc is not None:
                assert len(doc) == v["doc_len"] and hashlib.sha256(doc).hexdigest() == v["doc_sha256"], f"vector {i}: saved document differs"
                # ... and those bytes (= the reference's document) loaded again are the same document
                try:
                    eng.load_document(doc)
                    eng.replay()
                    again = json.loads(eng.patch_json())
                except engine.UnsupportedChanges:
                    again = None
                if again is not None:
                    assert again == json.loads(v["patch"]), f"vector {i}: Backend.load of the saved document gives another patch"
                    loaded[0] += 1
        equal += 1
    return equal, refused, loaded[0]


def test_engine_emulation_on_a_sample_of_reference_suite_vectors():
    """Every 12th vector (plus all documents and rejects) through the CPU emulation of the kernels; the GPU suite runs them all."""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    vs = _vectors()
    sample = [x for i, x in enumerate(vs) if i % 12 == 0 or x[0]["kind"] != "changes"]
    eng = engine.Engine(0, os.path.join(EMU_DIR, "libam355_emu.so"))
    try:
        equal, refused, loaded = _run_engine(eng, sample)
    finally:
        eng.close()
    assert equal >= len(sample) - 8 and refused <= 8 and loaded >= len(sample) - 40


@pytest.mark.gpu
def test_engine_on_every_reference_suite_vector():
    vs = _vectors()
    eng = engine.Engine(0)
    try:
        equal, refused, loaded = _run_engine(eng, vs)
    finally:
        eng.close()
    # 4 rejected batches + the couple of legal inputs the engine leaves to the JS path (counters in lists etc.)
    assert equal >= len(vs) - 10 and refused <= 10 and loaded >= 1400
