"""Golden fixtures: outputs of the unmodified reference JS backend (see oracle/js/make_golden.js)."""
import base64
import glob
import json
import os

from automerge_classic_amd.loggen import ChangeLog

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _all_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.json")))


def fixture_names():
    """Fixtures holding binary changes and the reference's loadChanges + getPatch result."""
    return [n for n in _all_names() if not n.startswith("synthetic_doc_") and n not in ("save_generated", "doc_history", "doc_history_longkey", "bloom_filters", "ref_apply_vector_doc_hashes", "list_quirks", "headline_pin")]


def save_digest_cases():
    """Length + SHA-256 of the block-size-patched reference's Backend.save on deterministic generated logs (oracle/make_save_golden.py)."""
    with open(os.path.join(GOLDEN_DIR, "save_generated.json")) as f:
        return json.load(f)["cases"]


def history_golden():
    """Backend.getAllChanges(Backend.load(doc)) of the unmodified reference per document (oracle/make_history_golden.py): digests, or
    the error it throws."""
    with open(os.path.join(GOLDEN_DIR, "doc_history.json")) as f:
        return json.load(f)


def longkey_history_golden():
    """Reference-made documents whose long map keys are overwritten by many changes (oracle/make_longkey_history_golden.py)."""
    import base64
    with open(os.path.join(GOLDEN_DIR, "doc_history_longkey.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases.values():
        c["doc_bytes"] = base64.b64decode(c["doc"])
    return cases


def history_digests(arena, offsets, hashes):
    """The two digests of doc_history.json over an engine result (arena, offsets, hashes)."""
    import hashlib
    import struct
    a = bytes(bytearray(arena))
    h = hashlib.sha256()
    for i in range(len(offsets) - 1):
        lo, hi = int(offsets[i]), int(offsets[i + 1])
        h.update(struct.pack("<I", hi - lo))
        h.update(a[lo:hi])
    return h.hexdigest(), hashlib.sha256(bytes(bytearray(hashes))).hexdigest()


def list_quirk_cases():
    """Counters / visible rows without a value inside lists: patches of the unmodified reference (oracle/js/make_list_quirk_golden.js).
    -> [(name, [change bytes], patch, document bytes, patch after load)]"""
    with open(os.path.join(GOLDEN_DIR, "list_quirks.json")) as f:
        cases = json.load(f)["cases"]
    return [(c["name"], [base64.b64decode(x) for x in c["changes"]], c["patch"], base64.b64decode(c["doc"]), c["load_patch"]) for c in cases]


def same_patch(got_text, want_text):
    """JSON.stringify-exact (property order of every object included) except for the key order of `clock`, which records the order of
    application."""
    order = lambda text: json.loads(text, object_pairs_hook=lambda pairs: tuple(pairs))  # noqa: E731
    got, want = dict(order(got_text)), dict(order(want_text))
    if list(got) != list(want):
        return False
    return all(dict(got[k]) == dict(want[k]) if k == "clock" else got[k] == want[k] for k in got)


# the one case left to the JS path (a counter whose increments have all been deleted: DESIGN.md §6); `link` ops are columns the
# engine's save() / history do not model (patches are served)
LIST_QUIRK_REFUSED = {"hand_increment_deleted"}
LIST_QUIRK_NO_SAVE = {"hand_link_on_element", "hand_link_inserted"}


def check_list_quirk_cases(eng, engine_module):
    """Every case of list_quirks.json through an engine: patch after the replay, Backend.save bytes, patch after load of the reference's
    document, the changes rebuilt from it -- all against the unmodified reference. Returns (served, refused names)."""
    served, refused = 0, []
    for name, blobs, patch, doc, load_patch in list_quirk_cases():
        try:
            eng.load_changes(ChangeLog.from_changes(blobs, name=name))
            eng.replay()
        except engine_module.UnsupportedChanges:
            refused.append(name)
            with __import__("pytest").raises(engine_module.UnsupportedChanges):
                eng.load_document(doc)
                eng.replay()
            continue
        assert same_patch(eng.patch_json(), patch), name
        if name in LIST_QUIRK_NO_SAVE:
            with __import__("pytest").raises(engine_module.UnsupportedChanges):
                eng.save()
        else:
            assert bytes(eng.save()) == doc, f"{name}: saved document differs from the reference's"
        eng.load_document(doc)
        eng.replay()
        assert same_patch(eng.patch_json(), load_patch), name + " (load)"
        if name not in LIST_QUIRK_NO_SAVE:
            arena, offs, _ = eng.doc_changes(deflate=True)
            a = bytes(bytearray(arena))
            assert sorted(a[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)) == sorted(blobs), name + " (history)"
        served += 1
    return served, refused


def doc_fixture_names():
    """Fixtures holding a saved document and the reference's load + getPatch result."""
    out = []
    for n in _all_names():
        if n == "list_quirks":
            continue
        with open(os.path.join(GOLDEN_DIR, n + ".json")) as f:
            if "doc" in json.load(f):
                out.append(n)
    return out


def load_fixture(name):
    with open(os.path.join(GOLDEN_DIR, name + ".json")) as f:
        fx = json.load(f)
    if "changes" in fx:
        changes = [base64.b64decode(c) for c in fx["changes"]]
        fx["log"] = ChangeLog.from_changes(changes, name=name)
        # the patch the engine/oracle must reproduce: the stock reference, unless its block-boundary defect fired
        fx["expected"] = fx["patch"] if fx.get("stock_equals_bigblock", True) else fx["patch_bigblock"]
    if "doc" in fx:
        fx["doc_bytes"] = base64.b64decode(fx["doc"])
        fx["expected_load"] = fx["load_patch"] if fx.get("stock_equals_bigblock", True) else fx["load_patch_bigblock"]
    return fx


def defect_fixture():
    """tests/golden/defect_block_boundary.json (oracle/make_defect_fixture.py): inputs on which the STOCK reference's
    block-boundary defect fires -- its answer depends on delivery order -- next to the block-size-patched reference's."""
    fx = load_fixture("defect_block_boundary")
    changes = [base64.b64decode(c) for c in fx["changes"]]
    fx["log_reversed"] = ChangeLog.from_changes([changes[i] for i in fx["order_reversed"]], name="defect_block_boundary+reversed")
    return fx


def headline_pin_cases():
    """tests/golden/headline_pin.json (oracle/make_headline_pin.py): digests of the BLOCK-SIZE-PATCHED reference's getPatch / save on the
    headline shape at 124,801 ops (c4_text_single x0.125), in the generator's delivery order and in bench.py's shuffled one.
    Yields (case, log): the logs are regenerated here (loggen.config is deterministic), only digests are committed."""
    import json
    import os
    import numpy as np
    from automerge_classic_amd import loggen
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "headline_pin.json")) as f:
        pin = json.load(f)
    base = loggen.config(pin["workload"], pin["scale"], False)
    perm = np.random.default_rng(int(pin["seed"], 16) & 0xFFFF).permutation(base.n_changes)
    logs = {"in_order": base, "shuffled": base.reordered(perm)}
    for case in pin["cases"]:
        assert case["n_ops"] == base.n_ops and case["n_changes"] == base.n_changes
        yield case, logs[case["order"]]
