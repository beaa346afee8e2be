"""Damaged-change campaigns shared by the emulated and the GPU suite: one byte of one change of a reference-made fixture is changed, the
chunk checksum repaired, and the batch goes through the engine and through the oracle. The engine may refuse more than the oracle (the JS
host then runs the reference path) but never accepts what the oracle rejects and never produces another patch."""
import base64
import hashlib
import json
import os
import random

import golden_util
import oracle_lib
from automerge_classic_amd import engine, loggen


# (equal, refused) of the two campaigns below -- they are deterministic (fixed seeds, fixed fixtures); every accepted batch has been
# compared with the oracle's patch, every refusal is the engine's right. A change of either number means the engine refuses or accepts
# something else than before: look at it, then update the pair.
COLUMN_CAMPAIGN = (16, 54)
HEADER_CAMPAIGN = (67, 64)


def _changes(fixture):
    with open(os.path.join(golden_util.GOLDEN_DIR, fixture + ".json")) as f:
        return [base64.b64decode(c) for c in json.load(f)["changes"]]


def _check(patch_fn, changes, ci, ch, pos, what):
    ch[4:8] = hashlib.sha256(bytes(ch[8:])).digest()[:4]
    log = loggen.ChangeLog.from_changes(changes[:ci] + [bytes(ch)] + changes[ci + 1:], name="mutated")
    try:
        want = oracle_lib.OracleDoc(log).patch_json()
    except oracle_lib.OracleError:
        want = None
    try:
        got = patch_fn(log)
    except engine.EngineError:
        got = None
    if got is None:
        return False
    assert want is not None, f"engine accepted a change the oracle rejects ({what}, change {ci}, byte {pos})"
    assert got == want, f"different patch ({what}, change {ci}, byte {pos})"
    return True


def column_mutations(patch_fn, rounds=70, changes=None, seed=11):
    """Damage in the op columns (the last 60 % of a change)."""
    changes = changes or _changes("frontend_mixed_3actors")
    rng = random.Random(seed)
    equal = refused = 0
    for _ in range(rounds):
        ci = rng.randrange(len(changes))
        ch = bytearray(changes[ci])
        pos = rng.randrange(9 + int((len(ch) - 9) * 0.4), len(ch))
        ch[pos] = rng.randrange(256)
        if _check(patch_fn, changes, ci, ch, pos, "columns"):
            equal += 1
        else:
            refused += 1
    return equal, refused


def header_mutations(patch_fn, rounds=60):
    """Damage in the header (dependencies, actor, seq, startOp, time, message, other actors, column directory): the first 45 % of a change."""
    rng = random.Random(29)
    equal = refused = 0
    for fixture in ("frontend_mixed_3actors", "frontend_text_8actors", "hand_conflicts"):
        changes = _changes(fixture)
        for _ in range(rounds):
            ci = rng.randrange(len(changes))
            ch = bytearray(changes[ci])
            if ch[8] != 1:
                continue  # (a DEFLATEd change: its header is inside the compressed stream)
            hi = max(11, 9 + int((len(ch) - 9) * 0.45))
            pos = rng.randrange(9, min(hi, len(ch)))
            ch[pos] = rng.randrange(256) if rng.random() < 0.5 else ch[pos] ^ (1 << rng.randrange(8))
            if _check(patch_fn, changes, ci, ch, pos, fixture):
                equal += 1
            else:
                refused += 1
    return equal, refused


def _uleb(b, o):
    v = s = 0
    while True:
        x = b[o]
        o += 1
        v |= (x & 0x7f) << s
        s += 7
        if not x & 0x80:
            return v, o


def _enc_uleb(v):
    out = bytearray()
    while True:
        x = v & 0x7f
        v >>= 7
        out.append(x | (0x80 if v else 0))
        if not v:
            return bytes(out)


def with_message_and_actor(change, message, actor=None):
    """The same (uncompressed) change with another commit message and, optionally, another author: chunk length and checksum
    rewritten (columnar.js:635-652 header layout: deps, actor, seq, startOp, time, message, ...). Used to give changes every
    length modulo the SHA-256 block size."""
    import hashlib
    assert change[8] == 1
    _, body0 = _uleb(change, 9)
    body = bytearray(change[body0:])
    o = 0
    ndeps, o = _uleb(body, o)
    o += 32 * ndeps
    alen, o = _uleb(body, o)
    if actor is not None:
        assert len(actor) == alen
        body[o:o + alen] = actor
    o += alen
    for _ in range(3):  # seq, startOp, time (signed, but one LEB128 number either way)
        _, o = _uleb(body, o)
    mlen, o2 = _uleb(body, o)
    body[o:o2 + mlen] = _enc_uleb(len(message)) + message
    chunk = bytes([1]) + _enc_uleb(len(body)) + bytes(body)
    return change[:4] + hashlib.sha256(chunk).digest()[:4] + chunk
