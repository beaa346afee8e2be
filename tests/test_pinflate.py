"""Chunked (parallel) raw-DEFLATE decode of one stream, csrc/am355_pinflate.cpp, against zlib -- the checker here is CPython's
zlib module (pako, which the reference calls in columnar.js:1062-1067, is a port of zlib). The decoder may give a stream up
(-1: the engine then runs the ordinary inflate); what it returns must be zlib's bytes."""
import ctypes
import os
import random
import subprocess
import zlib

import pytest

import oracle_lib
from automerge_classic_amd import engine, loggen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libam355_emu.so")


@pytest.fixture(scope="module")
def pinflate():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    lib = ctypes.CDLL(EMU_LIB)
    lib.am355_emu_pinflate.restype = ctypes.c_long
    lib.am355_emu_pinflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint]

    def run(comp, cap, chunk, threads):
        out = ctypes.create_string_buffer(max(cap, 1))
        n = lib.am355_emu_pinflate(comp, len(comp), out, cap, chunk, threads)
        return None if n < 0 else out.raw[:n]
    return run


def deflate_raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return co.compress(data) + co.flush()


def sample(kind, n, seed):
    rng = random.Random(seed)
    if kind == "random":
        return rng.randbytes(n)
    if kind == "low":
        return bytes(rng.choice(b"abcd") for _ in range(n))
    if kind == "zeros":
        return bytes(n)
    if kind == "keys":  # what a key column looks like: few thousand words, short binary in between
        words = [bytes(rng.randrange(97, 123) for _ in range(rng.randrange(3, 14))) for _ in range(3000)]
        out = bytearray()
        while len(out) < n:
            out += rng.choice(words)
            out.append(rng.randrange(20))
        return bytes(out[:n])
    # stretches of noise and of long repeats (distances up to the window, stored blocks at level 0)
    out = bytearray()
    while len(out) < n:
        out += rng.randbytes(rng.randrange(1, 40000)) if rng.random() < 0.5 else bytes([rng.randrange(256)]) * rng.randrange(1, 70000)
    return bytes(out[:n])


@pytest.mark.parametrize("kind", ["random", "low", "zeros", "keys", "mixed"])
def test_equals_zlib_on_every_block_type(pinflate, kind):
    for n in (0, 1, 300, 70000, 600000):
        data = sample(kind, n, n + 7)
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            comp = deflate_raw(data, level, strategy)
            for chunk, threads in ((4096, 1), (4096, 4), (50000, 3)):
                assert pinflate(comp, len(data) + 16, chunk, threads) == data, (kind, n, level, strategy, chunk, threads)


def test_chunks_are_really_used_and_the_cap_holds(pinflate):
    data = sample("keys", 3_000_000, 3)
    comp = deflate_raw(data)
    assert len(comp) > 20 * 8192
    assert pinflate(comp, len(data), 8192, 4) == data
    assert pinflate(comp, len(data) - 1, 8192, 4) is None   # beyond the cap: given up, never written past it
    assert pinflate(deflate_raw(data, 6, mem=9), len(data), 8192, 4) == data  # 32 K symbols per block


def test_zip_bombs_are_given_up_within_the_cap(pinflate):
    """the cap bounds the JOB's memory, inside the decode -- not each chunk at its block boundaries. 512 MB of zeros (0.5 MB
    compressed, 64 chunks that used to grow to the cap each) and one giant run inside few blocks must be given up after about `cap`
    symbols in total."""
    import resource
    cap = 8 << 20
    bomb = deflate_raw(bytes(512 << 20), 9, mem=9)
    assert len(bomb) < (1 << 20)
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert pinflate(bomb, cap, 8192, 4) is None
    assert pinflate(bomb, cap, len(bomb), 1) is None          # one chunk: the single-stream shape
    grown_mb = (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before) / 1024.0
    assert grown_mb < 160, grown_mb     # (cap symbols = 16 MB, the chunks' first allocations, the caller's 8 MB buffer; the old code: > 1 GB)
    small = deflate_raw(bytes(6 << 20), 9, mem=9)
    assert pinflate(small, cap, 2048, 4) == bytes(6 << 20)     # under the cap: decoded as before


def test_damaged_streams_are_given_up_or_decode_like_zlib(pinflate):
    data = sample("keys", 400000, 11)
    comp = deflate_raw(data)
    rng = random.Random(5)
    accepted = 0
    for t in range(300):
        c = bytearray(comp)
        if t % 3 == 0:
            c = c[:rng.randrange(len(c))]
        else:
            c[rng.randrange(len(c))] ^= 1 << rng.randrange(8)
        got = pinflate(bytes(c), 2 * len(data), 6000, 4)
        if got is None:
            continue
        do = zlib.decompressobj(-15)
        want = do.decompress(bytes(c))
        assert do.eof and got == want, t  # a stream zlib does not finish must not be accepted
        accepted += 1
    assert accepted > 20  # (a flipped bit inside a literal changes one byte of the output: legal DEFLATE)


@pytest.mark.parametrize("chunk", [700, 5000])
def test_document_columns_through_the_chunked_decode(chunk, monkeypatch):
    """Backend.load (columnar.js:1062-1067) with every compressed column of the document in chunks: same patch as the oracle's,
    same as with the path switched off; a damaged column is reported as before."""
    monkeypatch.setenv("AM355_PINFLATE_MIN", "1500")
    monkeypatch.setenv("AM355_PINFLATE_CHUNK", str(chunk))
    eng = engine.Engine(0, EMU_LIB)
    doc, rows = loggen.generate_document(n_actors=6, n_texts=3, text_len=900, n_maps=3, keys_per_map=300, n_submaps=2, n_lists=2, list_len=300, deflate=True, seed=0xD0C7)
    want = oracle_lib.OracleDoc.load_document(doc).patch_json()
    eng.load_document(doc)
    eng.replay()
    assert eng.stats().n_ops == rows and eng.patch_json() == want
    saved = eng.save()
    monkeypatch.setenv("AM355_PINFLATE", "0")
    eng.load_document(doc)
    eng.replay()
    assert eng.patch_json() == want and eng.save() == saved
    monkeypatch.delenv("AM355_PINFLATE")
    # damage inside the compressed columns, checksum repaired: rejected or the oracle's patch
    import hashlib
    rng = random.Random(chunk)
    verdicts = [0, 0]
    for _ in range(40):
        d = bytearray(doc)
        d[rng.randrange(len(d) // 2, len(d))] ^= 1 << rng.randrange(8)
        d[4:8] = hashlib.sha256(bytes(d[8:])).digest()[:4]
        try:
            w = oracle_lib.OracleDoc.load_document(bytes(d)).patch_json()
        except oracle_lib.OracleError:
            w = None
        try:
            eng.load_document(bytes(d))
            eng.replay()
            g = eng.patch_json()
        except engine.EngineError:
            g = None
        if g is not None:
            assert g == w
        verdicts[g is not None] += 1
    assert verdicts[0] > 5
    eng.close()


def test_backend_load_in_one_call_equals_the_two_calls(monkeypatch):
    """am355_backend_load (checksum thread beside the device stages, verdict at the end) against am355_load_document + am355_replay:
    same patch, same save bytes; a wrong checksum is reported by either form, and outranks damaged columns."""
    import hashlib
    monkeypatch.setenv("AM355_PINFLATE_MIN", "1500")
    monkeypatch.setenv("AM355_PINFLATE_CHUNK", "900")
    eng = engine.Engine(0, EMU_LIB)
    doc, rows = loggen.generate_document(n_actors=5, n_texts=3, text_len=700, n_maps=2, keys_per_map=200, n_submaps=2, n_lists=2, list_len=250, deflate=True, seed=0xD0C8)
    eng.load_document(doc)
    eng.replay()
    two = (eng.patch_json(), eng.save())
    eng.backend_load(doc)
    assert eng.stats().n_ops == rows
    assert (eng.patch_json(), eng.save()) == two
    assert two[0] == oracle_lib.OracleDoc.load_document(doc).patch_json()
    bad = bytearray(doc)
    bad[len(bad) - 100] ^= 4           # inside the last compressed column, checksum NOT repaired
    for call in (eng.load_document, eng.backend_load):
        with pytest.raises(engine.InvalidChanges) as e:
            call(bytes(bad))
        assert e.value.flags == (1 << 1)
        with pytest.raises(engine.EngineError):
            eng.patch_json()            # nothing is left to read after the failed load
    bad[4:8] = hashlib.sha256(bytes(bad[8:])).digest()[:4]
    for call in (eng.load_document, eng.backend_load):
        try:
            call(bytes(bad))
            if call == eng.load_document:
                eng.replay()
            got = eng.patch_json()
        except engine.InvalidChanges as e:
            assert e.flags != (1 << 1)
            got = None
        try:
            want = oracle_lib.OracleDoc.load_document(bytes(bad)).patch_json()
        except oracle_lib.OracleError:
            want = None
        assert got == want or got is None
    # the context is usable afterwards
    eng.backend_load(doc)
    assert eng.patch_json() == two[0]
    eng.close()
