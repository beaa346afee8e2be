"""ctypes binding for the synthetic change-log generator (loggen/libamlog.so).

The generator writes binary Automerge changes exactly as the reference encoder would
(reference: backend/columnar.js:710-739); workloads follow SURVEY.md §8(d).
"""
import ctypes
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "loggen", "libamlog.so")

KIND_TEXT_TYPING = 0
KIND_MAP_LWW = 1
KIND_TEXT_CONCURRENT = 2


class _Params(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_uint32), ("n_actors", ctypes.c_uint32), ("n_rounds", ctypes.c_uint32),
        ("ins_per_change", ctypes.c_uint32), ("del_per_change", ctypes.c_uint32), ("n_objects", ctypes.c_uint32),
        ("n_keys", ctypes.c_uint32), ("ops_per_change", ctypes.c_uint32), ("n_ops", ctypes.c_uint64),
        ("seed", ctypes.c_uint64), ("deflate", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
    ]


class _Log(ctypes.Structure):
    _fields_ = [
        ("arena", ctypes.POINTER(ctypes.c_uint8)), ("offsets", ctypes.POINTER(ctypes.c_uint64)),
        ("n_changes", ctypes.c_uint32), ("n_actors", ctypes.c_uint32), ("n_ops", ctypes.c_uint64),
        ("raw_bytes", ctypes.c_uint64),
    ]


class _DocParams(ctypes.Structure):
    _fields_ = [
        ("n_actors", ctypes.c_uint32), ("n_texts", ctypes.c_uint32), ("text_len", ctypes.c_uint32), ("n_maps", ctypes.c_uint32),
        ("keys_per_map", ctypes.c_uint32), ("n_submaps", ctypes.c_uint32), ("n_lists", ctypes.c_uint32), ("list_len", ctypes.c_uint32),
        ("deflate", ctypes.c_uint32), ("pad", ctypes.c_uint32), ("seed", ctypes.c_uint64),
    ]


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.amlog_generate.argtypes = [ctypes.POINTER(_Params), ctypes.POINTER(_Log)]
        _lib.amlog_generate.restype = ctypes.c_int
        _lib.amlog_free.argtypes = [ctypes.POINTER(_Log)]
        _lib.amlog_generate_document.argtypes = [ctypes.POINTER(_DocParams), ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                                 ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        _lib.amlog_generate_document.restype = ctypes.c_int
        _lib.amlog_free_bytes.argtypes = [ctypes.POINTER(ctypes.c_uint8)]
    return _lib


class ChangeLog:
    """A batch of binary changes: `arena` (uint8) holds them back to back, `offsets` (uint64, n+1) delimits them."""

    def __init__(self, arena, offsets, n_ops, raw_bytes=None, n_actors=None, name=""):
        self.arena = np.ascontiguousarray(arena, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n_ops = int(n_ops)
        self.raw_bytes = int(raw_bytes) if raw_bytes is not None else int(self.arena.size)
        self.n_actors = n_actors
        self.name = name

    @property
    def n_changes(self):
        return int(self.offsets.size - 1)

    def change(self, i):
        return self.arena[int(self.offsets[i]):int(self.offsets[i + 1])].tobytes()

    def changes(self):
        return [self.change(i) for i in range(self.n_changes)]

    @staticmethod
    def from_changes(changes, n_ops=0, name=""):
        offs = np.zeros(len(changes) + 1, dtype=np.uint64)
        for i, c in enumerate(changes):
            offs[i + 1] = offs[i] + len(c)
        arena = np.frombuffer(b"".join(changes), dtype=np.uint8) if changes else np.zeros(0, dtype=np.uint8)
        return ChangeLog(arena, offs, n_ops, name=name)

    def reordered(self, perm):
        return ChangeLog.from_changes([self.change(int(i)) for i in perm], self.n_ops, name=self.name + "+perm")

    def save(self, path):
        """File layout: u32 n_changes, u64 n_ops, u64 offsets[n+1], arena bytes (little endian)."""
        with open(path, "wb") as f:
            f.write(struct.pack("<IQ", self.n_changes, self.n_ops))
            f.write(self.offsets.tobytes())
            f.write(self.arena.tobytes())

    @staticmethod
    def load(path, name=""):
        with open(path, "rb") as f:
            n, n_ops = struct.unpack("<IQ", f.read(12))
            offs = np.frombuffer(f.read(8 * (n + 1)), dtype=np.uint64)
            arena = np.frombuffer(f.read(), dtype=np.uint8)
        return ChangeLog(arena, offs, n_ops, name=name or os.path.basename(path))


def generate(kind, *, n_actors=0, n_rounds=0, ins_per_change=0, del_per_change=0, n_objects=0, n_keys=0,
             ops_per_change=0, n_ops=0, seed=0x5EED0000, deflate=False, name=""):
    lib = _load()
    p = _Params(kind, n_actors, n_rounds, ins_per_change, del_per_change, n_objects, n_keys, ops_per_change,
                n_ops, seed, 1 if deflate else 0, 0)
    log = _Log()
    rc = lib.amlog_generate(ctypes.byref(p), ctypes.byref(log))
    if rc != 0:
        raise RuntimeError(f"amlog_generate failed rc={rc}")
    try:
        n = log.n_changes
        offsets = np.ctypeslib.as_array(log.offsets, shape=(n + 1,)).copy()
        total = int(offsets[-1])
        arena = np.ctypeslib.as_array(log.arena, shape=(max(total, 1),))[:total].copy()
        return ChangeLog(arena, offsets, log.n_ops, log.raw_bytes, log.n_actors, name=name)
    finally:
        lib.amlog_free(ctypes.byref(log))


# The benchmark configurations of BASELINE.json / SURVEY.md §8(d).
def config(name, scale=1.0, deflate=False):
    if name == "c2_text_typing":
        return generate(KIND_TEXT_TYPING, n_ops=int(100_000 * scale), ops_per_change=100, seed=0x5EED0002,
                        deflate=deflate, name=name)
    if name == "c3_map_lww":
        return generate(KIND_MAP_LWW, n_actors=32, n_rounds=max(1, int(8 * scale)), n_keys=10_000, seed=0x5EED0003,
                        deflate=deflate, name=name)
    if name == "c4_text_single":
        return generate(KIND_TEXT_CONCURRENT, n_actors=64, n_rounds=max(1, int(64 * scale)), ins_per_change=200,
                        del_per_change=50, n_objects=1, seed=0x5EED0004, deflate=deflate, name=name)
    if name == "c4_text_multi":
        return generate(KIND_TEXT_CONCURRENT, n_actors=64, n_rounds=max(1, int(64 * scale)), ins_per_change=200,
                        del_per_change=50, n_objects=64, seed=0x5EED0004, deflate=deflate, name=name)
    raise KeyError(name)


def generate_document(*, n_actors=16, n_texts=4, text_len=1000, n_maps=4, keys_per_map=200, n_submaps=2, n_lists=2, list_len=300,
                      deflate=True, seed=0x5EED0005):
    """A synthetic saved document (Backend.save format) with a mix of Text, nested maps/tables (conflicts, counters) and
    lists of primitives (BASELINE config 5). Returns (document bytes, number of op rows)."""
    lib = _load()
    p = _DocParams(n_actors, n_texts, text_len, n_maps, keys_per_map, n_submaps, n_lists, list_len, 1 if deflate else 0, 0, seed)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_uint64()
    rows = ctypes.c_uint64()
    rc = lib.amlog_generate_document(ctypes.byref(p), ctypes.byref(out), ctypes.byref(n), ctypes.byref(rows))
    if rc != 0:
        raise RuntimeError(f"amlog_generate_document failed rc={rc}")
    try:
        return ctypes.string_at(out, n.value), int(rows.value)
    finally:
        lib.amlog_free_bytes(out)


def document_config(scale=1.0, deflate=True):
    """BASELINE config 5 shape: ~40 % text, ~40 % map, ~20 % list rows; scale 1.0 is about 10 M rows."""
    s = max(scale, 1e-4)
    return generate_document(n_actors=64, n_texts=256, text_len=max(1, int(13000 * s)), n_maps=100, keys_per_map=max(4, int(22000 * s)),
                             n_submaps=4, n_lists=64, list_len=max(1, int(26000 * s)), deflate=deflate)
