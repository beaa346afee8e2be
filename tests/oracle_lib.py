"""ctypes binding for the CPU oracle (oracle/libam_oracle.so) -- test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libam_oracle.so")


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("am_oracle.c", "am_oracle_apply.c", "am_oracle.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(src) for src in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        L.amo_replay.restype = ctypes.c_void_p
        L.amo_replay.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
        L.amo_load_document.restype = ctypes.c_void_p
        L.amo_load_document.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        L.amo_patch_json.restype = ctypes.c_void_p
        L.amo_patch_json.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.c_size_t]
        L.amo_free.argtypes = [ctypes.c_void_p]
        L.amo_set_document_history.restype = None
        L.amo_set_document_history.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
        L.amo_init.restype = ctypes.c_void_p
        L.amo_init.argtypes = []
        L.amo_apply_changes.restype = ctypes.c_void_p
        L.amo_apply_changes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.c_size_t]
        L.amo_sha256.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        for name, rt in [("amo_num_changes", ctypes.c_uint32), ("amo_num_applied", ctypes.c_uint32),
                         ("amo_num_ops", ctypes.c_uint64), ("amo_max_op", ctypes.c_uint64),
                         ("amo_num_actors", ctypes.c_uint32), ("amo_num_rows", ctypes.c_uint64)]:
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [ctypes.c_void_p]
        L.amo_change_hashes.restype = ctypes.c_void_p
        L.amo_change_hashes.argtypes = [ctypes.c_void_p]
        L.amo_actor.restype = ctypes.c_void_p
        L.amo_actor.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        L.amo_rows.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 7
        _lib = L
    return _lib


class OracleError(Exception):
    pass


def sha256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    buf = ctypes.create_string_buffer(data, len(data))
    lib().amo_sha256(buf, len(data), out)
    return out.raw


class OracleDoc:
    """Backend.loadChanges(Backend.init(), changes) as restated by the oracle."""

    def __init__(self, log):
        L = lib()
        self._arena = np.ascontiguousarray(log.arena, dtype=np.uint8)
        self._offsets = np.ascontiguousarray(log.offsets, dtype=np.uint64)
        err = ctypes.create_string_buffer(512)
        self._h = L.amo_replay(self._arena.ctypes.data, self._offsets.ctypes.data, len(self._offsets) - 1, err, 512)
        if not self._h:
            raise OracleError(err.value.decode())

    @classmethod
    def load_document(cls, doc_bytes: bytes):
        """Backend.load(bytes) as restated by the oracle."""
        self = cls.__new__(cls)
        self._doc = np.frombuffer(doc_bytes, dtype=np.uint8).copy()
        err = ctypes.create_string_buffer(512)
        self._h = lib().amo_load_document(self._doc.ctypes.data, self._doc.size, err, 512)
        if not self._h:
            raise OracleError(err.value.decode())
        return self

    def patch_json(self) -> str:
        L = lib()
        n = ctypes.c_size_t()
        err = ctypes.create_string_buffer(512)
        p = L.amo_patch_json(self._h, ctypes.byref(n), err, 512)
        if not p:
            raise OracleError(err.value.decode())
        return ctypes.string_at(p, n.value).decode("utf-8")

    @property
    def n_ops(self):
        return lib().amo_num_ops(self._h)

    @property
    def n_applied(self):
        return lib().amo_num_applied(self._h)

    @property
    def max_op(self):
        return lib().amo_max_op(self._h)

    def hashes(self):
        n = lib().amo_num_changes(self._h)
        p = lib().amo_change_hashes(self._h)
        return np.frombuffer(ctypes.string_at(p, 32 * n), dtype=np.uint8).reshape(n, 32).copy()

    def actors(self):
        out = []
        for i in range(lib().amo_num_actors(self._h)):
            ln = ctypes.c_uint32()
            p = lib().amo_actor(self._h, i, ctypes.byref(ln))
            out.append(ctypes.string_at(p, ln.value))
        return out

    def rows(self):
        n = lib().amo_num_rows(self._h)
        a = dict(id_ctr=np.zeros(n, np.uint64), id_actor=np.zeros(n, np.uint32), obj_ctr=np.zeros(n, np.uint64),
                 obj_actor=np.zeros(n, np.uint32), insert=np.zeros(n, np.uint8), action=np.zeros(n, np.uint32),
                 succ_num=np.zeros(n, np.uint32))
        lib().amo_rows(self._h, *[a[k].ctypes.data for k in
                                  ("id_ctr", "id_actor", "obj_ctr", "obj_actor", "insert", "action", "succ_num")])
        return a

    def close(self):
        if self._h:
            lib().amo_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleSession:
    """A BackendDoc advanced call by call: Backend.init() (or Backend.load(doc)) followed by Backend.applyChanges calls, each
    returning the incremental patch of that call as the oracle restates it (am_oracle_apply.c)."""

    def __init__(self, doc_bytes: bytes = None, doc_hashes: bytes = None, graph_rebuilt: bool = False):
        """doc_hashes: the hashes of the document's changes as the reference rebuilds them (32 bytes each, document order): lets the
        session follow the reference when it rebuilds its hash graph (am_oracle.h amo_set_document_history); graph_rebuilt: the
        reference had been asked for the document's changes before the first call."""
        L = lib()
        self._keep = []
        if doc_bytes is None:
            self._h = L.amo_init()
        else:
            buf = np.frombuffer(doc_bytes, dtype=np.uint8).copy()
            err = ctypes.create_string_buffer(512)
            self._h = L.amo_load_document(buf.ctypes.data, buf.size, err, 512)
            if not self._h:
                raise OracleError(err.value.decode())
            if doc_hashes is not None:
                hb = np.frombuffer(doc_hashes, dtype=np.uint8).copy()
                self._keep.append(hb)
                L.amo_set_document_history(self._h, hb.ctypes.data if hb.size else None, hb.size // 32, 1 if graph_rebuilt else 0)

    def apply(self, changes, local=False) -> str:
        """changes: list of bytes. Returns JSON.stringify(patch); raises OracleError (the session is then dead)."""
        if self._h is None:
            raise OracleError("session is dead")
        arena = np.frombuffer(b"".join(changes), dtype=np.uint8).copy() if changes else np.zeros(1, np.uint8)
        offsets = np.zeros(len(changes) + 1, np.uint64)
        np.cumsum([len(c) for c in changes], out=offsets[1:])
        n = ctypes.c_size_t()
        err = ctypes.create_string_buffer(512)
        p = lib().amo_apply_changes(self._h, arena.ctypes.data, offsets.ctypes.data, len(changes), 1 if local else 0, ctypes.byref(n), err, 512)
        if not p:
            lib().amo_free(self._h)
            self._h = None
            raise OracleError(err.value.decode())
        return ctypes.string_at(p, n.value).decode("utf-8")

    def patch_json(self) -> str:
        """Backend.getPatch of the current state."""
        n = ctypes.c_size_t()
        err = ctypes.create_string_buffer(512)
        p = lib().amo_patch_json(self._h, ctypes.byref(n), err, 512)
        if not p:
            raise OracleError(err.value.decode())
        return ctypes.string_at(p, n.value).decode("utf-8")

    def close(self):
        if self._h:
            lib().amo_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
