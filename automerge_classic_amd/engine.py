"""ctypes binding of the C ABI in include/am355.h (the HIP replay engine, csrc/libam355.so).

There is no CPU implementation behind this module: if the HIP library is missing or no MI355X is visible,
constructing an Engine raises.  (tests/emu builds a CPU *emulation* of the kernels for logic tests in the
GPU-less container; it is only ever loaded when a test passes its path explicitly.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libam355.so")

AM355_OK = 0
AM355_E_DEVICE, AM355_E_ARG, AM355_E_INVALID, AM355_E_UNSUPPORTED, AM355_E_STATE, AM355_E_NOMEM = -1, -2, -3, -4, -5, -6

FLAG_NAMES = {
    1 << 0: "BAD_MAGIC", 1 << 1: "BAD_CHECKSUM", 1 << 2: "BAD_CHUNK", 1 << 3: "BAD_COLUMNS", 1 << 4: "BAD_LEB", 1 << 5: "BAD_RLE",
    1 << 6: "BAD_ROW", 1 << 7: "UNKNOWN_OBJECT", 1 << 8: "BAD_ELEM", 1 << 9: "BAD_PRED", 1 << 10: "DUP_OPID", 1 << 11: "BAD_COUNTER",
    1 << 12: "UNSUPPORTED", 1 << 13: "OVERFLOW", 1 << 16: "BAD_SEQ", 1 << 17: "UNKNOWN_ACTOR", 1 << 18: "BAD_DEFLATE",
}


class Stats(ctypes.Structure):
    _fields_ = [
        ("n_changes", ctypes.c_uint32), ("n_applied", ctypes.c_uint32), ("n_pending", ctypes.c_uint32),
        ("n_actors", ctypes.c_uint32), ("n_objects", ctypes.c_uint32), ("n_heads", ctypes.c_uint32),
        ("n_ops", ctypes.c_uint64), ("max_op", ctypes.c_uint64), ("raw_bytes", ctypes.c_uint64),
        ("n_map_values", ctypes.c_uint64), ("n_list_elems", ctypes.c_uint64), ("n_edits", ctypes.c_uint64),
        ("ir_bytes", ctypes.c_uint64),
        ("ms_total", ctypes.c_float), ("ms_parse", ctypes.c_float), ("ms_host_schedule", ctypes.c_float),
        ("ms_decode", ctypes.c_float), ("ms_merge", ctypes.c_float), ("ms_order", ctypes.c_float), ("ms_hash_stream", ctypes.c_float),
        ("fast_path", ctypes.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class EngineError(RuntimeError):
    def __init__(self, code, message, flags=0):
        names = [n for b, n in FLAG_NAMES.items() if flags & b]
        super().__init__(f"am355 error {code}: {message}" + (f" [{'|'.join(names)}]" if names else ""))
        self.code = code
        self.flags = flags
        self.flag_names = names


class InvalidChanges(EngineError):
    """The reference would throw on this input (the JS host re-runs it on the JS path to raise the exact error)."""


class UnsupportedChanges(EngineError):
    """Legal input outside the GPU-served subset (documented in DESIGN.md); the JS host uses the JS path."""


def _bind(path):
    if not os.path.exists(path):
        raise RuntimeError(f"HIP engine library not found: {path}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(path)
    vp, u32, u64p = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p
    L.am355_create.restype = vp
    L.am355_create.argtypes = [ctypes.c_int]
    L.am355_destroy.argtypes = [vp]
    L.am355_last_error.restype = ctypes.c_char_p
    L.am355_last_error.argtypes = [vp]
    L.am355_flags.restype = u32
    L.am355_flags.argtypes = [vp]
    L.am355_load_changes.argtypes = [vp, vp, u64p, u32]
    L.am355_load_document.argtypes = [vp, vp, ctypes.c_size_t]
    if hasattr(L, "am355_backend_load"):   # (tools/ab_libs.sh loads libraries of earlier rounds, which end before this entry point; tests/test_abi.py holds the shipped library to the header)
        L.am355_backend_load.argtypes = [vp, vp, ctypes.c_size_t]
        L.am355_backend_load.restype = ctypes.c_int
    L.am355_replay.argtypes = [vp]
    L.am355_patch_json.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t)]
    L.am355_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.am355_get_hashes.argtypes = [vp, vp]
    L.am355_test_sort.argtypes = [vp, vp, vp, u32, ctypes.c_int]
    L.am355_test_scan.argtypes = [vp, vp, vp, u32, vp]
    L.am355_get_rows.argtypes = [vp] * 15
    L.am355_save.argtypes = [vp, u32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.am355_get_applied.argtypes = [vp, vp, ctypes.POINTER(u32)]
    L.am355_fetch_ir.argtypes = [vp, vp]
    L.am355_set_shard.argtypes = [vp, u32, u32]
    L.am355_fragment_size.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
    L.am355_export_fragment.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
    L.am355_import_fragments.argtypes = [vp, vp, vp, u32]
    L.am355_shard_unique_id.argtypes = [vp]
    L.am355_shard_init.argtypes = [vp, vp, u32, u32]
    L.am355_sharded_replay.argtypes = [vp, ctypes.c_int]
    L.am355_shard_fragment_bytes.argtypes = [vp, vp, u32]
    L.am355_shard_finalize.argtypes = [vp]
    for f in ("am355_shard_unique_id", "am355_shard_init", "am355_sharded_replay", "am355_shard_fragment_bytes", "am355_shard_finalize"):
        getattr(L, f).restype = ctypes.c_int
    if hasattr(L, "am355_resident_counters"):   # (tools/ab_apply.sh loads libraries of earlier commits; tests/test_abi.py holds the shipped one to the header)
        L.am355_resident_counters.argtypes = [vp, vp]
        L.am355_resident_counters.restype = ctypes.c_int
    if hasattr(L, "am355_resident_maps_only_calls"):
        L.am355_resident_maps_only_calls.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
        L.am355_resident_maps_only_calls.restype = ctypes.c_int
    L.am355_get_raw.argtypes = [vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(u32)]
    L.am355_doc_changes.argtypes = [vp, u32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_void_p)]
    L.am355_apply_changes.argtypes = [vp, vp, u64p, u32]
    L.am355_apply_patch_json.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t)]
    L.am355_fetch_apply_ir.argtypes = [vp, vp]
    L.am355_reset.argtypes = [vp]
    L.am355_get_dep_graph.argtypes = [vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(u32)]
    L.am355_sync_bloom_build.argtypes = [vp, vp, u32, vp, ctypes.c_size_t]
    L.am355_sync_bloom_probe.argtypes = [vp, vp, u32, u32, u32, u32, vp, ctypes.c_size_t, vp]
    L.am355_get_pending.argtypes = [vp, vp, ctypes.POINTER(u32)]
    L.am355_forget_call_history.argtypes = [vp, ctypes.c_int]
    L.am355_hash_graph_known.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.am355_set_phase_events.argtypes = [vp, ctypes.c_int]
    for f in ("am355_load_changes", "am355_load_document", "am355_replay", "am355_patch_json", "am355_get_stats", "am355_get_hashes", "am355_test_sort",
              "am355_test_scan", "am355_get_rows", "am355_save", "am355_get_applied", "am355_fetch_ir", "am355_get_raw", "am355_set_shard", "am355_fragment_size", "am355_export_fragment",
              "am355_import_fragments", "am355_doc_changes", "am355_apply_changes", "am355_apply_patch_json", "am355_fetch_apply_ir", "am355_reset", "am355_get_pending", "am355_forget_call_history", "am355_hash_graph_known", "am355_set_phase_events", "am355_get_dep_graph", "am355_sync_bloom_build", "am355_sync_bloom_probe"):
        getattr(L, f).restype = ctypes.c_int
    return L


_libs = {}


def load_library(path=DEFAULT_LIB):
    if path not in _libs:
        _libs[path] = _bind(path)
    return _libs[path]


class Engine:
    """One replay context bound to one GPU (am355_ctx)."""

    def __init__(self, device=0, lib_path=DEFAULT_LIB):
        self._L = load_library(lib_path)
        self.device = device
        self.lib_path = lib_path
        self._h = self._L.am355_create(device)
        if not self._h:
            raise RuntimeError("am355_create failed: no usable MI355X / HIP device (the engine has no CPU fallback)")
        self._n_changes = 0

    def set_phase_events(self, on):
        """Measurement switch (am355_set_phase_events): HIP events between the phases of the following replays."""
        self._check(self._L.am355_set_phase_events(self._h, 1 if on else 0))

    def close(self):
        if getattr(self, "_h", None):
            self._L.am355_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == AM355_OK:
            return
        msg = self._L.am355_last_error(self._h).decode("utf-8", "replace")
        flags = self._L.am355_flags(self._h)
        cls = InvalidChanges if rc == AM355_E_INVALID else UnsupportedChanges if rc == AM355_E_UNSUPPORTED else EngineError
        raise cls(rc, msg, flags)

    # ---- the path -------------------------------------------------------------------------------------
    def load_changes(self, log):
        """Stage a batch (host inflate of DEFLATEd changes + copy to HBM). `log` has .arena (uint8) and .offsets (uint64)."""
        arena = np.ascontiguousarray(log.arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(log.offsets, dtype=np.uint64)
        self._n_changes = int(offsets.size - 1)
        self._check(self._L.am355_load_changes(self._h, arena.ctypes.data if arena.size else None, offsets.ctypes.data, self._n_changes))

    def load_document(self, doc: bytes):
        """Stage one saved document (Backend.save bytes): host header parse / checksum / inflate, op columns to HBM."""
        buf = np.frombuffer(bytes(doc), dtype=np.uint8)
        self._n_changes = 0
        self._check(self._L.am355_load_document(self._h, buf.ctypes.data, buf.size))

    def backend_load(self, doc: bytes):
        """Backend.load(bytes) in one call (am355_backend_load): load_document + replay with the chunk checksum on a thread beside the
        device stages and the patch IR on its way to the host when the call returns."""
        buf = doc if isinstance(doc, np.ndarray) else np.frombuffer(bytes(doc), dtype=np.uint8)
        self._n_changes = 0
        self._check(self._L.am355_backend_load(self._h, buf.ctypes.data, buf.size))

    def replay(self):
        """The hot path: decode + schedule + merge + patch IR, device-resident in and out."""
        self._check(self._L.am355_replay(self._h))

    def apply_changes(self, log):
        """Backend.applyChanges(state, changes): the state is what the context holds (earlier load_changes + replay or apply_changes
        calls, or nothing = Backend.init()). Replays everything, derives the incremental patch of the batch on the device."""
        arena = np.ascontiguousarray(log.arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(log.offsets, dtype=np.uint64)
        self._check(self._L.am355_apply_changes(self._h, arena.ctypes.data if arena.size else None, offsets.ctypes.data, int(offsets.size - 1)))
        self._n_changes = int(self.stats().n_changes)

    def reset(self):
        """Forget the state: the next apply_changes starts from Backend.init()."""
        self._check(self._L.am355_reset(self._h))
        self._n_changes = 0

    def forget_call_history(self, doc_changes=0):
        """The staged changes were replayed in one go, not by the Backend.applyChanges calls that built the state; doc_changes: how
        many of the leading ones are the rebuilt history of a loaded document (include/am355.h)."""
        self._check(self._L.am355_forget_call_history(self._h, int(doc_changes)))

    def hash_graph_known(self, set_to=None):
        """Lineage that began with a loaded document: has the reference rebuilt the document's hash graph (include/am355.h)?
        set_to True / False tells the engine; returns the state afterwards."""
        known = ctypes.c_int(0)
        self._check(self._L.am355_hash_graph_known(self._h, -1 if set_to is None else (1 if set_to else 0), ctypes.byref(known)))
        return bool(known.value)

    def pending(self):
        """Indexes (into the engine's list of changes) of the changes still queued for a missing dependency."""
        n = ctypes.c_uint32()
        self._check(self._L.am355_get_pending(self._h, None, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            self._check(self._L.am355_get_pending(self._h, out.ctypes.data, ctypes.byref(n)))
        return out

    # ---- sync protocol, bulk side ----------------------------------------------------------
    def dep_graph(self):
        """(dep_first[n + 1], dep_index[]) over the context's list of changes: change i depends on dep_index[dep_first[i]:dep_first[i + 1]]
        (0xffffffff: a change the context does not hold)."""
        f, x, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint32()
        self._check(self._L.am355_get_dep_graph(self._h, ctypes.byref(f), ctypes.byref(x), ctypes.byref(n)))
        first = np.ctypeslib.as_array(ctypes.cast(f, ctypes.POINTER(ctypes.c_uint32)), shape=(n.value + 1,)).copy()
        m = int(first[-1])
        index = np.ctypeslib.as_array(ctypes.cast(x, ctypes.POINTER(ctypes.c_uint32)), shape=(m,)).copy() if m else np.zeros(0, np.uint32)
        return first, index

    def bloom_build(self, idx):
        """`bits` of the sync protocol's Bloom filter (sync.js:38-128) over the hashes of the changes idx[], built on the device."""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        out = np.zeros((idx.size * 10 + 7) // 8, dtype=np.uint8)
        self._check(self._L.am355_sync_bloom_build(self._h, idx.ctypes.data if idx.size else None, idx.size, out.ctypes.data, out.size))
        return out

    def bloom_probe(self, idx, num_entries, bits_per_entry, num_probes, bits):
        """contains[k] for the hashes of the changes idx[] in a filter received from a peer."""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        out = np.zeros(max(idx.size, 1), dtype=np.uint8)
        self._check(self._L.am355_sync_bloom_probe(self._h, idx.ctypes.data if idx.size else None, idx.size, num_entries, bits_per_entry, num_probes,
                                                   bits.ctypes.data if bits.size else None, bits.size, out.ctypes.data))
        return out[:idx.size]

    def apply_patch_json(self):
        """JSON.stringify of the patch the last apply_changes returned (the reference's incremental patch)."""
        p = ctypes.c_char_p()
        n = ctypes.c_size_t()
        self._check(self._L.am355_apply_patch_json(self._h, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p, n.value).decode("utf-8")

    def fetch_ir(self):
        """Patch IR + envelope from HBM into host memory owned by the context (what the JS host materialises the patch from)."""
        self._check(self._L.am355_fetch_ir(self._h, None))

    def patch_json(self):
        p = ctypes.c_char_p()
        n = ctypes.c_size_t()
        self._check(self._L.am355_patch_json(self._h, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p, n.value).decode("utf-8")

    def save(self, reencode=False):
        """Backend.save(state): the document as one binary chunk (op columns encoded on the GPU). A loaded document returns
        the bytes it came from, as the reference does; reencode=True re-encodes its op columns instead (diagnostic)."""
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        self._check(self._L.am355_save(self._h, 1 if reencode else 0, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p, n.value)

    def applied(self):
        """Input indexes of the applied changes in application order (BackendDoc.changes / getAllChanges order)."""
        n = ctypes.c_uint32()
        self._check(self._L.am355_get_applied(self._h, None, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        self._check(self._L.am355_get_applied(self._h, out.ctypes.data if n.value else None, ctypes.byref(n)))
        return out

    def stats(self):
        s = Stats()
        self._check(self._L.am355_get_stats(self._h, ctypes.byref(s)))
        return s

    # ---- objectId sharding (one Engine per GPU / process) ---------------------------------------------------
    def set_shard(self, rank, world):
        self._check(self._L.am355_set_shard(self._h, rank, world))

    def fragment_size(self):
        n = ctypes.c_size_t()
        self._check(self._L.am355_fragment_size(self._h, ctypes.byref(n)))
        return n.value

    def export_fragment(self, ptr, capacity, device):
        """This rank's record tables into caller memory at `ptr` (device or host); returns the bytes written."""
        n = ctypes.c_size_t()
        self._check(self._L.am355_export_fragment(self._h, ctypes.c_void_p(ptr), capacity, 1 if device else 0, ctypes.byref(n)))
        return n.value

    def import_fragments(self, frags, offsets):
        """frags: uint8 host array holding the fragments of all ranks, offsets: uint64[world + 1]."""
        f = np.ascontiguousarray(frags, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self._L.am355_import_fragments(self._h, f.ctypes.data, o.ctypes.data, o.size - 1))

    # ---- the same with the collective inside the library: RCCL over xGMI (am355_shard_init / am355_sharded_replay) ----
    def shard_unique_id(self):
        """128 bytes from ncclGetUniqueId (rank 0 makes them, the host carries them to the other ranks' processes)."""
        buf = (ctypes.c_uint8 * 128)()
        if self._L.am355_shard_unique_id(buf) != AM355_OK:
            raise EngineError(AM355_E_DEVICE, "RCCL is not available (librccl.so.1, or AM355_RCCL_LIB)")
        return bytes(buf)

    def shard_init(self, unique_id, rank, world):
        assert len(unique_id) == 128
        self._check(self._L.am355_shard_init(self._h, ctypes.c_char_p(bytes(unique_id)), rank, world))

    def sharded_replay(self, stitch_on_all_ranks=False):
        """am355_replay of the staged batch + ncclAllGather of the fragments + stitch (rank 0, or every rank)."""
        self._check(self._L.am355_sharded_replay(self._h, 1 if stitch_on_all_ranks else 0))

    def shard_fragment_bytes(self, world):
        out = np.zeros(world, dtype=np.uint64)
        self._check(self._L.am355_shard_fragment_bytes(self._h, out.ctypes.data, world))
        return out

    def shard_finalize(self):
        self._check(self._L.am355_shard_finalize(self._h))

    def resident_counters(self):
        """(calls of apply_changes that merged the batch alone into the resident state, calls that asked for it and took the full replay,
        calls of the first kind that also merged their new list elements into the stored order in place)."""
        out = (ctypes.c_uint64 * 3)()
        self._check(self._L.am355_resident_counters(self._h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def resident_maps_only_calls(self):
        """Calls of apply_changes onto the resident state whose batch held plain map rows only (the map half of the merge ran alone)."""
        out = ctypes.c_uint64()
        self._check(self._L.am355_resident_maps_only_calls(self._h, ctypes.byref(out)))
        return int(out.value)

    def raw(self):
        """(arena, offsets) as staged: the uncompressed change containers back to back (copies)."""
        a, o, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint32()
        self._check(self._L.am355_get_raw(self._h, ctypes.byref(a), ctypes.byref(o), ctypes.byref(n)))
        offs = np.ctypeslib.as_array(ctypes.cast(o, ctypes.POINTER(ctypes.c_uint64)), shape=(n.value + 1,)).copy()
        size = int(offs[-1])
        arena = np.ctypeslib.as_array(ctypes.cast(a, ctypes.POINTER(ctypes.c_uint8)), shape=(size,)).copy() if size else np.zeros(0, dtype=np.uint8)
        return arena, offs

    def doc_changes(self, deflate=True):
        """History of a loaded document: (arena, offsets, hashes) = the binary changes Backend.getAllChanges(Backend.load(doc)) returns,
        back to back in document order, and their 32-byte hashes (reference new.js:1887-1912, columnar.js:876-981). Copies."""
        a, o, n, h = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint32(), ctypes.c_void_p()
        self._check(self._L.am355_doc_changes(self._h, 1 if deflate else 0, ctypes.byref(a), ctypes.byref(o), ctypes.byref(n), ctypes.byref(h)))
        offs = np.ctypeslib.as_array(ctypes.cast(o, ctypes.POINTER(ctypes.c_uint64)), shape=(n.value + 1,)).copy()
        size = int(offs[-1])
        arena = np.ctypeslib.as_array(ctypes.cast(a, ctypes.POINTER(ctypes.c_uint8)), shape=(size,)).copy() if size else np.zeros(0, dtype=np.uint8)
        hashes = np.ctypeslib.as_array(ctypes.cast(h, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value, 32)).copy() if n.value else np.zeros((0, 32), dtype=np.uint8)
        return arena, offs, hashes

    def hashes(self):
        out = np.zeros((self._n_changes, 32), dtype=np.uint8)
        self._check(self._L.am355_get_hashes(self._h, out.ctypes.data))
        return out

    # ---- diagnostics ------------------------------------------------------------------------------------
    def test_sort(self, keys, vals, key_bits=64):
        k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        self._check(self._L.am355_test_sort(self._h, k.ctypes.data, v.ctypes.data, k.size, key_bits))
        return k, v

    def test_scan(self, values):
        a = np.ascontiguousarray(values, dtype=np.uint32)
        out = np.zeros_like(a)
        total = np.zeros(1, dtype=np.uint32)
        self._check(self._L.am355_test_scan(self._h, a.ctypes.data, out.ctypes.data, a.size, total.ctypes.data))
        return out, int(total[0])

    def rows(self):
        n = int(self.stats().n_ops)
        names = ["obj_actor", "obj_ctr", "key_actor", "key_ctr", "key_off", "key_len", "action", "val_tl", "val_off", "pred_num",
                 "id_ctr", "id_actor", "insert", "succ_cnt"]
        arrs = {k: np.zeros(n, dtype=np.uint8 if k == "insert" else np.uint32) for k in names}
        self._check(self._L.am355_get_rows(self._h, *[arrs[k].ctypes.data for k in names]))
        return arrs


def replay_patch_json(log, device=0, lib_path=DEFAULT_LIB):
    """Backend.getPatch(Backend.loadChanges(Backend.init(), changes)) as JSON text, computed on the GPU."""
    eng = Engine(device, lib_path)
    try:
        eng.load_changes(log)
        eng.replay()
        return eng.patch_json()
    finally:
        eng.close()
