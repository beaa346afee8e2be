"""objectId sharding of ONE document over the GPUs of a node (include/am355.h "objectId sharding").

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests). Every rank
stages the whole batch of changes (rows keep their global indexes; a change none of whose rows it owns is decoded only as far as the
object columns) or the whole saved document (its columns are run-length streams: decoded everywhere), merges only the objects it owns
(am355_set_shard) and writes its part of the patch IR -- the map records / edit records / values of its objects -- as one
contiguous fragment. The fragments are the only data exchanged: an all_gather of uint8 tensors (device tensors with RCCL: the
fragment goes from HBM to HBM over xGMI), then every rank (or rank 0 only) stitches them by object index
(am355_import_fragments) into the patch of the whole document.

Reference semantics that make this a partition: ordering and pred / succ resolution never cross objects
(backend/new.js:1141-1145, 1173-1176); the one cross-object link is make op -> child object (new.js:894-897, 973-976), carried
as the object index, which is the same on every rank because make rows are decoded everywhere.
"""
import numpy as np


class ShardedReplay:
    """The sharded hot path of one rank. `dist`: torch.distributed (initialised), `device`: torch device of the collective."""

    def __init__(self, eng, dist, device, stitch_on_all_ranks=False):
        import torch
        self.torch, self.eng, self.dist, self.device = torch, eng, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.stitch_all = stitch_on_all_ranks
        eng.set_shard(self.rank, self.world)
        self._send = None
        self._recv = None
        self.last = {}

    def _buffers(self, cap):
        torch = self.torch
        if self._send is None or self._send.numel() < cap:
            cap = int(cap * 1.25) + 4096
            self._send = torch.empty(cap, dtype=torch.uint8, device=self.device)
            self._recv = torch.empty(cap * self.world, dtype=torch.uint8, device=self.device)
        return self._send, self._recv

    def step(self, stage):
        """stage(): stages the batch on this rank's engine (am355_load_changes). Returns True on the ranks that hold the stitched
        patch afterwards (rank 0, or all). Raises on every rank if any rank rejected the batch."""
        torch, dist, eng = self.torch, self.dist, self.eng
        err = None
        try:
            stage()
            eng.replay()
            need = eng.fragment_size()
        except Exception as e:  # (kept until every rank knows: a collective must not be entered by some ranks only)
            err, need = e, 0
        # one small all_gather: [failed, fragment bytes] of every rank
        mine = torch.tensor([1 if err else 0, need], dtype=torch.int64, device=self.device)
        allv = torch.empty(2 * self.world, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(allv, mine)
        allv = allv.cpu().numpy().reshape(self.world, 2)
        if allv[:, 0].any():
            raise err if err else RuntimeError(f"rank(s) {np.nonzero(allv[:, 0])[0].tolist()} rejected the batch")
        sizes = allv[:, 1].astype(np.uint64)
        cap = int(sizes.max())
        send, recv = self._buffers(cap)
        wrote = eng.export_fragment(send.data_ptr(), send.numel(), self.device.type != "cpu")
        assert wrote == need
        stride = send.numel()
        dist.all_gather_into_tensor(recv[: stride * self.world], send)  # the data-path collective: fragments, HBM -> HBM over xGMI
        self.last = {"fragment_bytes": sizes.tolist()}
        if self.rank != 0 and not self.stitch_all:
            return False
        host = recv[: stride * self.world].cpu().numpy()
        offsets = np.arange(self.world + 1, dtype=np.uint64) * np.uint64(stride)
        # (fragment r occupies host[offsets[r] : offsets[r] + sizes[r]]; the stride padding after it is ignored)
        eng.import_fragments(host, offsets)
        return True


class NativeShardedReplay:
    """The same hot path with the collective INSIDE the library (am355_shard_init / am355_sharded_replay: ncclAllGather on the
    context's stream, no torch tensor on the data path). `dist` only carries the 128-byte RCCL unique id from rank 0 to the other
    ranks once -- what the JS host does with process.send between its per-GPU workers (js/sharded.js)."""

    def __init__(self, eng, dist, stitch_on_all_ranks=False):
        self.eng, self.rank, self.world = eng, dist.get_rank(), dist.get_world_size()
        self.stitch_all = stitch_on_all_ranks
        box = [eng.shard_unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        eng.shard_init(box[0], self.rank, self.world)
        self.last = {}

    def step(self, stage):
        """stage(): stages the batch on this rank's engine. True on the ranks that hold the stitched patch afterwards. A staging
        failure on this rank is carried into the library's first collective (am355_sharded_replay is entered on every rank)."""
        err = None
        try:
            stage()
        except Exception as e:
            err = e
            self.eng.reset()   # (nothing staged: am355_sharded_replay reports this rank as failed to the others)
        try:
            self.eng.sharded_replay(self.stitch_all)
        except Exception as e:
            raise err if err else e
        self.last = {"fragment_bytes": [int(x) for x in self.eng.shard_fragment_bytes(self.world)]}
        return self.rank == 0 or self.stitch_all

    def close(self):
        self.eng.shard_finalize()


def bench_sharded(eng, stage, dist, device, steps, warmup, barrier):
    """K timed sharded replays of what `stage()` stages on this rank's engine -- am355_load_changes of a change log, or
    am355_load_document of a saved document -- (host buffers in -> stitched patch IR on rank 0's host). Returns seconds (this rank)."""
    import time
    sr = ShardedReplay(eng, dist, device)
    for _ in range(warmup):
        sr.step(stage)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sr.step(stage)
    barrier()
    dt = time.perf_counter() - t0
    info = dict(sr.last)
    eng.set_shard(0, 1)
    return dt, info
