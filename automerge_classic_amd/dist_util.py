"""Multi-GPU helpers for bench.py: one process per GPU, replicas only.

The headline document is a single Text object, which objectId sharding cannot split, so under torch.distributed
every rank replays its own document of the same shape; there is no data-path collective. The only communication
is the timing contract of bench.py: barrier, MAX of the elapsed times, SUM of the ops.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def rank_seed(base_seed, rank):
    """Every replica gets its own synthetic document (same shape, different seed)."""
    return base_seed + 1000 * rank


def aggregate(elapsed_s, ops, dist=None, device=None):
    """(max elapsed over ranks, total ops over ranks). `dist` is torch.distributed or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(ops)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    o = torch.tensor([float(ops)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(o, op=dist.ReduceOp.SUM)
    return float(t[0]), float(o[0])
