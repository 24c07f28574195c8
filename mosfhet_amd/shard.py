"""Batch sharding across the GPUs of one node.

Every bootstrap / key switch is an independent unit, so a batch is cut into contiguous slices, one per rank
(one process per GPU); the bootstrap and key-switch keys are replicated per GPU at set-up and the data path has
NO collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the barrier around a
timed region, the max-over-ranks of its duration, and -- optionally -- gathering results for verification.
"""
import time


def shard_bounds(count, rank, world):
    """Contiguous slice [lo, hi) of `count` units owned by `rank`: the first count % world ranks get one extra."""
    assert 0 <= rank < world
    base, extra = divmod(count, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(array, rank, world):
    lo, hi = shard_bounds(len(array), rank, world)
    return array[lo:hi]


def dist_info():
    """(rank, world, dist module or None) from torch.distributed when initialised, else a single-process view."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), dist
    except ImportError:
        pass
    return 0, 1, None


def timed_region(step, steps, sync=None, device=None):
    """barrier + sync, `steps` calls of step(), sync + barrier; returns the MAX over ranks of the elapsed seconds."""
    import torch
    rank, world, dist = dist_info()
    sync = sync or (lambda: None)
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def gather_rows(local_rows, count):
    """All-gather variable-length shards of a [count, width] int64 result (verification only, not the hot path)."""
    import torch
    rank, world, dist = dist_info()
    if not dist:
        return local_rows
    width = local_rows.shape[1]
    longest = -(-count // world)
    pad = torch.zeros(longest, width, dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r in range(world):
        lo, hi = shard_bounds(count, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out)
