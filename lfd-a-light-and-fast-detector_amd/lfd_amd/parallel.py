"""Image-parallel multi-GPU helpers (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The LFD hot path shards by independent images: inference has NO data-path
collective -- every rank runs forward + decode + NMS on its own images with replicated
weights (<= 3.7 MB fp16); only the tiny result lists are gathered.  Training adds one gradient
all-reduce per step and the all-reduce of the loss normalisers: the reference's
nn.DataParallel computes the loss once over the gathered outputs (executor.py:39,198-200), so
`n_pos + 1` / `n_pos` (lfd.py:340,383) are GLOBAL-batch quantities.
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def shard_range(num_items, rank, world):
    """Contiguous, balanced partition of `num_items` images: first (num_items % world) ranks get one more."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local_results, num_items):
    """local per-image result lists -> full list in image order on every rank."""
    if not is_dist():
        return list(local_results)
    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, local_results)
    out = []
    for p in parts:
        out.extend(p)
    assert len(out) == num_items
    return out


def sharded_map(num_items, fn):
    """Image-parallel map: this rank evaluates `fn(lo, hi)` -> list of per-item results on ITS contiguous shard of
    `num_items` items; every rank gets the full list in item order.  No data-path collective -- only the results travel
    (tools/infer_sharded.py: fn = forward + decode + NMS on the rank's frames; replaces nn.DataParallel's scatter / gather,
    lfd/execution/executor.py:39,230-236)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)
    lo, hi = shard_range(num_items, rank, world)
    local = list(fn(lo, hi))
    assert len(local) == hi - lo, 'sharded_map: fn(lo, hi) must return one result per item'
    return gather_results(local, num_items)


def global_count(t):
    """Sum a per-rank count tensor over all ranks (loss normalisers)."""
    if is_dist():
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_mean_(tensors):
    """In-place average of a list of tensors over ranks, as ONE flat bucket (the whole WF-S
    gradient is 6.26 MB fp32: one collective per step, well below the xGMI per-link time of
    the compute step)."""
    if not is_dist() or not tensors:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].view_as(t))
        o += n


def max_over_ranks(value, device=None):
    """Wall-clock aggregation for benchmarks: MAX over ranks of a python float."""
    if not is_dist():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
