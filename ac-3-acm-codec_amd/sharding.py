"""Multi-GPU layout of the engine: independent streams are split contiguously over ranks
(one process per GPU), with NO data-path collective.  The only communication is
the benchmark's own barrier and a MAX-reduce of the elapsed time."""
import os


def shard(n_streams, world, rank):
    """Contiguous [lo, hi) range of stream ids owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_streams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def reduce_max(values, dist=None, device=None):
    """MAX over ranks of a list of floats (identity without a process group)."""
    if dist is None:
        return list(values)
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def reduce_sum(values, dist=None, device=None):
    if dist is None:
        return list(values)
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t]
