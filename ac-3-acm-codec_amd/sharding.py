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


HBM_BYTES = 288 * 10 ** 9          # MI355X: 288 GB of HBM3E per GPU


def plan_transcode_bytes(streams, frames_per_stream=1, tile_frames=131072, frame_bytes=1536, nch=6):
    """HBM one rank needs for its shard of a decode -> s16 -> re-encode job (ac3mi_transcode_batch), in bytes: what grows
    with the shard (frames in and out, per-stream carry-over state, status words) plus the engine's workspaces, which stop
    growing at `tile_frames` frames (ac3mi_set_tile_frames: larger batches go through in tiles of whole streams).  The
    workspace figure comes from the library itself (ac3mi_transcode_workspace_plan: the allocation's own expressions, no GPU
    needed), so a change of the engine's layout cannot leave this plan behind; tests/test_full_size_gpu.py compares it with
    what a fresh engine really allocates (ac3mi_workspace_bytes)."""
    import ctypes
    from .capi import load_library
    lib = load_library()
    lib.ac3mi_transcode_workspace_plan.restype = ctypes.c_size_t
    lib.ac3mi_transcode_workspace_plan.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    nfr = streams * frames_per_stream
    per_stream = 2 * frames_per_stream * frame_bytes + frames_per_stream * 4      # frames in + out, status
    per_stream += 6 * 128 * 4 + 2 + nch * 256 * 2 + 4                              # overlap tails, dither state, encoder history, search state
    tile = min(nfr, max(tile_frames, frames_per_stream)) if tile_frames else nfr
    nfchans = nch - 1 if nch == 6 else nch                                         # (5.1: five full-bandwidth channels + LFE)
    workspace = int(lib.ac3mi_transcode_workspace_plan(tile, frames_per_stream, nch, nfchans, nch))
    total = streams * per_stream + workspace
    return {"streams": streams, "state_and_io": streams * per_stream, "workspace": workspace, "tile_frames": tile,
            "total": total, "fits": total < HBM_BYTES}
