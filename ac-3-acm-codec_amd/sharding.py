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
    growing at `tile_frames` frames (ac3mi_set_tile_frames: larger batches go through in tiles of whole streams).
    Per tile frame: coefficient planes + block-switch flags (decode front end -> transform), the split front end's
    descriptors / rows / coupling coordinates / generator positions, s16 PCM (transform -> encoder), and the encoder's MDCT
    coefficients, exponents, masks, strategies, search results (csrc/capi.hip: ensure_ws, split_bytes, ac3mi_transcode_batch)."""
    nfr = streams * frames_per_stream
    per_stream = 2 * frames_per_stream * frame_bytes + frames_per_stream * 4      # frames in + out, status
    per_stream += 6 * 128 * 4 + 2 + nch * 256 * 2 + 4                              # overlap tails, dither state, encoder history, search state
    rows = 6 * nch
    per_tile_frame = 6 * nch * 256 * 4 + 6 * 5 + 1                                # planes, blksw, zs
    per_tile_frame += 6 * 80 + 16 + 6 * 90 * 4 + 6 * 7 * 512                      # split front end
    per_tile_frame += 1536 * nch * 2                                              # s16 PCM
    per_tile_frame += rows * (1024 + 256 + 256 + 100 + 1 + 1) + nch * 4 + 8 + 32 # encoder
    tile = min(nfr, max(tile_frames, frames_per_stream)) if tile_frames else nfr
    return {"streams": streams, "state_and_io": streams * per_stream, "workspace": tile * per_tile_frame,
            "total": streams * per_stream + tile * per_tile_frame, "fits": streams * per_stream + tile * per_tile_frame < HBM_BYTES}
