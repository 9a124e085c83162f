"""Multi-GPU host logic: one process per GPU over torch.distributed (backend "nccl" == RCCL on ROCm).

The per-frame path shards over independent units with no data-path collective:

  frames  -- every rank owns the full calibration and stitches its own sub-sequence of frame sets
             (rank r takes global frames r, r + world, r + 2*world, ...).  This is what bench.py
             runs for N > 1: aggregate frames/s, "scaling": "weak".
  streams -- camera i belongs to rank i % world; used by the camera-sharded mode,
             where the owner of the canvas gathers the other ranks' contributions once per frame.

Only timing/bookkeeping values cross ranks here (a MAX all-reduce of the elapsed time).
"""
import torch
import torch.distributed as dist


def global_frame_index(rank, world, step):
    """Global index of the frame set that `rank` stitches at its local step `step`."""
    return step * world + rank


def frames_of_rank(rank, world, total_frames):
    return list(range(rank, total_frames, world))


def streams_of_rank(rank, world, num_cameras):
    """Camera (stream) indices owned by `rank`: camera i -> rank i % world."""
    return [i for i in range(num_cameras) if i % world == rank]


def owner_of_stream(camera, world):
    return camera % world


def max_over_ranks(values, device=None):
    """MAX all-reduce of a list of floats (elapsed times); identity when not initialised."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.cpu()]


def aggregate_throughput(world, steps_per_rank, max_elapsed_s):
    """Whole-job frames/s: units all ranks processed / slowest rank's time."""
    return world * steps_per_rank / max_elapsed_s


def gather_stream_tiles(local_tiles, owners, root, world, rank, device=None):
    """One gather per frame of per-stream tiles to `root`.

    local_tiles: {camera: uint8/int16 tensor} for the cameras this rank owns.
    owners: list camera -> rank.  Returns {camera: tensor} on root (all cameras), {} elsewhere.
    Ragged tiles, so point-to-point sends grouped per frame (each peer uses its own xGMI link to
    the root); shapes are static per calibration and known to every rank, so no size exchange.
    """
    out = {}
    if world == 1:
        return dict(local_tiles)
    reqs = []
    if rank == root:
        for cam, owner in enumerate(owners):
            if owner == root:
                out[cam] = local_tiles[cam]
        for cam, owner in enumerate(owners):
            if owner != root:
                shape, dtype = gather_stream_tiles.shapes[cam]
                buf = torch.empty(shape, dtype=dtype, device=device)
                reqs.append(dist.irecv(buf, src=owner, tag=cam))
                out[cam] = buf
    else:
        for cam, t in local_tiles.items():
            reqs.append(dist.isend(t.contiguous(), dst=root, tag=cam))
    for r in reqs:
        r.wait()
    return out


gather_stream_tiles.shapes = {}


def register_tile_shapes(shapes):
    """shapes: {camera: (shape tuple, torch dtype)} -- static per calibration, identical on all ranks."""
    gather_stream_tiles.shapes = dict(shapes)
