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


class _DevicePtr:
    """Zero-copy view of engine-owned device memory as a torch uint8 tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def pyramid_tensor(engine, camera, device):
    ptr, nbytes = engine.pyramid_buffer(camera)
    return torch.as_tensor(_DevicePtr(ptr, nbytes), device=device)


class ShardedStitcher:
    """Camera streams sharded one-per-rank, one gather per frame, final blend on `root`.

    Every rank creates the same PanoramaEngine and publishes the same calibration.  Per frame:
      1. each rank runs the per-camera kernels for the cameras it owns (mst_stage_cameras);
      2. the owners send those cameras' Gaussian pyramid blocks straight into the root engine's buffers
         (grouped point-to-point over RCCL: every peer has its own xGMI link to the root; no all-reduce --
         the payload is ragged and must arrive bit-exact);
      3. the root runs the canvas kernel (mst_stage_canvas).
    """

    def __init__(self, engine, rank, world, root=0, device=None):
        self.engine, self.rank, self.world, self.root = engine, rank, world, root
        self.device = device
        n = engine.num_cameras
        self.owners = [owner_of_stream(c, world) for c in range(n)]
        self.mine = streams_of_rank(rank, world, n)
        self.views = {c: pyramid_tensor(engine, c, device) for c in range(n)}

    def compose(self, frame_ptrs, frame_strides, pano_ptr=None, pano_stride=0, stream=None):
        """frame_ptrs / frame_strides: {camera: value} for the cameras this rank owns."""
        self.engine.stage_cameras(self.mine, frame_ptrs, frame_strides, stream=stream)
        if self.world > 1:
            reqs = []
            if self.rank == self.root:
                for c, owner in enumerate(self.owners):
                    if owner != self.root:
                        reqs.append(dist.irecv(self.views[c], src=owner, tag=c))
            else:
                for c in self.mine:
                    reqs.append(dist.isend(self.views[c], dst=self.root, tag=c))
            for r in reqs:
                r.wait()
        if self.rank == self.root:
            self.engine.stage_canvas(pano_ptr, pano_stride, stream=stream)
