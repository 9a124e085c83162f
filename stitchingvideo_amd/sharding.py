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

    The pyramid buffers belong to a calibration state and are freed when it is retired, so the zero-copy views
    are tied to the calibration generation: compose() latches the calibration (engine.acquire()) and rebuilds
    the views whenever the generation moved.  Every rank must publish the same calibrations in the same order.
    """

    def __init__(self, engine, rank, world, root=0, device=None):
        self.engine, self.rank, self.world, self.root = engine, rank, world, root
        self.device = device
        n = engine.num_cameras
        self.owners = [owner_of_stream(c, world) for c in range(n)]
        self.mine = streams_of_rank(rank, world, n)
        self.generation = None
        self.views = {}

    def refresh(self):
        """Latch the newest calibration; (re)build the views of its pyramid buffers if it changed."""
        _, gen = self.engine.acquire()
        if gen != self.generation:
            cams = range(self.engine.num_cameras) if self.rank == self.root else self.mine
            self.views = {c: pyramid_tensor(self.engine, c, self.device) for c in cams}
            self.generation = gen
        return gen

    def payload_bytes(self):
        """Bytes the root receives per frame (0 on one rank); valid after the first refresh()/compose()."""
        if self.generation is None:
            self.refresh()
        if self.rank != self.root:
            return sum(int(self.views[c].numel()) for c in self.mine)
        return sum(int(self.views[c].numel()) for c, o in enumerate(self.owners) if o != self.root)

    def compose(self, frame_ptrs, frame_strides, pano_ptr=None, pano_stride=0, stream=None):
        """frame_ptrs / frame_strides: {camera: value} for the cameras this rank owns.

        The kernels and the point-to-point transfers must share one stream order: RCCL orders its operations
        against torch's CURRENT stream only, so `stream` defaults to exactly that (never the engine's private
        stream) and a caller that passes one must have made it torch's current stream."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self.refresh()
        self.engine.stage_cameras(self.mine, frame_ptrs, frame_strides, stream=stream)
        if self.world > 1:
            ops = []
            if self.rank == self.root:
                for c, owner in enumerate(self.owners):
                    if owner != self.root:
                        ops.append(dist.P2POp(dist.irecv, self.views[c], owner, tag=c))
            else:
                for c in self.mine:
                    ops.append(dist.P2POp(dist.isend, self.views[c], self.root, tag=c))
            if ops:
                for r in dist.batch_isend_irecv(ops):      # one ncclGroupStart/End around all transfers of the frame
                    r.wait()
        if self.rank == self.root:
            self.engine.stage_canvas(pano_ptr, pano_stride, stream=stream)


def copy_payload(src_engine, dst_engine, cameras, device, stream=None):
    """One-GPU stand-in for the exchange step: copy the hand-off payload of `cameras` from one engine into
    another (tests emulate two ranks with two engines on one device)."""
    for c in cameras:
        src, dst = pyramid_tensor(src_engine, c, device), pyramid_tensor(dst_engine, c, device)
        assert src.numel() == dst.numel() and src.numel() > 0
        dst.copy_(src)
