"""Multi-GPU host logic: one process per GPU over torch.distributed (backend "nccl" == RCCL on ROCm).

The per-frame path shards over independent units with no data-path collective:

  frames  -- every rank owns the full calibration and stitches its own sub-sequence of frame sets
             (rank r takes global frames r, r + world, r + 2*world, ...).  This is what bench.py
             runs for N > 1: aggregate frames/s, "scaling": "weak".
  streams -- camera i belongs to rank i % world; used by the camera-sharded mode,
             where the owner of the canvas gathers the other ranks' contributions once per frame.

  strips  -- the CANVAS is cut into `world` column strips (second form): rank r composes strip r from
             the cameras that reach it (every kernel of the path restricted to what the strip depends on, bit-exact),
             and the strips are concatenated -- on the root by one grouped point-to-point step, or left where they are.

Only timing/bookkeeping values cross ranks here (a MAX all-reduce of the elapsed time) besides those exchanges.
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


def host_staged():
    """True when the process group cannot move device memory (gloo): the exchanges then stage through pinned host buffers.  That is
    the transport of the two-process ONE-GPU tests (RCCL refuses two ranks of a communicator on one device) -- never of a real
    multi-GPU run, where the backend is "nccl" (RCCL) and the tensors stay in HBM."""
    return dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl"


def exchange_to_root(send, recv, owners, root, rank, tag_base=0):
    """One exchange step: every non-root owner sends its cameras' payload tensors to `root`.

    send: {camera: tensor} for the cameras this rank owns (ignored on root); recv: {camera: tensor} landing
    buffers on root for the cameras it does not own.  All transfers of the step go out as ONE
    batch_isend_irecv (ncclGroupStart/End on RCCL: every peer streams over its own xGMI link to the root at the
    same time).  Returns the list of work handles (empty when there is nothing to move)."""
    ops = []
    if rank == root:
        for c, owner in enumerate(owners):
            if owner != root:
                ops.append(dist.P2POp(dist.irecv, recv[c], owner, tag=tag_base + c))
    else:
        for c in sorted(send):
            ops.append(dist.P2POp(dist.isend, send[c], root, tag=tag_base + c))
    return dist.batch_isend_irecv(ops) if ops else []


class ShardedStitcher:
    """Camera streams sharded one-per-rank, one exchange per frame, final blend on `root`.

    Every rank creates the same PanoramaEngine (max_batch >= 2 for pipelining) and publishes the same calibration.
    Per frame t, on pyramid set t % 2:
      1. each rank runs the per-camera kernels for the cameras it owns (mst_stage_cameras) and packs their
         hand-off payload -- the interiors of the launched tiles of every level, a few MB per camera whatever the
         size of its pyramid block (mst_payload_pack);
      2. the owners send the packed payloads to the root (grouped point-to-point over RCCL: every peer has its
         own xGMI link to the root; no all-reduce -- the payload is ragged and must arrive bit-exact);
      3. the root unpacks them into its own pyramids (mst_payload_unpack) and runs the canvas kernel.
    compose() does 1-3 for one frame.  submit() / finish() split it so that the exchange of frame t is in flight
    while the per-camera stage of frame t+1 runs (two pyramid sets, two sets of payload buffers).

    Payload sizes and layout depend on the calibration only: compose() latches the calibration
    (engine.acquire()) and re-sizes its buffers whenever the generation moved.  Every rank must publish the same
    calibrations in the same order.
    """

    def __init__(self, engine, rank, world, root=0, device=None, sets=2):
        self.engine, self.rank, self.world, self.root = engine, rank, world, root
        self.device = device
        n = engine.num_cameras
        self.owners = [owner_of_stream(c, world) for c in range(n)]
        self.mine = streams_of_rank(rank, world, n)
        self.sets = sets
        self.generation = None
        self.bufs = []               # per set: {camera: uint8 tensor}: send buffers (owners) / landing buffers (root)
        self.hbufs = None            # their pinned host mirrors when the process group cannot move device memory
        self.pending = {}            # set -> (work handles, pano args)

    def refresh(self):
        """Latch the newest calibration; (re)allocate the payload buffers if it changed."""
        _, gen = self.engine.acquire()
        if gen != self.generation:
            # exchanges still in flight send from / receive into the buffers replaced below: drain them first (their
            # frames belong to the old calibration and are dropped)
            for works in self.pending.values():
                for w in works or []:
                    w.wait()
            n = self.engine.num_cameras
            # the root needs a landing buffer for every camera it does not own, an owner a send buffer for its own
            cams = [c for c in range(n) if self.owners[c] != self.root] if self.rank == self.root else self.mine
            self.bufs = [{c: torch.empty(self.engine.payload_bytes(c), dtype=torch.uint8, device=self.device) for c in cams}
                         for _ in range(self.sets)]
            # (gloo: pinned host mirrors of the same buffers -- see host_staged())
            self.hbufs = [{c: torch.empty(t.numel(), dtype=torch.uint8).pin_memory() for c, t in b.items()} for b in self.bufs] \
                if self.world > 1 and host_staged() else None
            self.generation = gen
            self.pending = {}
        return gen

    def payload_bytes(self):
        """Bytes the root receives per frame (what this rank sends, elsewhere); 0 on one rank."""
        if self.generation is None:
            self.refresh()
        return sum(int(t.numel()) for t in self.bufs[0].values()) if self.bufs else 0

    def submit(self, t, frame_ptrs, frame_strides, stream=None):
        """Frame t, steps 1-2: per-camera stage + pack on this rank, start the exchange.  The kernels and the
        point-to-point transfers must share one stream order: RCCL orders its operations against torch's CURRENT
        stream, so `stream` defaults to exactly that and a caller that passes one must have made it current."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.generation is None:
            self.refresh()
        k = t % self.sets
        self.engine.select_stage_set(k)
        self.engine.stage_cameras(self.mine, frame_ptrs, frame_strides, stream=stream)
        works = []
        if self.world > 1:
            if self.rank != self.root:
                for c in self.mine:
                    self.engine.payload_pack(c, self.bufs[k][c].data_ptr(), stream=stream)
            if self.hbufs is not None:
                # host-staged transport: the packed payload leaves HBM on the same stream as the pack, and the stream is drained
                # before the host tensors are handed to the process group
                if self.rank != self.root:
                    for c in self.mine:
                        self.hbufs[k][c].copy_(self.bufs[k][c], non_blocking=True)
                    torch.cuda.current_stream(self.device).synchronize()
                works = exchange_to_root(self.hbufs[k], self.hbufs[k], self.owners, self.root, self.rank, tag_base=64 * k)
            else:
                works = exchange_to_root(self.bufs[k], self.bufs[k], self.owners, self.root, self.rank, tag_base=64 * k)
        self.pending[k] = works

    def finish(self, t, pano_ptr=None, pano_stride=0, stream=None):
        """Frame t, step 3: wait for its exchange, unpack, blend (root only)."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        k = t % self.sets
        for w in self.pending.pop(k, []):
            w.wait()
        if self.rank == self.root:
            self.engine.select_stage_set(k)
            if self.world > 1:
                for c, owner in enumerate(self.owners):
                    if owner != self.root:
                        if self.hbufs is not None:
                            self.bufs[k][c].copy_(self.hbufs[k][c], non_blocking=True)         # (same stream as the unpack below)
                        self.engine.payload_unpack(c, self.bufs[k][c].data_ptr(), stream=stream)
            self.engine.stage_canvas(pano_ptr, pano_stride, stream=stream)

    def compose(self, frame_ptrs, frame_strides, pano_ptr=None, pano_stride=0, stream=None):
        """One frame, unpipelined.  frame_ptrs / frame_strides: {camera: value} for the cameras this rank owns."""
        self.refresh()
        self.submit(0, frame_ptrs, frame_strides, stream=stream)
        self.finish(0, pano_ptr, pano_stride, stream=stream)


def copy_payload(src_engine, dst_engine, cameras, device, stream=None):
    """One-GPU stand-in for the exchange step: pack the hand-off payload of `cameras` on one engine, unpack it on
    another (tests emulate two ranks with two engines on one device).  Returns the bytes moved."""
    moved = 0
    for c in cameras:
        nbytes = src_engine.payload_bytes(c)
        assert nbytes == dst_engine.payload_bytes(c) and nbytes > 0
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        src_engine.payload_pack(c, buf.data_ptr(), stream=stream)
        dst_engine.payload_unpack(c, buf.data_ptr(), stream=stream)
        moved += nbytes
    return moved


# ---- canvas split into column strips (mst_config.strip_index / strip_count) -------------------------------------

def strip_columns(width, index, count):
    """Columns [x0, x1) of strip `index` of `count` in a canvas `width` columns wide: the host-side mirror of
    Engine::strip_columns (equal shares, boundaries on multiples of 256 columns, the last strip takes the rest).
    For the multi-band blend `width` is the PADDED canvas width (a multiple of 2^bands); the written columns are
    the intersection with the final canvas (mst_strip / PanoramaEngine.strip())."""
    if count <= 1:
        return 0, width

    def bound(i):
        if i >= count:
            return width
        return min(width, (i * width // count + 255) // 256 * 256)
    x0 = bound(index)
    return x0, max(x0, bound(index + 1))


def gather_strips_to_root(strip, rects, root, rank, pano=None):
    """One exchange step of the strip-sharded mode: every rank != root sends its strip (a CONTIGUOUS (H, w_r, 3) uint8
    tensor) to the root, which drops all strips into `pano` ((H, W, 3), its own strip included).  rects[r] = (x, w) of
    rank r's strip inside the panorama -- static per calibration and known to every rank (strip_columns), so no sizes
    travel.  All transfers go out as ONE batch_isend_irecv (ncclGroupStart/End on RCCL: every peer uses its own xGMI
    link to the root).  Returns `pano` on the root, None elsewhere."""
    world = len(rects)
    staged = host_staged()           # (gloo: through host memory; .cpu() drains the producing stream first)
    if rank != root:
        if rects[rank][1] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, strip.cpu() if staged else strip, root, tag=100 + rank)]):
                w.wait()
        return None
    ops, bufs = [], {}
    for r in range(world):
        x, w = rects[r]
        if r == root or w <= 0:
            continue
        bufs[r] = torch.empty((pano.shape[0], w, 3), dtype=torch.uint8, device="cpu" if staged else pano.device)
        ops.append(dist.P2POp(dist.irecv, bufs[r], r, tag=100 + r))
    works = dist.batch_isend_irecv(ops) if ops else []
    x, w = rects[root]
    if w > 0:
        pano[:, x:x + w] = strip
    for wk in works:
        wk.wait()
    for r, b in bufs.items():
        x, w = rects[r]
        pano[:, x:x + w] = b.to(pano.device) if staged else b
    return pano


class StripStitcher:
    """Canvas strips sharded one-per-rank.  Every rank creates its PanoramaEngine with strip=(rank, world), publishes
    the same calibration and holds the frames of the cameras that reach its strip (here: all of them -- the capture
    side delivers a camera's stream to the one or two GPUs whose strips it touches).  compose() runs this rank's
    launch sequence into a strip-sized buffer; gather() concatenates the strips on the root."""

    def __init__(self, engine, rank, world, root=0, device=None):
        self.engine, self.rank, self.world, self.root, self.device = engine, rank, world, root, device
        self.generation = None

    def refresh(self):
        (cx, cy, cw, ch), gen = self.engine.acquire()
        if gen != self.generation:
            sx, sy, sw, sh = self.engine.strip()
            self.canvas = (cw, ch)
            self.x0, self.w = sx - cx, sw
            # rank r's strip, from the same arithmetic every rank runs: (x, w) inside the final canvas
            self.rects = None
            self.strip_buf = torch.empty((ch, max(sw, 1), 3), dtype=torch.uint8, device=self.device)
            self.generation = gen
        return gen

    def set_rects(self, rects):
        """rects[r] = (x, w) of every rank's strip (each rank can compute them from the padded canvas width; the
        multi-band engine reports its own through strip(), so the simplest is one all-gather per calibration)."""
        self.rects = list(rects)

    def exchange_rects(self):
        mine = torch.tensor([self.x0, self.w], dtype=torch.int64, device=self.device if dist.get_backend() == "nccl" else None)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine)
        self.rects = [(int(t[0]), int(t[1])) for t in out]

    def compose(self, frame_ptrs, frame_strides, stream=None):
        """This rank's strip of one panorama into self.strip_buf ((H, w, 3), contiguous)."""
        if self.generation is None:
            self.refresh()
        if self.w <= 0:
            return self.strip_buf
        # the engine addresses the panorama from canvas column 0: hand it the strip buffer shifted left by the strip's origin
        self.engine.compose_device(frame_ptrs, frame_strides, self.strip_buf.data_ptr() - self.x0 * 3, self.strip_buf.stride(0),
                                   stream=stream)
        return self.strip_buf

    def gather(self, pano=None):
        if self.rects is None:
            self.exchange_rects()
        return gather_strips_to_root(self.strip_buf[:, :max(self.w, 0)].contiguous() if self.w != self.strip_buf.shape[1] else self.strip_buf,
                                     self.rects, self.root, self.rank, pano)
