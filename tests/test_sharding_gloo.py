"""world_size-2 gloo tests (CPU) of the N > 1 host logic: frame/stream sharding, the MAX-over-ranks
timing reduction bench.py uses, and the per-frame gather of stream tiles to the blending rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stitchingvideo_amd import sharding as sh


def test_shard_maps_cover_everything_once():
    for world in (1, 2, 4, 8):
        frames = sorted(f for r in range(world) for f in sh.frames_of_rank(r, world, 37))
        assert frames == list(range(37))
        cams = sorted(c for r in range(world) for c in sh.streams_of_rank(r, world, 8))
        assert cams == list(range(8))
        assert all(sh.owner_of_stream(c, world) == c % world for c in range(8))
        assert [sh.global_frame_index(r, world, 3) for r in range(world)] == [3 * world + r for r in range(world)]
    assert sh.aggregate_throughput(8, 100, 2.0) == 400.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # timing reduction: slowest rank wins
        mx = sh.max_over_ranks([1.0 + rank, 10.0 - rank])
        # one gather per frame of ragged per-stream tiles to rank 0
        n_cam = 5
        owners = [sh.owner_of_stream(c, world) for c in range(n_cam)]
        shapes = {c: ((3, 4 + c, 6), torch.int16) for c in range(n_cam)}
        sh.register_tile_shapes(shapes)
        mine = {c: torch.full(shapes[c][0], 100 * c + 7, dtype=torch.int16) for c in sh.streams_of_rank(rank, world, n_cam)}
        got = sh.gather_stream_tiles(mine, owners, 0, world, rank)
        ok = True
        if rank == 0:
            ok = sorted(got) == list(range(n_cam)) and all(
                got[c].shape == shapes[c][0] and bool((got[c] == 100 * c + 7).all()) for c in range(n_cam))
        else:
            ok = got == {}
        dist.barrier()
        q.put((rank, mx, ok))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_reduction_and_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mx, ok in res:
        assert mx == [2.0, 10.0], mx
        assert ok, rank


# ---- an engine-shaped exchange on CPU: ragged per-camera payloads, blended panorama checked against the oracle ----
def _payload_worker(rank, world, port, q):
    """Each rank warps + gain-compensates the cameras it owns (oracle arithmetic; on the GPU box this is
    mst_stage_cameras + mst_payload_pack), the owners send their ragged payloads to rank 0 through the same
    exchange_to_root() the GPU path uses (grouped point-to-point), and rank 0 feeds the blender and must land on the
    panorama the single-process oracle produces."""
    import sys
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import helpers as H
        import oracle_lib as O
        from stitchingvideo_amd import rig as rigmod
        rig = rigmod.build_rig(rigmod.scaled_config("C4", 12))         # 8 cameras, one of them wrapping the canvas
        frames = rigmod.frames_for(rig, 3)
        n = len(frames)
        owners = [sh.owner_of_stream(c, world) for c in range(n)]
        mine = sh.streams_of_rank(rank, world, n)
        kind = H.WARP_O[rig["warp"]]
        # static per calibration, identical on every rank: payload shapes = warped roi sizes (ragged)
        shapes = {c: (rig["sizes"][c][1], rig["sizes"][c][0], 3) for c in range(n)}
        send = {}
        for c in mine:
            cam = rig["cameras"][c]
            _, warped = O.warp(kind, rig["scale"], cam["K"], cam["R"], frames[c], O.INTER_LINEAR, O.BORDER_REFLECT)
            send[c] = torch.from_numpy(O.blocks_gain_apply(warped, cam["gain_map"]).astype(np.int16))
        recv = {c: torch.empty(shapes[c], dtype=torch.int16) for c in range(n) if owners[c] != 0} if rank == 0 else {}
        for w in sh.exchange_to_root(send, recv, owners, 0, rank):
            w.wait()
        ok = True
        if rank == 0:
            tiles = {c: (send[c] if owners[c] == 0 else recv[c]).numpy() for c in range(n)}
            b = O.Blender(O.BLEND_MULTI_BAND, 4)
            b.prepare(rig["corners"], rig["sizes"])
            for c in range(n):
                b.feed(tiles[c], rig["cameras"][c]["mask"], rig["corners"][c])
            _, pano16, pmask = b.blend()
            pano = np.clip(pano16, 0, 255).astype(np.uint8)
            rect, opano, omask = H.oracle_compose(rig, frames, num_bands=4)
            ok = bool(np.array_equal(pano, opano) and np.array_equal(pmask, omask))
            ok = ok and any(shapes[c] != shapes[0] for c in range(n))                 # the payload really is ragged
        dist.barrier()
        q.put((rank, ok))
    except Exception as e:          # noqa: BLE001  -- report instead of leaving the parent to time out
        q.put((rank, "%s: %s" % (type(e).__name__, e)))
        raise
    finally:
        dist.destroy_process_group()


def test_gloo_world2_ragged_payload_blends_to_the_oracle_panorama():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_payload_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok is True for _, ok in res), res
    assert all(p.exitcode == 0 for p in procs)


# ---- canvas strips: ragged column strips of the oracle's panorama, concatenated on the root --------------------
def test_strip_columns_partition_the_canvas():
    for width in (256, 1000, 3259, 7872, 11328, 15712):
        for count in (1, 2, 3, 4, 8):
            xs = [sh.strip_columns(width, k, count) for k in range(count)]
            assert xs[0][0] == 0 and xs[-1][1] == width
            assert all(a[1] == b[0] for a, b in zip(xs, xs[1:]))                    # no gap, no overlap
            assert all(x0 % 256 == 0 or x0 == width for x0, _ in xs)          # (x0 == width: an empty strip of a narrow canvas)
            assert all(x1 >= x0 for x0, x1 in xs)


def _strip_worker(rank, world, port, q):
    """Every rank holds its column strip of the panorama (here cut from the oracle's; on the GPU box a
    PanoramaEngine(strip=(rank, world)) composes exactly these columns -- tests/test_gpu_strips.py), the strips go to
    rank 0 through the same gather_strips_to_root() the GPU path uses, and rank 0 must hold the whole panorama."""
    import sys
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import helpers as H
        from stitchingvideo_amd import rig as rigmod
        rig = rigmod.build_rig(rigmod.scaled_config("C3", 6))
        frames = rigmod.frames_for(rig, 1)
        rect, opano, _ = H.oracle_compose(rig, frames, num_bands=3)
        cw, ch = rect[2], rect[3]
        padded = (cw + 7) // 8 * 8                                    # 3 bands: the canvas padded to a multiple of 8
        rects = []
        for r in range(world):
            x0, x1 = sh.strip_columns(padded, r, world)
            rects.append((min(x0, cw), min(x1, cw) - min(x0, cw)))
        x, w = rects[rank]
        mine = torch.from_numpy(np.ascontiguousarray(opano[:, x:x + w]))
        pano = torch.zeros((ch, cw, 3), dtype=torch.uint8) if rank == 0 else None
        out = sh.gather_strips_to_root(mine, rects, 0, rank, pano)
        ok = True
        if rank == 0:
            ok = bool(np.array_equal(out.numpy(), opano)) and len({w for _, w in rects}) > 1      # ragged strips
        else:
            ok = out is None
        dist.barrier()
        q.put((rank, ok))
    except Exception as e:          # noqa: BLE001
        q.put((rank, "%s: %s" % (type(e).__name__, e)))
        raise
    finally:
        dist.destroy_process_group()


def test_gloo_world2_strips_concatenate_to_the_oracle_panorama():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok is True for _, ok in res), res
    assert all(p.exitcode == 0 for p in procs)
