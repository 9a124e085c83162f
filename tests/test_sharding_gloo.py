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
