"""One rank of a TWO-PROCESS run of the multi-GPU data paths on ONE GPU (tests/test_gpu_two_process.py starts two of these).

Both processes open cuda:0; the process group is gloo (RCCL refuses two ranks of one communicator on one device), so the exchanges
stage through pinned host memory (sharding.host_staged) -- everything else is the code an 8-GPU run executes: ShardedStitcher's
submit / finish pipeline over both pyramid sets with a recalibration in the middle (DRV:736-759's split: per-camera work on the
owners, one composite on the root), and StripStitcher + gather_strips_to_root.  Rank 0 holds every panorama against the oracle.

usage: two_rank_worker.py <mode: cameras | strips> <rank> <port>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H                                  # noqa: E402
from stitchingvideo_amd import rig as rigmod         # noqa: E402
from stitchingvideo_amd import sharding              # noqa: E402

WORLD = 2


def rigs():
    """two calibrations of the 8-camera rig (C4 / 8): the second with another focal length -- another canvas, other payload sizes"""
    a = rigmod.scaled_config("C4", 8)
    b = dict(a, focal=a["focal"] * 1.07)
    return rigmod.build_rig(a), rigmod.build_rig(b)


def main():
    mode, rank, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)                   # the engine's launches, the staging copies and torch's ops: ONE stream order
    stream = tstream.cuda_stream
    rig_a, rig_b = rigs()
    n_frames, recal_at = 8, 4                        # frames 0..3 under calibration A, 4..7 under B
    failures = []

    def rig_of(t):
        return rig_a if t < recal_at else rig_b

    if mode == "cameras":
        eng = H.make_engine(rig_a, max_batch=2)      # two pyramid sets: the exchange of frame t overlaps the per-camera stage of t + 1
        st = sharding.ShardedStitcher(eng, rank, WORLD, root=0, device=dev)
        st.refresh()
        assert st.hbufs is not None and len(st.mine) == 4
        panos, keep = {}, {}

        def submit(t):
            fr = rigmod.frames_for(rig_of(t), frame_index=t)
            keep[t] = {c: torch.from_numpy(fr[c]).to(dev) for c in st.mine}
            st.submit(t, {c: x.data_ptr() for c, x in keep[t].items()}, {c: x.stride(0) for c, x in keep[t].items()}, stream=stream)

        def finish(t):
            if rank == 0:
                _, _, cw, ch = eng.canvas()
                panos[t] = torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev)
                st.finish(t, panos[t].data_ptr(), panos[t].stride(0), stream=stream)
            else:
                st.finish(t, stream=stream)

        def recalibrate():
            # every rank publishes the same calibration in the same order; refresh() drains the exchanges still in flight and
            # re-sizes the payload buffers (the frames in flight would belong to the old calibration: here none is)
            eng.update(rig_b["scale"], rig_b["cameras"])
            st.refresh()

        submit(0)
        for t in range(1, n_frames):
            if t == recal_at:
                finish(t - 1)                        # drain the pipeline at the calibration boundary, as a host would
                recalibrate()
                submit(t)
                continue
            submit(t)
            finish(t - 1)
        finish(n_frames - 1)
        tstream.synchronize()
        if rank == 0:
            for t in range(n_frames):
                rect, opano, _ = H.oracle_compose(rig_of(t), rigmod.frames_for(rig_of(t), frame_index=t), threads=4)
                got = panos[t].cpu().numpy()
                if got.shape != opano.shape or not np.array_equal(got, opano):
                    failures.append("cameras: frame %d differs from the oracle: %s" % (t, H.diff_report(got, opano) if got.shape == opano.shape else (got.shape, opano.shape)))
            assert panos[0].shape != panos[n_frames - 1].shape, "the recalibration was meant to change the canvas"
        eng.close()
    else:
        eng = H.make_engine(rig_a, strip=(rank, WORLD))
        ss = sharding.StripStitcher(eng, rank, WORLD, root=0, device=dev)
        for t in range(n_frames):
            if t == recal_at:
                eng.update(rig_b["scale"], rig_b["cameras"])
            ss.refresh()                             # latches the newest calibration; a new one re-sizes the strip and drops the rects
            fr = rigmod.frames_for(rig_of(t), frame_index=t)
            dfr = [torch.from_numpy(f).to(dev) for f in fr]
            ss.compose([x.data_ptr() for x in dfr], [x.stride(0) for x in dfr], stream=stream)
            tstream.synchronize()
            cw, ch = ss.canvas
            pano = torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
            ss.gather(pano)
            if rank == 0:
                rect, opano, _ = H.oracle_compose(rig_of(t), fr, threads=4)
                got = pano.cpu().numpy()
                if got.shape != opano.shape or not np.array_equal(got, opano):
                    failures.append("strips: frame %d differs from the oracle: %s" % (t, H.diff_report(got, opano) if got.shape == opano.shape else (got.shape, opano.shape)))
                assert all(w > 0 for _, w in ss.rects), ss.rects
        eng.close()
    dist.barrier()
    dist.destroy_process_group()
    if failures:
        sys.exit("\n".join(failures))
    print("rank %d %s ok" % (rank, mode))


if __name__ == "__main__":
    main()
