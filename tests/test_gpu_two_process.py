"""Both multi-GPU data paths executed by TWO PROCESSES (one GPU, gloo + host staging): the camera-sharded pipeline
(ShardedStitcher.submit / finish over both pyramid sets, a recalibration in the middle) and the canvas strips
(StripStitcher + gather_strips_to_root), every panorama against the oracle on rank 0.  See tests/two_rank_worker.py.
The reference's split this mirrors: `#pragma omp parallel for` over the cameras, then one composite (DRV:736-759)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["cameras", "strips"])
def test_two_processes_share_the_gpu(mode):
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_worker.py"), mode, str(r), str(port)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d (%s): rc %s\n%s\n%s" % (r, mode, p.returncode, o[-1500:], e[-3000:])
        assert "rank %d %s ok" % (r, mode) in o
