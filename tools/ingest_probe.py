"""Where the shared-memory ingest's time goes: each stage of IngestStitcher.submit alone and in pairs (one GPU box)."""
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stitchingvideo_amd as S                      # noqa: E402
from stitchingvideo_amd import ingest               # noqa: E402

n_cam, fw, fh = 6, 1920, 1080
dev = torch.device("cuda", 0)
names = ["/mst_probe_%d_%d" % (os.getpid(), i) for i in range(n_cam)]
procs = [subprocess.Popen([sys.executable, "-m", "stitchingvideo_amd.ingest", names[i], str(i), str(fw), str(fh), "bgr", "3", "0"],
                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT) for i in range(n_cam)]
for p in procs:
    assert p.stdout.readline().strip() == "ready"
clients = [ingest.FrameClient(n) for n in names]
sets = [[torch.empty((fh, fw, 3), dtype=torch.uint8, device=dev) for _ in range(n_cam)] for _ in range(3)]
s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
d_pano = torch.empty((1020, 7852, 3), dtype=torch.uint8, device=dev)
h_panos = [torch.empty((1020, 7852, 3), dtype=torch.uint8, pin_memory=True) for _ in range(3)]
pinned = [torch.empty((fh, fw, 3), dtype=torch.uint8, pin_memory=True) for _ in range(n_cam)]


def rate(fn, n=300, warm=20):
    for t in range(warm):
        fn(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        fn(t)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def get(t, prefetch):
    fs = sets[t % 3]
    ingest.get_device_multi(clients, [x.data_ptr() for x in fs], [x.stride(0) for x in fs], [x.numel() for x in fs], stream=s_in.cuda_stream,
                            timeout_ms=20000, prefetch=prefetch)


def h2d_pinned(t):
    with torch.cuda.stream(s_in):
        for i in range(n_cam):
            sets[t % 3][i].copy_(pinned[i], non_blocking=True)
    s_in.synchronize()


def d2h(t):
    with torch.cuda.stream(s_out):
        h_panos[t % 3].copy_(d_pano, non_blocking=True)


print("multi-get alone, no prefetch     %.0f sets/s" % rate(lambda t: get(t, False)))
print("multi-get alone, prefetch        %.0f sets/s" % rate(lambda t: get(t, True)))
print("H2D of 6 hipHostMalloc frames    %.0f sets/s" % rate(h2d_pinned))
print("D2H of a panorama alone          %.0f /s" % rate(lambda t: (d2h(t), s_out.synchronize())))
print("H2D pinned + D2H in flight       %.0f sets/s" % rate(lambda t: (d2h(t), h2d_pinned(t))))
print("multi-get (prefetch) + D2H       %.0f sets/s" % rate(lambda t: (d2h(t), get(t, True))))
# ---- bisect: the pull beside a copy-out, adding the loop's pieces one at a time ----
evs = [torch.cuda.Event() for _ in range(3)]


def b_a(t): d2h(t); get(t, True)
def b_b(t): get(t, True); d2h(t)
def b_c(t): d2h(t); evs[t % 3].record(s_out); get(t, True)
def b_d(t):
    if t >= 3:
        evs[t % 3].synchronize()
    d2h(t); evs[t % 3].record(s_out); get(t, True)
def b_e(t):
    if t >= 3:
        evs[t % 3].synchronize()
    get(t, True); d2h(t); evs[t % 3].record(s_out)
def b_f(t):                      # a different device source per slot, as the loop has
    with torch.cuda.stream(s_out):
        h_panos[t % 3].copy_(d3[t % 3], non_blocking=True)
    get(t, True)


d3 = [torch.empty((1020, 7852, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
for nm, fn in (("a: copy-out, pull", b_a), ("b: pull, copy-out", b_b), ("c: a + event record", b_c), ("d: c + event wait at slot re-use", b_d),
               ("e: pull, copy-out, record, wait", b_e), ("f: a with one device source per slot", b_f), ("a again", b_a)):
    print("%-40s %.0f sets/s" % (nm, rate(fn, n=400)))

for c in clients:
    c.close()
for p in procs:
    p.kill()

# ---- the real loop, by parts (host time per call of each section) ----
from stitchingvideo_amd import rig as rigmod        # noqa: E402
rig = rigmod.build_rig(rigmod.CONFIGS["C3"])
procs = [subprocess.Popen([sys.executable, "-m", "stitchingvideo_amd.ingest", names[i], str(i), str(fw), str(fh), "bgr", "3", "0"],
                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT) for i in range(n_cam)]
for p in procs:
    assert p.stdout.readline().strip() == "ready"
eng = S.PanoramaEngine(6, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, device=0, overlap_frames=True, overlap_depth=4)
eng.update(rig["scale"], rig["cameras"])
st = ingest.IngestStitcher(eng, names, [c["size"] for c in rig["cameras"]], device=0)
acc = {"get": 0.0, "compose": 0.0, "copy_out": 0.0, "wait": 0.0}
_get = ingest.get_device_multi
_ce = eng.compose_device_batch_ex


def timed(key, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[key] += time.perf_counter() - t
        return r
    return w


ingest.get_device_multi = timed("get", _get)
eng.compose_device_batch_ex = timed("compose", _ce)
st.wait = timed("wait", st.wait)


def loop(n):
    for t in range(n):
        if t >= st.depth:
            st.wait()
        st.submit()
    while st.completed < st.submitted:
        st.wait()


loop(30)
for k in acc:
    acc[k] = 0.0
t0 = time.perf_counter()
N = 600
loop(N)
dt = time.perf_counter() - t0
print("IngestStitcher: %.0f sets/s; per set: total %.3f ms, get %.3f, compose call %.3f, wait %.3f, rest %.3f" %
      (N / dt, dt / N * 1e3, acc["get"] / N * 1e3, acc["compose"] / N * 1e3, acc["wait"] / N * 1e3, (dt - sum(acc.values())) / N * 1e3))
st.close()
eng.close()
for p in procs:
    p.kill()

# ---- the copy-out beside the pull: what makes the two directions share the link badly ----
procs = [subprocess.Popen([sys.executable, "-m", "stitchingvideo_amd.ingest", names[i], str(i), str(fw), str(fh), "bgr", "3", "0"],
                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT) for i in range(n_cam)]
for p in procs:
    assert p.stdout.readline().strip() == "ready"
clients = [ingest.FrameClient(n) for n in names]
ingest.get_device_multi = _get
s_c = torch.cuda.Stream(device=dev)
d_panos = [torch.empty((1020, 7852, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
eng = S.PanoramaEngine(6, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, device=0, overlap_frames=True, overlap_depth=4)
eng.update(rig["scale"], rig["cameras"])
eng.acquire()
ev_in = [torch.cuda.Event() for _ in range(3)]
ev_c = [torch.cuda.Event() for _ in range(3)]
ev_out = [torch.cuda.Event() for _ in range(3)]


def compose(k, out):
    fs = sets[k]
    eng.compose_device_batch_ex([[x.data_ptr() for x in fs]], [[x.stride(0) for x in fs]], [out.data_ptr()], out.stride(0),
                                stream=s_c.cuda_stream, frames_ready=ev_in[k].cuda_event, outputs_free=None)


def step(t, mode):
    k = t % 3
    if mode == "as shipped":
        if t >= 3:
            ev_out[k].synchronize()
        get(t, True)
        ev_in[k].record(s_in)
        compose(k, d_panos[k])
        ev_c[k].record(s_c)
        s_out.wait_event(ev_c[k])
        with torch.cuda.stream(s_out):
            h_panos[k].copy_(d_panos[k], non_blocking=True)
        ev_out[k].record(s_out)
    elif mode == "copy-out on the compose stream":
        if t >= 3:
            ev_out[k].synchronize()
        get(t, True)
        ev_in[k].record(s_in)
        compose(k, d_panos[k])
        with torch.cuda.stream(s_c):
            h_panos[k].copy_(d_panos[k], non_blocking=True)
        ev_out[k].record(s_c)
    elif mode == "panorama written straight into page-locked host memory":
        if t >= 3:
            ev_out[k].synchronize()
        get(t, True)
        ev_in[k].record(s_in)
        compose(k, h_panos[k])
        ev_out[k].record(s_c)
    elif mode == "copy-out queued before the pull":
        # set t - 1's compose and copy-out are queued first, then set t is pulled beside them
        if t >= 3:
            ev_out[k].synchronize()
        kp = (t - 1) % 3
        if t >= 1:
            compose(kp, d_panos[kp])
            ev_c[kp].record(s_c)
            s_out.wait_event(ev_c[kp])
            with torch.cuda.stream(s_out):
                h_panos[kp].copy_(d_panos[kp], non_blocking=True)
            ev_out[kp].record(s_out)
        get(t, True)
        ev_in[k].record(s_in)


def step2(t, lag, wait_compose):
    """copy-out of panorama t - lag queued BEFORE set t is pulled (and, wait_compose, only once its compose has finished)"""
    k = t % 3
    kp = (t - lag) % 3
    if t >= lag:
        if wait_compose:
            ev_c[kp].synchronize()
        s_out.wait_event(ev_c[kp])
        with torch.cuda.stream(s_out):
            h_panos[kp].copy_(d_panos[kp], non_blocking=True)
        ev_out[kp].record(s_out)
    get(t, True)
    ev_in[k].record(s_in)
    if t >= 3:
        ev_out[k].synchronize()          # (the copy-out that last read this slot's panorama buffer)
    compose(k, d_panos[k])
    ev_c[k].record(s_c)


for lag, wc in ((2, True), (1, True), (2, True), (2, False), (3, False), (2, True), (1, False), (3, True)):
    print("copy-out of t - %d first%-28s %.0f sets/s" % (lag, ", after its compose finished" if wc else "", rate(lambda t: step2(t, lag, wc), n=400)))
for mode in ("as shipped", "copy-out on the compose stream", "panorama written straight into page-locked host memory", "copy-out queued before the pull",
             "as shipped"):
    print("%-58s %.0f sets/s" % (mode, rate(lambda t: step(t, mode), n=400)))
eng.close()
for c in clients:
    c.close()
for p in procs:
    p.kill()
