"""The reference's shipped x64 machine code, RUN in place, against the oracle (tests/ref_run.py; oracle/ref_run/pe_run.c): fresh random
inputs every primitive the runner can reach without an import, and the committed golden file regenerated bit for bit.  Needs
/root/reference: skipped on the GPU box, where tests/test_ref_goldens.py checks the committed vectors instead."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402
import ref_run as R             # noqa: E402

pytestmark = pytest.mark.skipif(not R.available(), reason="opt-in: needs the reference tree (/root/reference) and MST_RUN_REFERENCE_BINARIES=1 "
                                                           "(its binaries are run in place, sha256-pinned, in a seccomp-confined child)")


def test_goldens_regenerate_bit_for_bit():
    import make_ref_goldens as M
    fresh = M.cases()
    with np.load(M.OUT) as z:
        assert sorted(z.files) == sorted(fresh)
        for k in z.files:
            assert fresh[k].dtype == z[k].dtype and np.array_equal(fresh[k].view(np.uint8), z[k].view(np.uint8)), k


def test_pipeline_goldens_regenerate_bit_for_bit():
    """the reference's blender / whole functions / compose loop, run again: the committed panoramas come back byte for byte"""
    import make_ref_pipeline_goldens as M
    for fresh, path in ((M.cases(), M.OUT), (M.stills_case(), M.OUT_STILLS)):
        with np.load(path) as z:
            assert sorted(z.files) == sorted(fresh)
            for k in z.files:
                a, b = np.asarray(fresh[k]), z[k]
                assert a.shape == b.shape and np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), k


def test_full_size_digests_regenerate():
    """C2 / C3 / C4 / C5 at full size through the reference's compose loop again: the committed SHA-256s come back"""
    import json
    import make_ref_full_size as F
    fresh = F.cases()
    with open(F.OUT) as f:
        assert json.load(f) == json.loads(json.dumps(fresh))


def test_unconfined_process_cannot_call_reference_code():
    """pe_call refuses (exit 76) before pe_lockdown: no instruction of the reference runs in an unconfined process"""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import ref_run as R, ctypes as C; L = R._lib(); L.pe_prepare(); d = R.image_bytes('core'); "
            "i = L.pe_map(d, len(d), 0, b'core'); L.pe_bind(i, 0, None, None); a = (C.c_uint64 * 8)(); L.pe_call(L.pe_entry(i), a)") % os.path.join(ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ))
    assert r.returncode == 76 and "refusing" in r.stderr, (r.returncode, r.stderr[-300:])


def test_confined_process_cannot_open_or_connect():
    """what the filter leaves: after the lockdown the job process cannot open a file, make a socket or start a program"""
    out = R.run("confinement_probe")
    assert int(out["open_errno"]) == 1 and int(out["socket_errno"]) == 1 and int(out["exec_errno"]) == 1 and int(out["uid"]) == 65534


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_remap_runs_equal_the_oracle(seed):
    rng = np.random.default_rng(seed)
    sh, sw = int(rng.integers(20, 90)), int(rng.integers(20, 120))
    cn = (3, 1)[seed % 2]
    src = rng.integers(0, 256, (sh, sw) + ((3,) if cn == 3 else ())).astype(np.uint8)
    dh, dw = int(rng.integers(10, 80)), int(rng.integers(10, 100))
    xm = rng.uniform(-sw * 0.6, sw * 1.6, (dh, dw)).astype(np.float32)          # far outside: several reflections deep
    ym = rng.uniform(-sh * 0.6, sh * 1.6, (dh, dw)).astype(np.float32)
    xy, fxy = R.fixed_maps(xm, ym)
    for border in (2, 1, 4):
        out = R.run("remap_bilinear", src=src, xy=xy, fxy=fxy, border=np.array(border))["out"]
        assert np.array_equal(out, O.remap(src, xm, ym, 1, border)), border


@pytest.mark.parametrize("seed", [1, 2])
def test_pyramid_runs_equal_the_oracle(seed):
    rng = np.random.default_rng(100 + seed)
    h, w = int(rng.integers(2, 60)), int(rng.integers(2, 126))
    s16 = rng.integers(-32768, 32767, (h, w, 3)).astype(np.int16)                # full range: the casts saturate nowhere, they wrap nowhere
    assert np.array_equal(R.run("pyr_down", src=s16)["out"], O.pyr_down_16s(s16))
    hu, wu = (h + 1) // 2, min((w + 1) // 2, 54)             # (pyrUp's row buffers stay on cv::AutoBuffer's stack up to 111 destination pixels)
    u16 = rng.integers(-8000, 8000, (hu, wu, 3)).astype(np.int16)
    assert np.array_equal(R.run("pyr_up", src=u16, dsize=np.array((2 * wu, 2 * hu)))["out"], O.pyr_up_16s(u16))
    f = rng.uniform(0, 1, (h, w)).astype(np.float32)
    assert np.array_equal(R.run("pyr_down", src=f)["out"].view(np.uint32), O.pyr_down_32f(f).view(np.uint32))


def test_float_pyrdown_scalar_loops_alone_differ_from_the_sse_bodies():
    """PE_RUN_NO_SSE=1 answers cv::checkHardwareSupport with "no": the reference's scalar loops then process every column, and the last
    bits move -- the association the audit read off the SSE body is real, and the oracle follows the SSE build."""
    rng = np.random.default_rng(7)
    f = rng.uniform(0, 1, (40, 64)).astype(np.float32)
    sse = R.run("pyr_down", src=f)["out"]
    os.environ["PE_RUN_NO_SSE"] = "1"
    try:
        scalar = R.run("pyr_down", src=f)["out"]
    finally:
        del os.environ["PE_RUN_NO_SSE"]
    assert np.array_equal(sse.view(np.uint32), O.pyr_down_32f(f).view(np.uint32))
    assert np.abs(sse - scalar).max() < 1e-6 and not np.array_equal(sse.view(np.uint32), scalar.view(np.uint32))


def test_normalise_and_conversion_runs_equal_the_oracle():
    rng = np.random.default_rng(11)
    w = rng.uniform(0, 6, (29, 41)).astype(np.float32)
    w[rng.uniform(size=w.shape) < 0.1] = 0
    v = rng.integers(-32768, 32767, (29, 41, 3)).astype(np.int16)
    exp = v.copy()
    O.lib().mo_normalize_using_weight_map_32f(w.ctypes.data_as(C.c_void_p), exp.ctypes.data_as(C.c_void_p), 41, 29)
    assert np.array_equal(R.run("normalize_32f", weight=w, src=v)["out"], exp)
    img = rng.integers(0, 256, (17, 23, 3)).astype(np.uint8)
    for gv in rng.uniform(0.3, 3.0, 6).astype(np.float32):
        out = R.run("cvt_scale", src=img, scale=np.array(gv), shift=np.array(0.0), to_float=np.array(0))["out"]
        assert np.array_equal(out, O.gain_apply(img.copy(), float(gv))), float(gv)


def test_border_interpolate_runs_equal_the_oracle():
    """cv::borderInterpolate (an export): REPLICATE, REFLECT, WRAP and REFLECT_101 for lengths 1 .. 8 and positions several lengths outside"""
    ps, ls, bs = [], [], []
    for b in (1, 2, 3, 4):
        for n in range(1, 9):
            for p in range(-3 * n - 2, 4 * n + 3):
                ps.append(p); ls.append(n); bs.append(b)
    ps, ls, bs = (np.array(a, np.int32) for a in (ps, ls, bs))
    out = R.run("border_interpolate", p=ps, length=ls, border=bs)["out"]
    exp = np.array([O.lib().mo_border_interpolate(int(a), int(b), int(c)) for a, b, c in zip(ps, ls, bs)], np.int32)
    assert np.array_equal(out, exp)


def test_distance_transform_and_nv12_runs_equal_the_oracle():
    rng = np.random.default_rng(21)
    for p in (0.01, 0.3, 2.0):
        m = (rng.uniform(size=(33, 47)) > p).astype(np.uint8) * 255
        assert np.array_equal(R.run("distance_l1", mask=m)["out"].view(np.uint32), O.distance_transform_l1(m).view(np.uint32)), p
    yuv = rng.integers(0, 256, (30, 36)).astype(np.uint8)
    assert np.array_equal(R.run("nv12", yuv=yuv, fn=np.array(R.NV12_TO_BGR))["out"], O.cvt_nv12_bgr(yuv))


@pytest.mark.parametrize("shape", [(1, 1), (1, 5), (5, 1), (2, 2)])
def test_remap_runs_on_degenerate_sources(shape):
    """one-pixel and one-row sources under maps seven pixels outside: every tap is a reflection of a reflection"""
    rng = np.random.default_rng(shape[0] * 10 + shape[1])
    src = rng.integers(0, 256, shape + (3,)).astype(np.uint8)
    xm = rng.uniform(-7, shape[1] + 7, (11, 13)).astype(np.float32)
    ym = rng.uniform(-7, shape[0] + 7, (11, 13)).astype(np.float32)
    xy, fxy = R.fixed_maps(xm, ym)
    for border in (2, 1, 4):
        assert np.array_equal(R.run("remap_bilinear", src=src, xy=xy, fxy=fxy, border=np.array(border))["out"], O.remap(src, xm, ym, 1, border)), border


def test_linear_resize_runs_equal_the_oracle():
    rng = np.random.default_rng(31)
    for _ in range(6):
        sh, sw = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        dw, dh = int(rng.integers(1, 400)), int(rng.integers(1, 200))
        m = rng.uniform(0.5, 2.0, (sh, sw)).astype(np.float32)
        assert np.array_equal(R.run("resize_linear", src=m, dsize=np.array((dw, dh)))["out"].view(np.uint32),
                              O.resize_linear_32f(m, dw, dh).view(np.uint32)), (sw, sh, dw, dh)
        u = rng.integers(0, 256, (sh, sw)).astype(np.uint8)
        assert np.array_equal(R.run("resize_linear", src=u, dsize=np.array((dw, dh)))["out"], O.resize_linear_8u(u, dw, dh)), (sw, sh, dw, dh)


def test_constant_border_remap_with_no_tap_inside():
    rng = np.random.default_rng(33)
    src = rng.integers(0, 256, (30, 41, 3)).astype(np.uint8)
    xm = rng.uniform(-20, 61, (25, 37)).astype(np.float32)
    ym = rng.uniform(-20, 50, (25, 37)).astype(np.float32)
    xy, fxy = R.fixed_maps(xm, ym)
    out = R.run("remap_bilinear", src=src, xy=xy, fxy=fxy, border=np.array(0))["out"]
    assert np.array_equal(out, O.remap(src, xm, ym, 1, 0)) and np.array_equal(out, O.remap_fixed(src, xy, fxy))


def test_blocks_gain_apply_of_the_exe_equals_the_oracle():
    """the reference's OWN compiled per-frame gain loop (BlocksGainCompensator::apply inside OpenCV2.4.11-Stitching.exe, found through its PDB)"""
    rng = np.random.default_rng(41)
    for h, w in ((31, 47), (5, 3), (64, 96)):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        gm = rng.uniform(0.3, 3.0, (h, w)).astype(np.float32)
        assert np.array_equal(R.run("blocks_gain_apply", image=img, gain_map=gm)["out"], O.blocks_gain_apply(img.copy(), gm)), (h, w)


def test_saturating_16_bit_add_and_subtract():
    """cv::subtract (the Laplacian levels) and cv::add (their collapse) on CV_16S SATURATE, in the scalar loop and in the SSE2 one alike"""
    rng = np.random.default_rng(43)
    a = rng.integers(-32768, 32767, (11, 29, 3)).astype(np.int16)
    b = rng.integers(-32768, 32767, (11, 29, 3)).astype(np.int16)
    for sub in (1, 0):
        exp = np.clip(a.astype(np.int32) - b if sub else a.astype(np.int32) + b, -32768, 32767).astype(np.int16)
        for sse in (0, 1):
            assert np.array_equal(R.run("arith16s", a=a, b=b, sub=np.array(sub), sse=np.array(sse))["out"], exp), (sub, sse)
