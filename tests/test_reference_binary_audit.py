"""The oracle's order-sensitive float primitives against the reference's OWN machine code: tools/ref_binary_audit.py disassembles the
x64 Release opencv_imgproc2411.dll the reference ships and prints the expression every store of the audited loops writes
(oracle/REFERENCE_BINARY_AUDIT.md).  Runs where /root/reference exists (this container); skipped elsewhere -- the GPU box has no
reference tree, and nothing of the DLL is copied into the repository."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_binary_audit as A          # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(A.DLL) or shutil.which("objdump") is None,
                                reason="needs the reference tree (/root/reference) and binutils objdump")

SHA = "5f0902d99da96d26db4bb304f047365eb7e308af53da1f47bd0c0907d1344a0f"


@pytest.fixture(scope="module")
def audit():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_binary_audit.py")], capture_output=True, text=True, check=True).stdout
    lines = out.strip().splitlines()
    assert lines[0].endswith(SHA), lines[0]
    got = {}
    for ln in lines[1:]:
        if ln.startswith("binary "):                      # the other audited binaries, each with its digest
            assert ln.split()[-1] == dict([A.STITCHING, A.CORE, A.EXE, A.EXE32, A.IMGPROC])[ln.split()[1]], ln
            continue
        name, rest = ln.split("  ", 1)
        got.setdefault(name, []).append(rest)
    return got


def _shape(expr):
    """the expression with every memory operand / register replaced by a letter per distinct operand, in order of appearance"""
    import re
    names = {}

    def sub(m):
        return names.setdefault(m.group(0), chr(ord("A") + len(names)))
    return re.sub(r"\[[^\]]*\]|xmm\d+", sub, expr.split(" = ", 1)[1])


def test_sobel_row_forms(audit):
    # SSE body: ((m + m) + c) + a -- the doubled operand is added to itself first, the two neighbours one after the other
    assert [_shape(e) for e in audit["row121_sse"]] == ["(((A + A) + B) + C)"] * 2
    # scalar pairs: ((m * 2) + a) + c
    assert [_shape(e) for e in audit["row121_pairs"]] == ["(((A * B) + C) + D)"] * 2
    # generic tail: (S[j] + S[-j]) * kx[k] + kx[0]*S[0]
    assert [_shape(e) for e in audit["row_generic_tail"]] == ["(((A + B) * C) + D)"]


def test_sobel_column_forms(audit):
    # SSE body: (((m + m) + b) + t) + delta
    assert [_shape(e) for e in audit["col121_sse"]] == ["((((A + A) + B) + C) + D)"] * 2
    # every column after it -- the fours AND the rest: (((m * 2) + t) + b) + delta
    assert [_shape(e) for e in audit["col121_unrolled"]] == ["((((A * B) + C) + D) + E)"] * 4
    assert [_shape(e) for e in audit["col121_rest"]] == ["((((A * B) + C) + D) + E)"]
    # ... with the SAME operand roles in both: the doubled row is S1, the first neighbour added is the one at the loop's base pointer
    assert "[r9-0x30" in audit["col121_sse"][0] and "[rcx-0x14" in audit["col121_unrolled"][0] and "[rax-0x4" in audit["col121_rest"][0]


def test_pyrdown_float_forms(audit):
    assert [_shape(e) for e in audit["pyrdown32f_sse"]] == ["(((((A + B) + C) * D) + ((E + F) + (C + C))) * G)"] * 2
    assert [_shape(e) for e in audit["pyrdown32f_tail"]] == ["((((((A + B) * C) * D) + ((E * F) * D)) + (G * D)) + (H * D))"]


def test_gray_float_is_left_to_right(audit):
    # (g * cg + b * cb) + r * cr: the first two products summed first, the third added last -- b*0.114f + g*0.587f + r*0.299f left to right
    e = audit["rgb2gray_f32"][0]
    assert _shape(e) == "(((A * B) + (C * D)) + (E * F))" and e.endswith("+ ([r11+0x8] * xmm6))")


def test_constants_present(audit):
    assert audit["rgb2gray_u8_coeffs"][0].startswith("4899,9617,1868  at 0x")
    assert audit["yuv420sp_coeffs"][0].startswith("1220542,2116026  at 0x")


def test_oracle_sobel_is_the_audited_table():
    """The forms above, evaluated in float32 on values that split them, against oracle/dpseam.c through mo_dp_gradients."""
    import oracle_lib as O
    F = np.float32
    w = 13                                    # W8 = 8: columns 0..7 SSE, 8..11 pairs / fours, 12 tail / rest
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 255, size=(9, w, 3)).astype(np.float32)
    img[..., 0] = 0
    img[..., 2] = 0                           # gray = g * 0.587f: one product, no sum
    gx, gy = O.dp_gradients(img)
    g = (img[..., 1] * F(0.587)).astype(F)
    r = lambda i, n: -i if i < 0 else (2 * n - 2 - i if i >= n else i)       # noqa: E731  (REFLECT_101)
    y = 4
    for x in range(w):
        a = [g[yy, r(x - 1, w)] for yy in (y - 1, y, y + 1)]
        m = [g[yy, x] for yy in (y - 1, y, y + 1)]
        c = [g[yy, r(x + 1, w)] for yy in (y - 1, y, y + 1)]
        d = [F(c[k] - a[k]) for k in range(3)]
        if x < 8:
            s = [F(a[k] + F(F(m[k] + m[k]) + c[k])) for k in range(3)]
            want_gx = F(d[0] + F(d[2] + F(d[1] + d[1])))
        else:
            s = [F(F(a[k] + F(m[k] * F(2))) + c[k]) if x < 12 else F(F(m[k] * F(2)) + F(c[k] + a[k])) for k in range(3)]
            want_gx = F(F(d[0] + F(d[1] * F(2))) + d[2])
        assert gx[y, x] == want_gx, x
        assert gy[y, x] == F(s[2] - s[0]), x


# ---- the blender and the gain apply: which conversions and operations run, in which order ------------------------------------------------

def _chain(audit, name):
    return audit[name][0].split("chain: ", 1)[1].split()


def test_blender_chains(audit):
    px = ["movsx", "cvtdq2ps", "mulss", "cvttss2si", "add16"]            # dst += short(float(src) * w): ONE float product, truncated
    assert _chain(audit, "mb_feed_32f") == px * 3 + ["addss", "store32"]  # ... and dst_weight += w
    assert _chain(audit, "feather_feed") == px * 3 + ["addss"]
    # normalizeUsingWeightMap: v / (w + eps) -- the sum redone and a true division per channel (no reciprocal), truncated
    assert _chain(audit, "normalize_32f") == ["movsx", "cvtdq2ps", "addss", "divss", "cvttss2si"] * 3
    # the CV_16S weight forms: (src * w) >> 8 and (v << 8) / (w + 1)
    assert _chain(audit, "mb_feed_16s")[:15] == ["movsx", "movsx", "imul", "sar8", "add16"] * 3
    assert _chain(audit, "normalize_16s") == ["movsx", "inc", "movsx", "shl8", "idiv", "movsx", "shl8", "idiv", "movsx", "shl8", "idiv"]


def test_gain_apply_chains(audit):
    # saturate_cast<uchar>(uchar * float): a float product, widened to double, rounded to nearest even by cvtsd2si, clamped
    assert _chain(audit, "blocks_gain_apply") == ["movzx", "cvtdq2ps", "mulss", "cvtps2pd", "cvtsd2si"]
    # `image *= gain` -> cvtScale_<uchar,uchar,float>: src * (float)scale + (float)shift, the same rounding
    assert _chain(audit, "cvtscale_8u8u")[:6] == ["movzx", "cvtdq2ps", "mulss", "addss", "cvtps2pd", "cvtsd2si"]
    # weight.convertTo(CV_32F, 1./255.): src * scale + shift in float, two per trip
    want = ["movzx", "cvtdq2ps", "mulss", "addss", "store32"] * 2
    want.remove("movzx")                                   # (the first of the two loads sits above the window)
    assert sorted(_chain(audit, "cvtscale_8u32f")) == sorted(want)


def test_the_32_bit_build_parts_from_the_x64_build(audit):
    """The stock sample's 32-bit build (x64/Release/old-exe-stitch, x87 code, its own compiled blenders.cpp) keeps the weighted product
    and the normalisation's sum and quotient at the FPU's precision before truncating; the x64 build rounds each to float first.  They
    disagree on about one product in 3000 -- the oracle (and the engine) follow the x64 build, SURVEY 8c's choice."""
    px = ["fild", "fmul", "call:0x448600", "add16"]                       # short(x87(src * w)): no float rounding between fmul and _ftol2_sse
    c = [m for m in _chain(audit, "win32_mb_feed_32f") if m not in ("movsx", "inc")]
    assert c == px * 3 + ["fadd", "fstp32"]
    c = [m for m in _chain(audit, "win32_normalize_32f") if m != "movsx"]
    assert c == ["fild", "fadd", "fdivp", "call:0x448600"] * 3
    # a sample on which the two forms differ, and which one the oracle computes
    import ctypes as C
    import oracle_lib as O
    F = np.float32
    v = np.array([[[26003, 15182, 24431]]], np.int16)
    w = np.array([[1.4341636]], F)
    x64 = np.trunc((v.astype(F) / (w + F(1e-5))[..., None]).astype(F)).astype(np.int16)
    x87 = np.trunc(v.astype(np.float64) / (w.astype(np.float64) + np.float64(F(1e-5)))[..., None]).astype(np.int16)
    assert x64[0, 0, 0] == 18131 and x87[0, 0, 0] == 18130
    got = v.copy()
    O.lib().mo_normalize_using_weight_map_32f(w.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), 1, 1)
    assert np.array_equal(got, x64)
    for src, wt, a, b in ((24544, 0.9827656, 24121, 24120), (15182, 0.8517982, 12932, 12931)):
        assert int(np.trunc(F(F(src) * F(wt)))) == a and int(np.trunc(float(src) * float(F(wt)))) == b
        assert O.lib().mo_trunc_s16(C.c_float(float(F(F(src) * F(wt))))) == a


def test_spherical_map_backward_as_compiled_into_the_exe(audit):
    """warpers_inl.hpp:253-268 as the x64 exe holds it (/fp:precise): u / scale and v / scale by true divisions, sinv = sinf(pi - v),
    x_ = sinv * sinf(u), y_ = cosf(pi - v), z_ = sinv * cosf(u), each output (k0 * x_ + k1 * y_) + k2 * z_ in float, then a true division
    by z -- the order oracle/warpers.c and the engine's per-axis tables follow."""
    outs = [e for e in audit["spherical_map_backward"] if not e.startswith("[rsp")]
    assert len(outs) == 4
    x, y, xd, yd = outs
    # the same three products summed the same way for x, y and z; f_ed76 = sinf, f_ed7c = cosf (two thunks, two arguments)
    core = "((((f_ed76((A - (B / C))) * f_ed76((D / C))) * E) + (F * f_ed7c((A - (B / C))))) + ((f_ed76((A - (B / C))) * f_ed7c((D / C))) * G))"
    assert _shape(x) == core and _shape(y) == core
    assert xd.startswith("[rdi] = ([rdi] / (") and yd.startswith("[rax] = ([rax] / (")
    assert xd.split(" / ", 1)[1] == yd.split(" / ", 1)[1]                 # both by the same z
    assert "[rbx+0x70]" in x and "[rbx+0x74]" in x and "[rbx+0x78]" in x and "[rbx+0x88]" in xd and "[rbx+0x90]" in xd   # k_rinv[0..2], [6..8]


def test_invert3x3_determinant_is_in_double(audit):
    """cv::invert of a 3 x 3 CV_32F matrix (K.inv() in ProjectorBase::setCameraParams): 2.4.11's det3 casts its inner products to double, and
    the shipped DLL widens EVERY operand before it multiplies -- nine double products, no float one.  Rounds 1-4 restated the determinant in
    float; oracle/warpers.c and csrc/geometry.cpp were corrected from this reading."""
    c = _chain(audit, "invert3x3_det")
    assert not {"mulss", "subss", "addss"} & set(c)
    assert c.count("mulsd") == 9 and c.count("subsd") == 4 and c.count("addsd") == 1 and c.count("divsd") == 1 and c.count("cvtps2pd") >= 15
    # a matrix on which the two forms give different floats, and which one the oracle returns
    import oracle_lib as O
    import ctypes as C
    K = np.array([[1250.1234, 0.0, 961.77], [0.0, 1249.8765, 539.31], [0.0, 0.0, 1.0]], np.float32)
    rng = np.random.default_rng(5)
    found = False
    for _ in range(2000):
        M = (K + rng.normal(0, 3, (3, 3)).astype(np.float32) * np.array([[1, 1e-3, 1], [1e-3, 1, 1], [1e-6, 1e-6, 1e-3]], np.float32)).astype(np.float32)
        m = M.astype(np.float64)
        det64 = m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0]) + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0])
        F = np.float32
        det32 = F(F(M[0, 0] * F(F(M[1, 1] * M[2, 2]) - F(M[1, 2] * M[2, 1]))) - F(M[0, 1] * F(F(M[1, 0] * M[2, 2]) - F(M[1, 2] * M[2, 0])))) + F(M[0, 2] * F(F(M[1, 0] * M[2, 1]) - F(M[1, 1] * M[2, 0])))
        a = np.float32((m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * (1.0 / det64))
        b = np.float32((m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * (1.0 / float(det32)))
        if a != b:
            out = np.zeros(9, np.float32)
            O.lib().mo_invert3x3_32f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            O.lib().mo_invert3x3_32f(np.ascontiguousarray(M).ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), None)
            assert out[0] == a
            found = True
            break
    assert found


def test_undistort_model_in_source_order(audit):
    """initUndistortRectifyMap's per-pixel model in the shipped imgproc DLL, every operation in double and in the order of the published source:
    w = 1 / _w, x = w _x, y = w _y, r2 = x x + y y, _2xy = (x 2) y, kr = (((k3 r2 + k2) r2 + k1) r2 + 1) / (((k6 r2 + k5) r2 + k4) r2 + 1),
    u = ((kr x + _2xy p1) + (x2 2 + r2) p2) fx + u0, v = ((kr y + (r2 + y2 2) p1) + _2xy p2) fy + v0 -- oracle/undistort.c and the engine's host map
    builder (csrc/geometry.cpp) are written in exactly this order."""
    def simplify(e):
        e = e.split(" = ", 1)[1]
        e = e.replace("((xmm8 / xmm9) * xmm10)", "X").replace("((xmm8 / xmm9) * xmm11)", "Y").replace("((X * X) + (Y * Y))", "R2")
        e = e.replace("((((((R2 * xmm0) + [rsp+0x58]) * R2) + [rsp+0x68]) * R2) + xmm8)", "NUM")
        e = e.replace("((((((R2 * xmm13) + xmm14) * R2) + xmm15) * R2) + xmm8)", "DEN")
        return e.replace("(NUM / DEN)", "KR")
    u, v = simplify(audit["undistort_u"][0]), simplify(audit["undistort_v"][0])
    assert u == "(((((KR * X) + (((X * [rip+0x36d64]) * Y) * [rsp+0x38])) + ((((X * X) * [rip+0x36cfa]) + R2) * [rsp+0x40])) * [rsp+0x48]) + [rsp+0x50])"
    assert v == "(((((KR * Y) + ((R2 + ((Y * Y) * [rip+0x36cd5])) * [rsp+0x38])) + (((X * [rip+0x36d64]) * Y) * [rsp+0x40])) * [rsp+0x60]) + [rsp+0x70])"


def test_lu_solve_sums_in_source_order(audit):
    """cv::solve(DECOMP_LU, CV_64F) behind the exposure gains: the row update is element by element and the back substitution subtracts its
    products strictly one after the other (unrolled by four, not reassociated), then multiplies by the stored reciprocal of the pivot --
    oracle/exposure_feed.c: mo_solve_lu_64f is written in exactly that order."""
    assert [_shape(e) for e in audit["lu_row_update"]] == ["((A * B) + C)"] * 4
    assert all(e.split(" = ")[0] == e.split(" + ")[-1].rstrip(")") for e in audit["lu_row_update"])          # A[j][k] = ... + A[j][k]
    assert _shape(audit["lu_back_substitution"][0]) == "((((A - (B * C)) - (D * E)) - (F * G)) - (H * I))"
    assert _shape(audit["lu_back_substitution_end"][0]) == "((A - (B * C)) * D)"


def test_integer_constants_and_the_map_rounding(audit):
    # remapBilinear, 8-bit 3 channels: four products, + (1 << 14), >> 15
    assert _chain(audit, "remap_bilinear_8uc3") == ["movzx", "imul"] * 4 + ["lea+16384", "sar15"]
    # float maps -> fixed point: x * 32 rounded to NEAREST EVEN (cvtps2dq, not a truncation), & 31 for the table index, >> 5 and a
    # saturating pack for the integer part
    c = _chain(audit, "remap_maps_to_fixed")
    assert c[:8] == ["mulps", "cvtps2dq"] * 4 and "cvttps2dq" not in c
    assert c.count("psrad5") == 4 and c.count("packssdw") == 4 and "psllw5" in c and c.count("pand") == 4
    # the integer pyramids' casts
    assert _chain(audit, "fixptcast_8") == ["lea+128", "sar8"]
    assert _chain(audit, "fixptcast_6") == ["lea+32", "sar6"]
    # the 8-bit vertical resize of the mask fix-up: (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    assert _chain(audit, "vresize_linear_8u") == ["sar4", "sar4", "imul", "sar16", "imul", "sar16", "lea+2", "sar2"]


def test_oracle_blender_arithmetic_is_the_audited_chain():
    """The chains above evaluated in numpy float32 on values chosen to split the alternatives (a reciprocal multiply, a rounded instead of
    a truncated conversion, a double product), against the oracle's own entry points."""
    import ctypes as C
    import oracle_lib as O
    F = np.float32
    rng = np.random.default_rng(11)
    n = 4096
    v = rng.integers(-32768, 32767, size=(1, n, 3)).astype(np.int16)
    w = rng.uniform(0, 6, size=(1, n)).astype(F)
    w[0, :64] = 0                                              # the eps alone divides
    got = v.copy()
    O.lib().mo_normalize_using_weight_map_32f(w.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), n, 1)
    q = (v.astype(F) / (w + F(1e-5))[..., None]).astype(F)      # divss on float32 operands
    want = np.trunc(q).astype(np.int64).astype(np.int32)
    want = np.where(np.abs(q) >= 2147483648.0, np.int32(-2147483648), want).astype(np.int32).astype(np.int16)   # cvttss2si, then the low 16 bits
    assert np.array_equal(got, want)
    recip = np.trunc(v.astype(F) * (F(1) / (w + F(1e-5)))[..., None]).astype(np.int64).astype(np.int32).astype(np.int16)
    assert not np.array_equal(recip, want)                     # the sample does tell a reciprocal multiply from the division

    img = rng.integers(0, 256, size=(8, 64, 3)).astype(np.uint8)
    g = rng.uniform(0.5, 1.9, size=(8, 64)).astype(F)
    got = O.blocks_gain_apply(img.copy(), g)
    p = (img.astype(F) * g[..., None]).astype(F)               # mulss
    want = np.clip(np.rint(p.astype(np.float64)), 0, 255).astype(np.uint8)     # cvtsd2si = nearest even, then the clamp
    assert np.array_equal(got, want)
