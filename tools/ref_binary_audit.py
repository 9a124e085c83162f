"""What the reference's OWN binaries compute, read off their machine code (part one of oracle/REFERENCE_BINARY_AUDIT.md; part two RUNS the
same binaries: oracle/ref_run/pe_run.c, tests/ref_run.py).  Besides opencv_imgproc2411.dll it reads opencv_core2411.dll,
opencv_stitching2411.dll, the x64 exe and the 32-bit build of the stock sample, each checked against its SHA-256 first.

The reference cannot be built or run in this image (OpenCV 2.4.11 + <windows.h>), and it ships no test vectors -- but it does ship the
x64 Release DLLs it runs (`/root/reference/stitching/OpenCV2.4.11-Stitching-64bit/x64/Release/opencv_imgproc2411.dll`).  Where the oracle
restates an OpenCV primitive whose result depends on the ORDER of floating-point operations (the SSE body of a filter associates a
three-term sum differently from its scalar tail), the published source alone does not settle what the shipped build does: the compiler's
schedule does.  This tool disassembles the DLL (binutils `objdump`), symbolically executes the few SSE instructions of the loop bodies in
question (loads become symbols, addps / addss / subps / subss / mulps / mulss build expression trees, stores print them) and prints the
expression every store writes -- commutative operands sorted, so that `a + b` and `b + a` read the same, association kept.

    python tools/ref_binary_audit.py            -> one line per audited store / constant; exit code 1 if the DLL is not the audited one

`tests/test_reference_binary_audit.py` runs it where /root/reference exists and compares with the forms the oracle implements
(oracle/REFERENCE_BINARY_AUDIT.md explains every block: which function it is, how the operands map to rows and neighbours).
Nothing of the DLL is copied into this repository: only the facts read off it."""
import hashlib
import os
import re
import subprocess
import sys

DLL = "/root/reference/stitching/OpenCV2.4.11-Stitching-64bit/x64/Release/opencv_imgproc2411.dll"
DLL_SHA256 = "5f0902d99da96d26db4bb304f047365eb7e308af53da1f47bd0c0907d1344a0f"      # the audited file; a different one is reported (exit code 1), not trusted

# (name, first address, last address) of straight-line loop bodies inside .text
BLOCKS = [
    ("row121_sse", 0x180061970, 0x1800619b5),        # SymmRowSmallVec_32f::operator(), ksize 3, kx = [1 2 1]: 8 outputs per trip
    ("row121_pairs", 0x18006a0c0, 0x18006a100),      # SymmRowSmallFilter<float,float,...>::operator(): scalar pairs after the SSE body
    ("row_generic_tail", 0x18006a4f2, 0x18006a538),  # ... its generic loop for what is left: kx[0]*S[0] + kx[k]*(S[-j] + S[j])
    ("col121_sse", 0x180065c40, 0x180065c80),        # SymmColumnSmallVec_32f::operator(), ky = [1 2 1]
    ("col121_unrolled", 0x1800801e0, 0x18008026a),   # SymmColumnSmallFilter<Cast<float,float>,...>::operator(): is_1_2_1, four per trip
    ("col121_rest", 0x180080375, 0x18008039b),       # ... and the elements left after the fours
    ("pyrdown32f_sse", 0x180130020, 0x180130090),    # pyrDown_<FltCast<float,8>, PyrDownVec_32f>: vertical pass, 8 outputs per trip
    ("pyrdown32f_tail", 0x1801300d0, 0x180130120),   # ... its scalar tail
    ("rgb2gray_f32", 0x180025880, 0x1800258b2),      # RGB2Gray<float>::operator(): xmm4 / xmm5 / xmm6 = coeffs[0 .. 2] (cb, cg, cr for BGR input)
]
# initUndistortRectifyMap's per-pixel distortion model (double): (name, first, last, register that holds the result)
IMGPROC_REG_BLOCKS = [
    ("undistort_u", 0x1801776d0, 0x1801777bd, "xmm8"),     # u = fx * (x kr + p1 _2xy + p2 (r2 + 2 x2)) + u0
    ("undistort_v", 0x1801776d0, 0x1801777bd, "xmm7"),     # v = fy * (y kr + p1 (r2 + 2 y2) + p2 _2xy) + v0
]
# cv::solve(DECOMP_LU, CV_64F) behind the exposure gains = LUImpl<double> in opencv_core2411.dll: double sums whose ORDER is the result
# (name, first, last, register to print at the end or None)
CORE_BLOCKS = [
    ("lu_row_update", 0x1800a1060, 0x1800a10b7, None),             # A[j][k] += alpha * A[i][k], four per trip (xmm2 = alpha)
    ("lu_back_substitution", 0x1800a13c0, 0x1800a13ff, "xmm2"),    # s -= A[i][k] * b[k], four per trip: strictly one after the other
    ("lu_back_substitution_end", 0x1800a1441, 0x1800a1476, None),  # ... the rest one by one, then b[i] = s * A[i][i] (the stored -d = 1 / pivot)
]

# the exe's own compiled projector (warpers_inl.hpp:253-268, /fp:precise): the float products and sums of SphericalProjector::mapBackward
EXE_BLOCKS = [
    ("spherical_map_backward", 0x14002118a, 0x14002128e, None),
]

# The blender and the gain apply are scalar loops whose results hang on WHICH conversions and operations run, in which order (a true
# division or a reciprocal multiply; truncation or rounding; a float or a double product).  For those the audit prints the chain of
# arithmetic mnemonics between two addresses of one of the reference's other binaries: (name, binary, sha256, first, last).
RELEASE = os.path.dirname(DLL)
STITCHING = ("opencv_stitching2411.dll", "0d9365a326abaa4d8cd3a0e0ed6b80d3848927fff24ff146d0c56b6e3c0d09c3")
IMGPROC = (os.path.basename(DLL), DLL_SHA256)
CORE = ("opencv_core2411.dll", "cfedf4b256c68aa18a89ce1f7f75d4e3fdaa6fd721b3133b63577919be5c1e60")
EXE = ("OpenCV2.4.11-Stitching.exe", "6ea54dc72eefb94e8f2681ec31bac41f731c7e7dca44de49348a4e2aea2f6997")
# the 32-bit build of the stock sample that sits in x64/Release/old-exe-stitch with 32-bit DLLs of its own (PE32, i386; its PDB names
# blenders.cpp: this one DOES hold the project's own compiled blenders) -- x87 code, audited to record where the two builds part
EXE32 = (os.path.join("old-exe-stitch", "OpenCV2.4.11-Stitching.exe"), "ca6114ce2a6b4d15e7f766b9186b59be3166e246130f5bb445c1e20d22c60d51")
CHAINS = [
    ("mb_feed_32f", STITCHING, 0x18000c2a1, 0x18000c2f4),          # MultiBandBlender::feed, CV_32F weights: one pixel (3 channels + the weight sum)
    ("mb_feed_16s", STITCHING, 0x18000c450, 0x18000c48e),          # ... CV_16S weights: (src * w) >> 8
    ("normalize_32f", STITCHING, 0x1800089a0, 0x1800089f4),        # normalizeUsingWeightMap, float weights: one pixel
    ("normalize_16s", STITCHING, 0x180008ba0, 0x180008bce),        # ... CV_16S weights: (v << 8) / (w + 1)
    ("feather_feed", STITCHING, 0x1800093f3, 0x180009442),         # FeatherBlender::feed: one pixel
    ("blocks_gain_apply", EXE, 0x140008170, 0x1400081a3),          # BlocksGainCompensator::apply (the exe's own copy): one channel
    ("cvtscale_8u8u", CORE, 0x18004e080, 0x18004e0b4),             # cvtScale_<uchar,uchar,float>: `image *= gain` of GainCompensator::apply
    ("cvtscale_8u32f", CORE, 0x180052205, 0x18005222a),            # cvtScale_<uchar,float,float>: weight.convertTo(CV_32F, 1./255.), two per trip
    ("win32_mb_feed_32f", EXE32, 0x40536c, 0x4053c1),              # MultiBandBlender::feed of the 32-bit build: fild / fmul / _ftol2_sse -- the product is NOT rounded to float
    ("win32_normalize_32f", EXE32, 0x405e08, 0x405e54),            # normalizeUsingWeightMap of the 32-bit build: w + eps and the quotient at the FPU's precision
    ("invert3x3_det", CORE, 0x1800AF63F, 0x1800AF728),             # cv::invert, 3 x 3 CV_32F: det3 -- every float widened (cvtps2pd) BEFORE its product
    # integer work: order-free, so what is read is the CONSTANTS (rounding offset and shift) and the rounding mode of the float -> fixed step
    ("remap_bilinear_8uc3", IMGPROC, 0x1800ff0ab, 0x1800ff0d7),    # remapBilinear<FixedPtCast<int,uchar,15>, RemapVec_8u, short>, cn == 3: one channel
    ("remap_maps_to_fixed", IMGPROC, 0x1800dcd30, 0x1800dcdb6),    # remap(): CV_32FC1 map pair -> CV_16SC2 + the 5-bit table index, 16 per trip
    ("fixptcast_8", IMGPROC, 0x18012cc30, 0x18012cc36),            # FixPtCast<short, 8>::operator(): the integer pyrDown's (v + 128) >> 8
    ("fixptcast_6", IMGPROC, 0x18012ccb0, 0x18012ccb3),            # FixPtCast<short, 6>::operator(): the integer pyrUp's (v + 32) >> 6
    ("vresize_linear_8u", IMGPROC, 0x1800e8f20, 0x1800e8f48),      # VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>: the mask fix-up's resize, one output
]
ARITH = {"movsx", "movzx", "cvtdq2ps", "cvtsi2ss", "mulss", "addss", "subss", "divss", "cvttss2si", "cvtss2si", "cvtps2pd", "cvtss2sd", "cvtsd2si",
         "cvttsd2si", "imul", "idiv", "inc", "sar", "shl", "mulsd", "addsd", "subsd", "divsd", "rcpss", "rcpps",
         "mulps", "cvtps2dq", "cvttps2dq", "pand", "por", "packssdw", "psllw", "psrad",
         "fild", "fmul", "fmulp", "fadd", "faddp", "fdiv", "fdivp", "fdivr", "fdivrp", "fistp", "fisttp"}

# immediates that must appear as a group in one function: (name, values)
CONSTANT_GROUPS = [
    ("rgb2gray_u8_coeffs", [0x1323, 0x2591, 0x74C]),               # RGB2Gray<uchar>: R2Y 4899, G2Y 9617, B2Y 1868 (>> 14)
    ("yuv420sp_coeffs", [0x129FBE, 0x2049BA]),                     # ITUR_BT_601_CY 1220542, CUB 2116026
]

LINE = re.compile(r"^\s*([0-9a-f]+):\s+(\S+)\s*(.*)$")


def disassemble(path=DLL):
    out = subprocess.run(["objdump", "-d", "-M", "intel", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
    code = {}
    for ln in out.splitlines():
        m = LINE.match(ln)
        if m:
            code[int(m.group(1), 16)] = (m.group(2), m.group(3).split("#")[0].strip())
    return code


def canon(e):
    """expression tree -> string; operands of + and * sorted"""
    if isinstance(e, str):
        return e
    op, a, b = e
    sa, sb = canon(a), canon(b)
    if op in "+*" and sb < sa:
        sa, sb = sb, sa
    return "(%s %s %s)" % (sa, op, sb)


def mem(operand):
    m = re.search(r"\[(.*)\]", operand)
    return "[" + m.group(1).replace(" ", "") + "]"


def run_block(code, first, last, result_reg=None):
    """-> list of (store operand, expression) in program order.  Registers hold expression trees; an xmm register read before it is
    written is a loop-invariant (a broadcast constant: named by the register)."""
    regs, stores = {}, []
    # pointer bumps inside the body move later operands: track `add reg, imm` so that [rcx-0x20] after `add rcx,0x20` names the same cell
    bump = {}

    def norm(operand):
        s = mem(operand)
        for r, d in bump.items():
            # rewrite r+disp by the accumulated bump (only plain `r` occurrences with optional +-hex displacement are shifted as a whole)
            if re.search(r"\b%s\b" % r, s):
                s = s[:-1] + ("%+d]" % d if d else "]")
        return s
    for addr in sorted(a for a in code if first <= a <= last):
        op, args = code[addr]
        parts = [p.strip() for p in args.split(",")] if args else []
        if op in ("movups", "movaps", "movss", "movdqa", "movdqu", "movsd", "movapd", "movupd"):
            dst, src = parts
            if "PTR" in dst:
                stores.append((norm(dst), canon(regs.get(src, src))))
            elif "PTR" in src:
                regs[dst] = norm(src)
            else:
                regs[dst] = regs.get(src, src)
        elif op in ("addps", "addss", "subps", "subss", "mulps", "mulss", "addsd", "subsd", "mulsd", "divsd", "divss"):
            dst, src = parts
            rhs = norm(src) if "PTR" in src else regs.get(src, src)
            regs[dst] = ({"a": "+", "s": "-", "m": "*", "d": "/"}[op[0]], regs.get(dst, dst), rhs)
        elif op == "call" and re.fullmatch(r"0x[0-9a-f]+", args):
            regs["xmm0"] = "f_%s(%s)" % (args[-4:], canon(regs.get("xmm0", "xmm0")))          # a float function of xmm0 (sinf / cosf thunks)
        elif op == "add" and len(parts) == 2 and re.fullmatch(r"0x[0-9a-f]+", parts[1]) and not parts[0].startswith("e"):
            bump[parts[0]] = bump.get(parts[0], 0) + int(parts[1], 16)
        # (everything else -- counters, lea, jumps -- does not touch the audited data flow)
    if result_reg:
        stores.append((result_reg, canon(regs.get(result_reg, result_reg))))
    return stores


def run_chain(code, first, last):
    """-> the arithmetic mnemonics between two addresses, in program order; a 16-bit read-modify-write `add WORD PTR [..], reg` prints as
    add16 and a float store as store32 (they are where a result lands); shifts carry their count, a 32-bit `lea reg, [.. + imm]` prints as lea+imm"""
    out = []
    for addr in sorted(a for a in code if first <= a <= last):
        op, args = code[addr]
        imm = re.search(r",(0x[0-9a-f]+)$", args)
        if op in ("sar", "shl", "psrad", "psllw") and imm:
            out.append("%s%d" % (op, int(imm.group(1), 16)))               # the shift count is the point
        elif op in ARITH:
            out.append(op)
        elif op == "lea" and re.search(r"\+0x[0-9a-f]+\]$", args) and not args.startswith("r"):
            out.append("lea+%d" % int(re.search(r"\+(0x[0-9a-f]+)\]$", args).group(1), 16))   # a 32-bit lea with a displacement: the rounding offset
        elif op == "add" and args.startswith("WORD PTR"):
            out.append("add16")
        elif op == "movss" and args.startswith("DWORD PTR"):
            out.append("store32")
        elif op == "call" and re.fullmatch(r"0x[0-9a-f]+", args):
            out.append("call:%s" % args)                                        # (the 32-bit build truncates through _ftol2_sse)
        elif op == "fstp" and args.startswith("DWORD PTR"):
            out.append("fstp32")
    return out


def find_constant_group(code, values):
    """first address window of 64 instructions whose immediates contain every value"""
    addrs = sorted(code)
    imm = {}
    for i, a in enumerate(addrs):
        for v in values:
            if re.search(r"\b0x%x\b" % v, code[a][1]):
                imm.setdefault(v, []).append(i)
    for i in imm.get(values[0], []):
        if all(any(abs(j - i) < 64 for j in imm.get(v, [])) for v in values[1:]):
            return addrs[i]
    return None


def main():
    if not os.path.exists(DLL):
        print("reference DLL not present: %s" % DLL)
        return 2
    sha = hashlib.sha256(open(DLL, "rb").read()).hexdigest()
    print("dll %s sha256 %s" % (os.path.basename(DLL), sha))
    if sha != DLL_SHA256:
        print("not the audited file (%s): the addresses below would mean nothing" % DLL_SHA256)
        return 1
    code = disassemble()
    for name, first, last in BLOCKS:
        for dst, expr in run_block(code, first, last):
            print("%s  %s = %s" % (name, dst, expr))
    for name, first, last, reg in IMGPROC_REG_BLOCKS:
        print("%s  %s = %s" % (name, reg, run_block(code, first, last, reg)[-1][1]))
    for name, values in CONSTANT_GROUPS:
        at = find_constant_group(code, values)
        print("%s  %s  %s" % (name, ",".join(str(v) for v in values), ("at 0x%x" % at) if at else "NOT FOUND"))
    rc, listings = 0, {}
    core_path = os.path.join(RELEASE, CORE[0])
    if os.path.exists(core_path) and hashlib.sha256(open(core_path, "rb").read()).hexdigest() == CORE[1]:
        listings[CORE[0]] = disassemble(core_path)
        print("binary %s sha256 %s" % CORE)
        for name, first, last, reg in CORE_BLOCKS:
            for dst, expr in run_block(listings[CORE[0]], first, last, reg):
                print("%s  %s = %s" % (name, dst, expr))
    exe_path = os.path.join(RELEASE, EXE[0])
    if os.path.exists(exe_path) and hashlib.sha256(open(exe_path, "rb").read()).hexdigest() == EXE[1]:
        listings[EXE[0]] = disassemble(exe_path)
        print("binary %s sha256 %s" % EXE)
        for name, first, last, reg in EXE_BLOCKS:
            for dst, expr in run_block(listings[EXE[0]], first, last, reg):
                print("%s  %s = %s" % (name, dst, expr))
    for name, (binary, want), first, last in CHAINS:
        path = os.path.join(RELEASE, binary)
        if binary not in listings:
            got = hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None
            print("binary %s sha256 %s" % (binary, got))
            listings[binary] = disassemble(path) if got == want else None
            if got != want:
                print("not the audited file (%s)" % want)
                rc = 1
        if listings[binary] is not None:
            print("%s  chain: %s" % (name, " ".join(run_chain(listings[binary], first, last))))
    return rc


if __name__ == "__main__":
    sys.exit(main())
