#!/usr/bin/env python3
"""Static instruction histogram of the multi-band kernels (profiles/r06_floor.md): compiles stitchingvideo_amd/csrc/kernels_mb.hip
to gfx950 assembly with the library's flags and counts, per kernel body, VALU / SALU / LDS / VMEM / control instructions and a few
VALU families.  `python tools/isa_histogram.py [file.hip]`; needs hipcc only (no GPU)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "stitchingvideo_amd", "csrc", "kernels_mb.hip")
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-rdc", "-S", "--cuda-device-only", "-o", asm, src],
                   check=True, cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    text = open(asm).read()
res, cur = {}, None
for line in text.split("\n"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        res[cur] = collections.Counter()
        continue
    if cur is None:
        continue
    if line.startswith(".Lfunc_end"):
        cur = None
        continue
    m = re.match(r"^\t([a-z_0-9]+)", line)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("v_"):
        cls = "valu"
        if op.startswith(("v_div", "v_rcp", "v_fma", "v_mul_f32", "v_add_f32", "v_cvt")):
            res[cur]["float"] += 1
        if "dot" in op:
            res[cur]["dot"] += 1
        if "perm" in op or "alignb" in op:
            res[cur]["perm"] += 1
        if "dpp" in line:
            res[cur]["dpp"] += 1
    elif op.startswith("s_load"):
        cls = "smem"
    elif op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_endpgm")):
        cls = "ctl"
    elif op.startswith("s_"):
        cls = "salu"
    elif op.startswith("ds_"):
        cls = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cls = "vmem"
    else:
        cls = "other"
    res[cur][cls] += 1
print("%-64s %6s %6s %6s %5s %5s %5s | %5s %4s %4s %4s" % ("kernel", "total", "valu", "salu", "lds", "vmem", "ctl", "float", "dot", "perm", "dpp"))
for k, v in res.items():
    tot = sum(v[c] for c in ("valu", "salu", "lds", "vmem", "smem", "ctl"))
    if tot < 300:
        continue
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"mst::\(anonymous namespace\)::", "", name).split("(")[0]
    print("%-64s %6d %6d %6d %5d %5d %5d | %5d %4d %4d %4d" % (name[:64], tot, v["valu"], v["salu"], v["lds"], v["vmem"], v["ctl"], v["float"], v["dot"], v["perm"], v["dpp"]))
