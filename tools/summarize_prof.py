#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into a small text table.
Usage: tools/summarize_prof.py gpurun_out/prof_<tag> [> profiles/<name>.txt]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    import re
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name[:40]


cmd = "python3 bench.py (see tools/profile_gpu.sh)"
try:
    cmd = open(os.path.join(root, "command.txt")).read().strip()
except OSError:
    pass
print("# rocprofv3 --kernel-trace --stats (+ separate --pmc passes) of: %s" % cmd)
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("\n## kernel stats (%s)" % os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    print("%-28s %8s %12s %12s %12s %8s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
    for r in rows:
        if "k_" not in r["Name"]:
            continue
        print("%-28s %8s %12.2f %12.2f %12.2f %8s" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                      float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
counters = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
if counters:
    print("\n## PMC counters, mean per launch")
    names = sorted({c for k in counters.values() for c in k})
    for k, d in sorted(counters.items()):
        if not k.startswith("k_"):
            continue
        print("\n%s" % k)
        for c in names:
            if c in d:
                v = sum(d[c]) / len(d[c])
                extra = ""
                if c == "FETCH_SIZE":
                    extra = "  KB -> x2 (gfx950 wide-read correction) = %.1f MB" % (v * 2 / 1024)
                if c == "WRITE_SIZE":
                    extra = "  KB = %.1f MB" % (v / 1024)
                print("  %-26s %16.1f%s" % (c, v, extra))


# ---- HBM-side bytes per frame set (profiles/traffic.json) ---------------------------------------------------
# Usage: summarize_prof.py <dir> <traffic.json>: the entry "<workload>/<frame sets per launch>" of that file is
# (re)written from this run; workload and batch come from the command line recorded in <dir>/command.txt, the per-frame
# kernels and their launches per sequence from the run itself (every kernel launched at least as often as the least
# launched per-frame kernel; calibration kernels run once and drop out).
if len(sys.argv) > 2 and counters:
    import hashlib
    import json
    import re
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for rel in ["stitchingvideo_amd/csrc/kernels_mb.hip", "stitchingvideo_amd/csrc/kernels.hip",
                "stitchingvideo_amd/csrc/kernels_compose.hip", "stitchingvideo_amd/csrc/kernels_common.hpp",
                "stitchingvideo_amd/csrc/device_types.hpp"]:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    sha = h.hexdigest()[:16]
    m = re.search(r"--batch\s+(\d+)", cmd)
    batch = int(m.group(1)) if m else 8
    m = re.search(r"--config\s+(\S+)", cmd)
    workload = m.group(1) if m else "C3"
    m = re.search(r"bench_paths\.py\s+(\S+)", cmd)
    if m:
        workload = m.group(1)
    if "--undistort" in cmd:
        workload += "+undistort"
    # per-frame kernels of every path (calibration kernels -- tables, weight pyramids -- run per mst_update and drop out);
    # launches per sequence come from the run itself
    per_frame = {"k_warp_pyr", "k_warp_pyr_lds", "k_pyr_chain", "k_pyrdown_stream", "k_blend_coarse", "k_blend_mid", "k_blend_fine",
                 "k_compose", "k_compose_lds", "k_compose_general", "k_remap_fixed", "k_nv12_to_bgr"}
    frame_kernels = {k: d for k, d in counters.items()
                     if k.split("<")[0] in per_frame and "FETCH_SIZE" in d and "WRITE_SIZE" in d}
    if frame_kernels:
        # a sequence launches most kernels once; short side lists (gathered / general tiles) ride along at the same count
        counts = sorted(len(d["FETCH_SIZE"]) for d in frame_kernels.values())
        n_seq = counts[0]
        per = {}
        for k, d in frame_kernels.items():
            f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 2 * 1024      # gfx950: FETCH_SIZE counts half of a wide read
            w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024
            launches = round(len(d["FETCH_SIZE"]) / n_seq)
            per[k] = {"bytes_per_frame": int((f + w) * launches / batch), "launches_per_sequence": launches}
            # instruction-issue side (round 5): wave-instructions and LDS-busy cycles per frame set, where the passes collected them
            for c, key in (("SQ_INSTS_VALU", "valu_winst_per_frame"), ("SQ_INSTS_LDS", "lds_winst_per_frame"), ("SQ_INSTS_SALU", "salu_winst_per_frame"),
                           ("SQ_INSTS_VMEM_RD", "vmem_rd_winst_per_frame"), ("SQ_INSTS_VMEM_WR", "vmem_wr_winst_per_frame"),
                           ("SQ_LDS_IDX_ACTIVE", "lds_active_cycles_per_frame"), ("SQ_LDS_BANK_CONFLICT", "lds_conflict_cycles_per_frame"),
                           ("SQ_WAVES", "waves_per_frame")):
                if c in d:
                    per[k][key] = int(sum(d[c]) / len(d[c]) * launches / batch)
        entry = {"bytes_per_frame": sum(v["bytes_per_frame"] for v in per.values()), "frame_sets_per_launch": batch,
                 "per_kernel": per, "source": "tools/summarize_prof.py (%s)" % os.path.basename(root.rstrip("/"))}
        for key in ("valu_winst_per_frame", "lds_winst_per_frame", "salu_winst_per_frame", "vmem_rd_winst_per_frame", "vmem_wr_winst_per_frame",
                    "lds_active_cycles_per_frame", "lds_conflict_cycles_per_frame"):
            if all(key in v for v in per.values()):
                entry[key] = sum(v[key] for v in per.values())
        try:
            out = json.load(open(sys.argv[2]))
            if out.get("kernel_source_sha16") != sha:
                out = {}
        except (OSError, ValueError):
            out = {}
        out.setdefault("_comment", "HBM-side bytes per frame set from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate "
                                   "runs with --kernel-trace only; FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; "
                                   "Infinity-Cache hits are counted too), per launch sequence divided by its frame sets.  Keys: "
                                   "<workload>/<frame sets per launch>.")
        out["kernel_source_sha16"] = sha
        out.setdefault("entries", {})["%s/%d" % (workload, batch)] = entry
        json.dump(out, open(sys.argv[2], "w"), indent=1)
        print("\n# traffic per frame set (%s, %d frame sets per launch): %.1f MB -> %s" % (workload, batch, entry["bytes_per_frame"] / 1e6, sys.argv[2]))
