#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per dispatch, per kernel, over every pass directory under <dir>."""
import csv, glob, os, re, sys, collections

def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "")
                m = re.search(r"\b(k_[a-z0-9_]+)", name)
                if not m:
                    continue
                short = m.group(1)
                slot = acc[short][row["Counter_Name"]]
                slot[0] += float(row["Counter_Value"]); slot[1] += 1
    for kern in sorted(acc):
        print(kern)
        for cname in sorted(acc[kern]):
            tot, n = acc[kern][cname]
            print(f"  {cname:32s} {tot / max(n, 1):16.1f}   (n={n})")

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
