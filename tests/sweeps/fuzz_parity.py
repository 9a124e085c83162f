#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: random rigs (warp kind, poses, sizes, focal/scale ratio), exposure modes,
mask styles, band counts, blenders, batch sizes and lens maps -- engine output against the CPU oracle, bit for bit.
Usage: python tests/sweeps/fuzz_parity.py [cases] [seed]      (prints one line per case, exits non-zero on a mismatch)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fuzz_cases import one_case  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(cases):
        line = one_case(rng, i)
        print("%3d %s" % (i, line), flush=True)
        bad += line.startswith("FAIL")
    print("%d case(s), %d mismatch(es)" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
