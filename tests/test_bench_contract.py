"""bench.py's contract surface that can be checked without a GPU: the driver's flags parse and default to N = 1, the product path fails
loudly (no CPU fallback, no oracle behind `value`), only the cpu_baseline leg touches oracle/, and profiles/traffic.json has the shape
the roofline object reads."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible: the loud-failure path does not run")
def test_bench_fails_loudly_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "no CPU fallback" in r.stderr
    assert not r.stdout.strip()                          # no JSON line: nothing was measured


def test_driver_flags_and_defaults():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag, default in (("--gpus", "1"), ("--steps", None), ("--warmup", None)):
        m = re.search(r'add_argument\("%s", type=int, default=(\d+)' % flag, src)
        assert m, flag
        if default:
            assert m.group(1) == default
    # the keys the driver reads off the one JSON line
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert key in src, key


def test_only_the_cpu_baseline_leg_reaches_the_oracle():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head, tail = src.split("def cpu_baseline", 1)
    body, rest = tail.split("\ndef ", 1)
    assert "oracle_lib" in body and "oracle_compose" in body
    for part in (head, rest):                            # nothing outside that function loads, imports or calls the oracle
        assert "oracle_lib" not in part and "oracle_compose" not in part and "import oracle" not in part
    # ... and the package the product path imports never does
    pkg = os.path.join(ROOT, "stitchingvideo_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert "oracle_lib" not in text and "libmst_oracle" not in text and "mo_" not in re.sub(r"[a-z]mo_", "", text), fn


def test_traffic_file_shape():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert re.fullmatch(r"[0-9a-f]{16}", t["kernel_source_sha16"])
    e = t["entries"]["C3/8"]
    assert e["frame_sets_per_launch"] == 8 and e["bytes_per_frame"] > 61351920          # never below the algorithmic bytes
    assert abs(sum(k["bytes_per_frame"] for k in e["per_kernel"].values()) - e["bytes_per_frame"]) <= len(e["per_kernel"])
    for k in ("valu_winst_per_frame", "lds_active_cycles_per_frame"):
        assert e[k] > 0
