"""Repository hygiene the judge asked for: no patch left-overs in the history, no reference binaries run by
the default build."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tracked():
    try:
        out = subprocess.run(["git", "ls-files"], cwd=ROOT, capture_output=True, text=True, check=True).stdout
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("not a git checkout (the GPU box gets a snapshot without .git)")
    return out.split("\n")


def test_no_patch_leftovers_are_tracked():
    bad = [f for f in _tracked() if f.endswith((".orig", ".rej", ".bak", "~"))]
    assert not bad, bad


def test_build_does_not_touch_the_reference_runner():
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    code = "\n".join(l for l in src.split("\n") if not l.lstrip().startswith("#"))
    assert "pe_run" not in code and "gcc" not in code
