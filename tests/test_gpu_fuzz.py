"""A fixed-seed slice of the randomised parity sweep (tests/fuzz_cases.py) inside the driver-run suite:
200 random rigs / exposure modes / mask styles / band counts / blenders / weight types / batch sizes /
lens maps, engine vs oracle, bit for bit.  tests/sweeps/fuzz_parity.py runs the same generator for longer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHUNKS, PER_CHUNK = 8, 25


@pytest.mark.parametrize("chunk", range(CHUNKS))
def test_fuzz_slice_bit_exact(chunk):
    import fuzz_cases
    rng = np.random.default_rng(20261003 + chunk)
    lines = [fuzz_cases.one_case(rng, i) for i in range(PER_CHUNK)]
    bad = [ln for ln in lines if ln.startswith("FAIL")]
    assert not bad, "\n".join(bad)
    assert sum(ln.startswith("ok") for ln in lines) >= PER_CHUNK // 2, "too many degenerate cases:\n" + "\n".join(lines)


def test_fuzz_slice_multi_tile_frames(monkeypatch):
    """The same generator with frames up to 1400 px wide: several warp / chain / canvas tiles per camera."""
    import fuzz_cases
    monkeypatch.setenv("FUZZ_MAX", "1400")
    rng = np.random.default_rng(77)
    lines = [fuzz_cases.one_case(rng, i) for i in range(12)]
    bad = [ln for ln in lines if ln.startswith("FAIL")]
    assert not bad, "\n".join(bad)
