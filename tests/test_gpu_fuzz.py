"""A fixed-seed slice of each fuzzer under tools/ inside the gpu suite: random meshes / scenes / instance
clouds / rays through the C ABI against the oracle, byte for byte.  The campaigns with fresh seeds (`python tools/fuzz_*.py
--seed k`, tens of thousands of cases, profiles/r0N_fuzz.log) are the builder's; these slices are what every run of the suite sees.
Budget: about a minute altogether."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "tools") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tools"))


def _log(lines):
    return lambda m: lines.append(m)


def test_fuzz_slice_blas(ctx, oracle):
    import fuzz_blas
    out = []
    bad, degenerate = fuzz_blas.run(100, seed=606, ctx=ctx, log=_log(out), max_tris=5000)
    assert bad == 0, "\n".join(out[:10])
    assert degenerate < 100          # most cases build a tree on both sides


def test_fuzz_slice_cull(ctx, oracle):
    import fuzz_cull
    out = []
    bad, far_cases = fuzz_cull.run(50, seed=606, ctx=ctx, log=_log(out))
    assert bad == 0, "\n".join(out[:10])
    assert far_cases > 0             # the finite far plane decided at least one case: is_visible's third return is taken


def test_fuzz_slice_tlas(ctx, oracle):
    import fuzz_tlas
    out = []
    assert fuzz_tlas.run(40, seed=606, ctx=ctx, log=_log(out), max_n=1500) == 0, "\n".join(out[:10])


def test_fuzz_slice_trace(ctx, oracle):
    import fuzz_trace
    out = []
    bad, _overflows = fuzz_trace.run(20, seed=606, ctx=ctx, log=_log(out), sides=(48, 160, 400))
    assert bad == 0, "\n".join(out[:10])
