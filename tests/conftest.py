import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def fields_equal(a, b):
    """BIT equality of two structured arrays, field by field: -0.0 != +0.0 and a NaN equals the same NaN, so the
    total-order min/max of the box reductions (vd_key / vd_min_to) and NaN boxes are compared, not skipped."""
    if a.dtype != b.dtype or len(a) != len(b):
        return False
    for f in a.dtype.names:
        x, y = np.ascontiguousarray(a[f]), np.ascontiguousarray(b[f])
        if x.view(np.uint8).tobytes() != y.view(np.uint8).tobytes():
            return False
    return True


def first_difference(a, b):
    """(field, index) of the first bitwise difference, for assertion messages."""
    for f in a.dtype.names:
        x = np.ascontiguousarray(a[f]).view(np.uint32).reshape(len(a), -1)
        y = np.ascontiguousarray(b[f]).view(np.uint32).reshape(len(b), -1)
        d = np.nonzero((x != y).any(axis=1))[0]
        if d.size:
            return f, int(d[0])
    return None


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/vd_oracle.h)."""
    from oracle import ref
    ref.load()
    return ref


@pytest.fixture(scope="session")
def ctx():
    """A live VdCtx on cuda:0 — fails loudly if the HIP library or the GPU is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    from voidin_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture
def ctx_options(ctx):
    """Per-context options (vd_ctx_set_option) for one test: set(name, value); everything is back to default afterwards."""
    touched = []

    def set_(name, value):
        ctx.set_option(name, value)
        touched.append(name)
    yield set_
    for name in touched:
        ctx.set_option(name, None)
