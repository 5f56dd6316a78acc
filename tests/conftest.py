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
    return a.dtype == b.dtype and len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


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
