import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "multi-view-registration_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) the in-tree native libraries."""
    import __graft_entry__ as g
    g.build_hip()
    return g


@pytest.fixture(scope="session")
def mvr(built):
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session", params=["culled", "culled_w1", "culled_w2", "culled_w4", "brute"])
def gpu(mvr, request):
    """A GPU context; -m gpu tests fail loudly (no skip, no fallback) without one.
    Every parity test runs against both exact search kernels: the spatially
    culled one (default: waves per query set chosen by launch size; also forced to 1, 2 and 4) and the brute-force one."""
    # torch (used by the ring tests for streams / the edge table) initialises its HIP state FIRST, as in bench.py:
    # one full run hung for minutes at the first torch use after dozens of library streams already existed
    import torch
    torch.cuda.init()
    ctx = mvr.Context(0)
    ctx.tune(nn_mode=0 if request.param == "brute" else 1)
    if request.param.startswith("culled_w"):
        ctx.tune(cull_w=int(request.param[-1]))
    ctx.mode = request.param
    yield ctx
    ctx.close()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rand_cloud(rng, n, scale=100.0, centre=(0.0, 0.0, 900.0)):
    p = np.empty((n, 4), np.float32)
    p[:, :3] = (rng.standard_normal((n, 3)) * scale + np.asarray(centre)).astype(np.float32)
    p[:, 3] = 1.0
    return p
