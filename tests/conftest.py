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


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    """The first `import torch` on a fresh box pages in ~1.3 GB of shared objects (libtorch_hip, librccl, ...) and can
    take minutes; round 1 met that cost in the middle of the session, inside the first ring test, and read it as a GPU
    hang (the library's streams play no part, tools/hang_probe.py, profiles/r02_a_hang_probe.log).
    Pay it here, where the log says what it is, whenever a GPU test is going to run -- by marker expression or by file
    (`pytest tests/test_gpu_world.py`).  The library itself needs no import order to WORK (the `gpu` fixture does not touch
    torch), but a process that first runs a collective through the library's RCCL and only then imports torch aborts at
    interpreter exit ("double free or corruption", after every test has passed: the two runtimes' exit handlers run in
    the wrong order; tools/exit_abort_bisect.sh), so the tests that count devices with torch must not be
    the ones that import it."""
    if any(it.get_closest_marker("gpu") is not None for it in items):
        import time
        t0 = time.time()
        import torch  # noqa: F401
        sys.stderr.write("[conftest] import torch took %.1f s\n" % (time.time() - t0))


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
