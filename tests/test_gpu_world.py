"""The native multi-GPU host (csrc/mvr_world.cpp) on the one GPU a test box has: a world of ONE rank goes through the
whole RCCL path (communicator, ncclAllReduce of the edge table on the library's stream) and must reproduce the
single-context loop bit for bit.  More ranks: the partition itself is covered on the CPU (tests/test_ring_dist.py,
tests/test_host.py::test_ring_segments...) and by the fake worlds of tests/test_gpu_ring.py; the 8-GPU run is the
driver's."""
import importlib

import numpy as np
import pytest

from conftest import PKG, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(mvr):
    g = load_golden("ring_12x2048.npz")
    return list(g["scans"]), [p.copy() for p in g["poses0"]], np.array(mvr.synth_params(12, 3).pivot)


def reference_run(mvr, scans, poses0, origin, steps):
    V = len(scans)
    with mvr.Context(0) as ctx:
        for v in range(V):
            ctx.upload(V + v, scans[v])
        return ctx.ring_step(list(range(V)), [V + v for v in range(V)], [(i, (i + 1) % V) for i in range(V)], poses0, 8.0, origin, steps=steps)


def test_world_of_one_equals_single_context(mvr, scene):
    scans, poses0, origin = scene
    V = len(scans)
    edges = [(i, (i + 1) % V) for i in range(V)]
    ref, rinfo = reference_run(mvr, scans, poses0, origin, 3)
    assert "rccl" in mvr.rccl_library().lower()
    with mvr.World(1) as w:
        for v in range(V):
            w.upload(V + v, scans[v])
        new, info = w.ring_run(list(range(V)), [V + v for v in range(V)], edges, poses0, 8.0, origin, steps=3)
        assert w.ctx(0).comm_info() == (0, 1, 1)           # rank 0 of 1, and RCCL agrees
    assert np.array_equal(new, ref) and np.array_equal(info["rows"], rinfo["rows"])
    assert info["pair_n"] == rinfo["pair_n"] and info["lum_iterations"] == rinfo["lum_iterations"]


def test_comm_init_rank_path_equals_single_context(mvr, scene):
    """one process per GPU: unique id -> ncclCommInitRank -> mvr_ring_run_sharded (the path bench.py takes for N > 1)"""
    scans, poses0, origin = scene
    V = len(scans)
    edges = [(i, (i + 1) % V) for i in range(V)]
    ref, rinfo = reference_run(mvr, scans, poses0, origin, 2)
    uid = mvr.comm_unique_id()
    assert len(uid) == 128
    with mvr.Context(0) as ctx:
        assert ctx.comm_info() == (0, 1, 0)                 # no communicator yet
        for v in range(V):
            ctx.upload(V + v, scans[v])
        plain, _ = ctx.ring_run_sharded(list(range(V)), [V + v for v in range(V)], edges, poses0, 8.0, origin, steps=2)
        ctx.comm_init(uid, 0, 1)
        assert ctx.comm_info() == (0, 1, 1)
        with pytest.raises(mvr.MvrError):
            ctx.comm_init(uid, 0, 1)                        # one communicator per context
        new, info = ctx.ring_run_sharded(list(range(V)), [V + v for v in range(V)], edges, poses0, 8.0, origin, steps=2)
        ctx.comm_destroy()
        assert ctx.comm_info() == (0, 1, 0)
    assert np.array_equal(plain, ref) and np.array_equal(new, ref) and np.array_equal(info["rows"], rinfo["rows"])


def test_world_refuses_devices_it_does_not_have(mvr):
    import torch
    have = torch.cuda.device_count()
    with pytest.raises(mvr.MvrError) as e:
        mvr.World(have + 1)
    assert e.value.status == mvr.E_ARG
    with pytest.raises(mvr.MvrError):
        mvr.World(2, device_ids=[0, 0])                     # one rank per GPU


def test_ring_lum_native_comm_wrapper(mvr, scene):
    """ring.RingLUM(native_comm=True) -- what bench.py drives under torch.distributed.run -- on a world of one"""
    ring = importlib.import_module(PKG + ".ring")
    scans, poses0, origin = scene
    V = len(scans)
    ref, rinfo = reference_run(mvr, scans, poses0, origin, 3)
    be = ring.HipBackend(scans, device=0)
    try:
        be.ctx.comm_init(mvr.comm_unique_id(), 0, 1)
        r = ring.RingLUM(be, V, [len(s) for s in scans], 8.0, origin, rank=0, world=1, native_comm=True)
        new = r.run([p.copy() for p in poses0], 3)
        assert np.array_equal(np.asarray(new), ref) and r.last["n_corr"] == sum(rinfo["pair_n"])
    finally:
        be.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_rows_of_all_ranks_sum_to_the_pass(mvr, scene, world):
    """What N ranks would all-reduce, computed rank by rank on the one GPU here (mvr_ring_rows_sharded -- the very code
    mvr_ring_run_sharded runs before its ncclAllReduce): the tables add up to the unsharded pass (counts exactly, sums
    to rounding), only the rows of a rank's own edges are non-zero, and the host step fed with the sum lands on the
    poses of the single-context step."""
    scans, poses0, origin = scene
    V = len(scans)
    edges = [(i, (i + 1) % V) for i in range(V)]
    posed, raw = list(range(V)), [V + v for v in range(V)]
    with mvr.Context(0) as ctx:
        for v in range(V):
            ctx.upload(V + v, scans[v])
        ref, rinfo = ctx.ring_step(posed, raw, edges, poses0, 8.0, origin)
        total = np.zeros((V, 32))
        for rank in range(world):
            rows = ctx.ring_rows_sharded(rank, world, posed, raw, edges, poses0, 8.0, origin)
            mine = {e for e, _, _ in mvr.ring_segments([len(scans[s]) for s, _ in edges], world, rank)}
            assert all((rows[e, 0] > 0) == (e in mine) or rows[e, 0] == 0 for e in range(V))
            assert all(np.all(rows[e] == 0) for e in range(V) if e not in mine)
            total += rows
        assert np.array_equal(total[:, 0], rinfo["rows"][:, 0])
        assert np.allclose(total[:, 4:], rinfo["rows"][:, 4:], rtol=1e-12, atol=1e-7)
        rc, new, info = mvr.ring_host_step(V, edges, total, origin, poses0)
        assert rc == 0 and np.abs(np.asarray(new) - ref).max() < 1e-9
