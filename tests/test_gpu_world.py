"""The native multi-GPU host (csrc/mvr_world.cpp) on the one GPU a test box has: a world of ONE rank goes through the
whole RCCL path (communicator, ncclAllReduce of the edge table on the library's stream) and must reproduce the
single-context loop bit for bit.  More ranks: the partition itself is covered on the CPU (tests/test_ring_dist.py,
tests/test_host.py::test_ring_segments...) and by the fake worlds of tests/test_gpu_ring.py; the 8-GPU run is the
driver's."""
import importlib
import os

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


# ---------------------------------------------------------------- the sequential mode's native sharded host
SEQ_V, SEQ_N, SEQ_D = 12, 4000, 6.0


@pytest.fixture(scope="module")
def seq_scene(mvr, orc):
    import ref_driver
    sp = mvr.synth_params(SEQ_V, 5)
    scans = [mvr.synth_view(sp, v, SEQ_N) for v in range(SEQ_V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = ref_driver.init_poses(orc, SEQ_V, piv, ax)
    return sp, scans, poses0


@pytest.mark.parametrize("multi_iter", [False, True])
def test_native_sequential_world_of_one(mvr, orc, seq_scene, multi_iter):
    """mvr_seq_run_sharded (the native host of the target-sharded sequential mode: registrator.cpp:563-577 with
    ncclAllReduce(min) of the keys and ncclAllReduce(sum) of the sums on the library's stream) in a world of ONE: with a real
    RCCL communicator of one rank == without a communicator == the part-by-part Python rehearsal of the same loop over the
    same C entry points, bit for bit; and all of them equal the oracle's unsharded driver (counts, iterations, poses)."""
    import ref_driver
    import torch
    seq = importlib.import_module(PKG + ".seq")
    sp, scans, poses0 = seq_scene
    origin = np.array(sp.pivot)
    kw = dict(max_dist=SEQ_D) if not multi_iter else dict(max_dist=SEQ_D, max_iter=3, feps=-1e300)
    params = mvr.icp_params(**kw)
    ref_poses, ref_log = ref_driver.sequential_icp(orc, scans, poses0, orc.make_params(**kw), fitness_last=False)
    outs = []
    for with_comm in (False, True):
        drv = seq.NativeShardedSequentialICP(scans, device=0, origin=origin)
        try:
            if with_comm:
                drv.comm_init(mvr.comm_unique_id(), 0, 1)
                assert drv.ctx.comm_info() == (0, 1, 1)
            poses, log = drv.run(poses0, params)
            assert drv.shard_size() == SEQ_V * SEQ_N
            outs.append((np.stack(poses), log))
        finally:
            drv.close()
    part = seq.HipPart(scans, device=0, tstream=torch.cuda.Stream(device=0))
    try:
        py_poses, py_log = seq.ShardedSequentialICP([part], SEQ_V, SEQ_N, 1, origin=origin).run(poses0, params)
    finally:
        part.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][0], np.stack(py_poses))
    for a, b, c in zip(outs[0][1], outs[1][1], py_log):
        assert (a["view"], a["n_corr"], a["iterations"], a["mse"]) == (b["view"], b["n_corr"], b["iterations"], b["mse"]) == \
               (c["view"], c["n_corr"], c["iterations"], c["mse"])
        assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["T"], c["T"])
    log = outs[0][1]
    assert [e["n_corr"] for e in log] == [e["n_corr"] for e in ref_log]
    assert [e["iterations"] for e in log] == [e["iterations"] for e in ref_log]
    assert [e["state"].replace("NOT_CONVERGED", "NOT") for e in log] == [e["state"] for e in ref_log]
    for v in range(SEQ_V):
        assert np.abs(outs[0][0][v][:3, :3] - ref_poses[v][:3, :3]).max() < 1e-5, v
        assert np.abs(outs[0][0][v][:3, 3] - ref_poses[v][:3, 3]).max() < 1e-4, v


# ---------------------------------------------------------------- a rank's failure never leaves anybody in a collective
@pytest.mark.timeout(120)
@pytest.mark.parametrize("at", [0, 1, 5])
def test_ring_local_failure_travels_through_the_collective(mvr, scene, at):
    """A rank whose LOCAL work fails in pass `at` (injected) still joins that pass's ncclAllReduce -- with zero rows and a
    raised failure count -- and returns its own status from that very pass; the communicator stays intact and the next run
    on the same context reproduces the single-context loop.  (at = 5 falls into the pipelined stretch of the run.)"""
    scans, poses0, origin = scene
    V = len(scans)
    edges = [(i, (i + 1) % V) for i in range(V)]
    posed, raw = list(range(V)), [V + v for v in range(V)]
    ref, rinfo = reference_run(mvr, scans, poses0, origin, 8)
    with mvr.Context(0) as ctx:
        for v in range(V):
            ctx.upload(V + v, scans[v])
        ctx.comm_init(mvr.comm_unique_id(), 0, 1)
        ctx.tune(inject_fail_pass=at)
        with pytest.raises(mvr.MvrError) as e:
            ctx.ring_run_sharded(posed, raw, edges, poses0, 8.0, origin, steps=8)
        assert e.value.status == mvr.E_HIP and "local work failed" in str(e.value)
        ctx.tune(inject_fail_pass=-1)
        assert ctx.comm_info() == (0, 1, 1)
        new, info = ctx.ring_run_sharded(posed, raw, edges, poses0, 8.0, origin, steps=8)
        assert np.array_equal(new, ref) and np.array_equal(info["rows"], rinfo["rows"])


@pytest.mark.timeout(90)
@pytest.mark.parametrize("at", [0, 5])
def test_ring_peer_that_never_arrives_is_a_timeout_not_a_hang(mvr, scene, at):
    """The stream of this rank stalls in front of pass `at`'s collective (injected: what a peer that died looks like from
    here).  The rank waits `wait_timeout_ms`, aborts its communicator (ncclCommAbort), drains and returns MVR_E_RCCL; the
    multi-GPU entry points refuse until the broken communicator is dropped; the context itself is as good as new."""
    import time
    scans, poses0, origin = scene
    V = len(scans)
    edges = [(i, (i + 1) % V) for i in range(V)]
    posed, raw = list(range(V)), [V + v for v in range(V)]
    ref, rinfo = reference_run(mvr, scans, poses0, origin, 8)
    with mvr.Context(0) as ctx:
        for v in range(V):
            ctx.upload(V + v, scans[v])
        ctx.comm_init(mvr.comm_unique_id(), 0, 1)
        ctx.tune(wait_timeout_ms=400, inject_stall_pass=at)
        t0 = time.time()
        with pytest.raises(mvr.MvrError) as e:
            ctx.ring_run_sharded(posed, raw, edges, poses0, 8.0, origin, steps=8)
        assert e.value.status == mvr.E_RCCL and time.time() - t0 < 20.0
        assert ctx.comm_info()[2] == 0                                   # the communicator is gone
        with pytest.raises(mvr.MvrError) as e2:
            ctx.ring_run_sharded(posed, raw, edges, poses0, 8.0, origin, steps=2)
        assert e2.value.status == mvr.E_RCCL and "aborted" in str(e2.value)
        ctx.comm_destroy()
        new, info = ctx.ring_run_sharded(posed, raw, edges, poses0, 8.0, origin, steps=8)      # a world of its own again
        assert np.array_equal(new, ref) and np.array_equal(info["rows"], rinfo["rows"])
        ctx.comm_init(mvr.comm_unique_id(), 0, 1)                        # ... and a new communicator works
        new, info = ctx.ring_run_sharded(posed, raw, edges, poses0, 8.0, origin, steps=8)
        assert np.array_equal(new, ref)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("how", ["fail", "stall"])
def test_sequential_sharded_failures(mvr, seq_scene, how):
    import time
    seq = importlib.import_module(PKG + ".seq")
    sp, scans, poses0 = seq_scene
    params = mvr.icp_params(max_dist=SEQ_D)
    drv = seq.NativeShardedSequentialICP(scans, device=0, origin=np.array(sp.pivot))
    try:
        good, glog = drv.run(poses0, params)
        drv.comm_init(mvr.comm_unique_id(), 0, 1)
        if how == "fail":
            drv.ctx.tune(inject_fail_pass=3)
        else:
            drv.ctx.tune(wait_timeout_ms=400, inject_stall_pass=3)
        t0 = time.time()
        with pytest.raises(mvr.MvrError) as e:
            drv.run(poses0, params)
        assert time.time() - t0 < 20.0
        assert e.value.status == (mvr.E_HIP if how == "fail" else mvr.E_RCCL)
        drv.ctx.tune(inject_fail_pass=-1, inject_stall_pass=-1)
        if how == "stall":
            with pytest.raises(mvr.MvrError):
                drv.run(poses0, params)                                  # the aborted communicator
            drv.ctx.comm_destroy()
        again, alog = drv.run(poses0, params)
        assert np.array_equal(np.stack(again), np.stack(good)) and [e["n_corr"] for e in alog] == [e["n_corr"] for e in glog]
    finally:
        drv.close()


# ---------------------------------------------------------------- more than one GPU (skipped on the one-GPU test box)
def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.timeout(300)
def test_world_of_two_devices(mvr, scene):
    """mvr_world_create(2): one process, two GPUs, ncclCommInitAll; every rank runs its half of the queries, the edge table is
    all-reduced over xGMI, the ranks end with bit-identical poses (mvr_world_ring_run checks it, and so does the pose hash
    inside mvr_ring_run_sharded) that agree with the single-context run to rounding (the sums are added in another order)."""
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    scans, poses0, origin = scene
    V = len(scans)
    edges = [(i, (i + 1) % V) for i in range(V)]
    ref, rinfo = reference_run(mvr, scans, poses0, origin, 6)
    with mvr.World(2) as w:
        for v in range(V):
            w.upload(V + v, scans[v])
        new, info = w.ring_run(list(range(V)), [V + v for v in range(V)], edges, poses0, 8.0, origin, steps=6)
        assert w.ctx(0).comm_info() == (0, 2, 2) and w.ctx(1).comm_info() == (1, 2, 2)
    assert info["pair_n"] == rinfo["pair_n"]
    assert np.abs(np.asarray(new) - ref).max() < 1e-9


@pytest.mark.timeout(600)
def test_bench_with_two_ranks(mvr):
    """`python bench.py --gpus 2` end to end: two processes, library-owned RCCL communicator, one JSON line with ranks = 2"""
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--points", "20000", "--no-cpu-baseline",
                        "--no-bruteforce-pass", "--repeats", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["rccl_ranks"] == 2 and line["value"] > 0


@pytest.mark.timeout(300)
def test_a_process_that_imports_torch_after_a_collective_exits_cleanly(built):
    """A process that ran a collective through the library's RCCL (PyTorch-ROCm's copy) and imported torch only afterwards used to
    abort at interpreter exit ("double free or corruption"); the package now imports torch ahead of its first
    multi-GPU entry point.  In a fresh interpreter: a world of one, a sharded run, THEN `import torch` -- exit code 0."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, importlib
import numpy as np
sys.path.insert(0, %r)
mvr = importlib.import_module("multi-view-registration_amd")
assert "torch" not in sys.modules
V = 4
sp = mvr.synth_params(V, 3)
piv, ax = mvr.synth_prior(sp)
poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
edges = [(i, (i + 1) %% V) for i in range(V)]
with mvr.Context(0) as ctx:
    ctx.comm_init(mvr.comm_unique_id(), 0, 1)
    for v in range(V):
        ctx.upload(V + v, mvr.synth_view(sp, v, 3000))
    ctx.ring_run_sharded(list(range(V)), [V + v for v in range(V)], edges, poses0, 4.0, np.array(sp.pivot), steps=2)
import torch
print("done", torch.cuda.device_count() >= 1)
""" % root
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "done True" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
