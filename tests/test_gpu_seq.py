"""GPU tests of the target-sharded sequential mode (seq.ShardedSequentialICP over seq.HipPart): a 1-GPU "fake
world" walks G shards serially (each shard its own library context, reductions between them by torch on one
stream) and must reproduce the oracle's unsharded registrationICP and the single-context mvr_icp_align loop."""
import importlib

import numpy as np
import pytest

import ref_driver
from conftest import PKG

pytestmark = pytest.mark.gpu

V, N, MAX_D = 12, 4000, 6.0


@pytest.fixture(scope="module")
def seq(mvr):
    return importlib.import_module(PKG + ".seq")


@pytest.fixture(scope="module")
def scene(mvr, orc):
    sp = mvr.synth_params(V, 5)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = ref_driver.init_poses(orc, V, piv, ax)
    ref = {False: ref_driver.sequential_icp(orc, scans, poses0, orc.make_params(max_dist=MAX_D), fitness_last=False),
           True: ref_driver.sequential_icp(orc, scans, poses0, orc.make_params(max_dist=MAX_D, max_iter=3, feps=-1e300), fitness_last=False)}
    return sp, scans, poses0, ref


@pytest.mark.parametrize("nn_mode", [1, 0], ids=["culled", "brute"])
@pytest.mark.parametrize("parts", [1, 2, 3])
@pytest.mark.parametrize("multi_iter", [False, True])
def test_sharded_sequential_fake_world(mvr, seq, scene, parts, nn_mode, multi_iter):
    import torch
    sp, scans, poses0, ref = scene
    ref_poses, ref_log = ref[multi_iter]
    ts = torch.cuda.Stream(device=0)
    ps = [seq.HipPart(scans, device=0, tstream=ts) for _ in range(parts)]
    try:
        for p in ps:
            p.ctx.tune(nn_mode=nn_mode)
        drv = seq.ShardedSequentialICP(ps, V, N, parts, origin=np.array(sp.pivot))
        params = mvr.icp_params(max_dist=MAX_D) if not multi_iter else mvr.icp_params(max_dist=MAX_D, max_iter=3, feps=-1e300)
        poses, log = drv.run(poses0, params)
        assert [e["n_corr"] for e in log] == [e["n_corr"] for e in ref_log]
        assert [(e["iterations"], e["state"]) for e in log] == [(e["iterations"], e["state"]) for e in ref_log]
        for v in range(V):
            assert np.abs(poses[v][:3, :3] - ref_poses[v][:3, :3]).max() < 1e-5, v
            assert np.abs(poses[v][:3, 3] - ref_poses[v][:3, 3]).max() < 1e-4, v
        for a, b in zip(log, ref_log):
            assert abs(a["mse"] - b["mse"]) <= 1e-7 * max(1.0, b["mse"])
        # shard sizes: every part holds its slice of all V merged scans
        for k, p in enumerate(ps):
            assert p.ctx.size(p.TARGET) == V * (drv.bounds[k + 1] - drv.bounds[k])
    finally:
        for p in ps:
            p.close()


def test_global_numbering_and_ownership(gpu, mvr, orc):
    """mvr_nn_forward_keys / mvr_pair_moments2_from_keys on a hand-built shard: keys carry GLOBAL indices, ties go to
    the lowest global index after the MIN, and a rank only reduces the matches it owns."""
    import torch
    rng = np.random.default_rng(5)
    full = np.ones((3000, 4), np.float32); full[:, :3] = rng.standard_normal((3000, 3)) * 10 + [0, 0, 900]
    full[1500:] = full[:1500]                                   # every target point exists twice: exact ties across shards
    src = np.ones((800, 4), np.float32); src[:, :3] = full[rng.integers(0, 1500, 800), :3] + rng.standard_normal((800, 3)).astype(np.float32) * 0.05
    gpu.upload(5, full); gpu.upload(6, src)
    # shard A = global [0, 1000) + [2000, 3000), shard B = global [1000, 2000)
    gpu.clear(7); gpu.append_range(7, 5, 0, 1000, 0); gpu.append_range(7, 5, 2000, 1000, 2000)
    gpu.clear(8); gpu.append_range(8, 5, 1000, 1000, 1000)
    ka = torch.empty(800, dtype=torch.int64, device="cuda"); kb = torch.empty_like(ka)
    gpu.nn_forward_keys(6, 7, 1e9, ka.data_ptr()); gpu.nn_forward_keys(6, 8, 1e9, kb.data_ptr())
    gpu.sync()
    k = torch.minimum(ka, kb)
    oi, od = orc.nn(src, full, kdtree=False)                    # brute force: lowest index on ties
    assert np.array_equal((k.cpu().numpy() & 0xFFFFFFFF).astype(np.uint32), oi)
    assert np.array_equal((k.cpu().numpy() >> 32).astype(np.uint32), np.ascontiguousarray(od, np.float32).view(np.uint32))
    rows = []
    for slot in (7, 8):
        r = torch.zeros(32, dtype=torch.float64, device="cuda")
        gpu.pair_moments2_from_keys(6, slot, k.data_ptr(), 2.0, np.zeros(3), r.data_ptr())
        gpu.sync()
        rows.append(r.cpu().numpy())
    whole = gpu.pair_moments2(6, 5, 2.0, np.zeros(3))           # unsharded
    tot = rows[0] + rows[1]
    assert tot[0] == whole.n and rows[0][0] > 0 and rows[1][0] > 0
    assert np.allclose(tot[4:7], np.ctypeslib.as_array(whole.sp), rtol=1e-12)
    assert np.allclose(tot[22:31], np.ctypeslib.as_array(whole.spq), rtol=1e-12)
