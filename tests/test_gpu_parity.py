"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C-ABI of include/mvr_hip.h, against the CPU oracle on the same inputs and
against the committed golden fixtures.

Bars (BASELINE.json north_star): indices and float distances bit-exact;
rotation entries within 1e-5 and translation within 1e-4 mm of the oracle.
"""
import os

import numpy as np
import pytest

from conftest import load_golden, rand_cloud

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-5, 1e-4     # north_star tolerances (rotation entries / mm)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_pose_close(T, To):
    assert np.abs(np.asarray(T)[:3, :3] - np.asarray(To)[:3, :3]).max() <= ROT_TOL
    assert np.abs(np.asarray(T)[:3, 3] - np.asarray(To)[:3, 3]).max() <= TRANS_TOL


# ----------------------------------------------------------------- clouds

def test_cloud_roundtrip_and_append(gpu):
    rng = np.random.default_rng(100)
    a, b = rand_cloud(rng, 1000), rand_cloud(rng, 333)
    gpu.upload(0, a); gpu.upload(1, b[:, :3])          # 16-byte and packed 12-byte uploads
    assert np.array_equal(gpu.download(0), a)
    assert np.array_equal(gpu.download(1)[:, :3], b[:, :3]) and np.all(gpu.download(1)[:, 3] == 1)
    assert np.array_equal(gpu.download(0, packed=True), a[:, :3])
    gpu.append(0, 1)                                   # *target += source  (registrator.cpp:576)
    assert gpu.size(0) == 1333
    assert np.array_equal(gpu.download(0)[:, :3], np.concatenate([a, b])[:, :3])
    gpu.copy(2, 0); gpu.append(2, 2)                   # self-append
    assert gpu.size(2) == 2666 and np.array_equal(gpu.download(2)[1333:], gpu.download(2)[:1333])
    gpu.upload(3, np.empty((0, 4), np.float32))
    assert gpu.size(3) == 0 and gpu.download(3).shape == (0, 4)
    gpu.append(0, 3)
    assert gpu.size(0) == 1333
    # the caller keeps ownership: mutating the host array after upload has no effect
    c = a.copy(); gpu.upload(4, c); c[:] = 0
    assert np.array_equal(gpu.download(4), a)


def test_grown_target_searches_like_a_fresh_upload(gpu, orc, mvr):
    """A target that grows by mvr_cloud_append keeps its ordering and extends it by the appended cloud's own (no re-sort
    of the merged cloud: the sequential mode's growing target, registrator.cpp:576).  Whatever the ordering, searches
    are exact: the grown target answers like the same points uploaded at once -- ragged sizes (tile seams inside 256-point
    tiles), several appends, an append after the ordering was shared by a copy (falls back to a rebuild), a self-append."""
    rng = np.random.default_rng(106)
    parts = [rand_cloud(rng, n, scale=25) for n in (3000, 1777, 256, 5001, 33)]
    q = rand_cloud(rng, 4000, scale=25)
    gpu.upload(0, parts[0]); gpu.reserve(0, 40000); gpu.upload(5, q)
    whole = parts[0]
    for k, p in enumerate(parts[1:]):
        i0, d0 = gpu.nn(5, 0)                       # the target is indexed (and searched) before it grows again
        oi, od = orc.nn(q, whole, kdtree=True)
        assert np.array_equal(i0, oi) and np.array_equal(bits(d0), bits(od))
        gpu.upload(1, p)
        gpu.nn(1, 0)                                # the appended cloud has an ordering of its own (as an aligned scan has)
        if k == 2:
            gpu.copy(7, 0)                          # a copy now shares the target's ordering: the next append must not touch it
            gpu.nn(5, 7)
        gpu.append(0, 1)
        whole = np.concatenate([whole, p])
        if k == 2:
            i7, d7 = gpu.nn(5, 7)                   # the copy still answers for the OLD point set
            o7, od7 = orc.nn(q, whole[:len(whole) - len(p)], kdtree=True)
            assert np.array_equal(i7, o7) and np.array_equal(bits(d7), bits(od7))
    gi, gd = gpu.nn(5, 0)
    oi, od = orc.nn(q, whole, kdtree=True)
    assert gpu.size(0) == len(whole) and np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))
    gpu.upload(2, whole)
    fi, fd = gpu.nn(5, 2)
    assert np.array_equal(gi, fi) and np.array_equal(bits(gd), bits(fd))
    qq, mm, dd = gpu.correspondences(5, 0, 6.0)
    c = orc.correspondences(q, whole, 6.0, kdtree=True)
    assert np.array_equal(qq, c["query"]) and np.array_equal(mm, c["match"]) and np.array_equal(bits(dd), bits(c["dist2"]))
    gpu.append(0, 0)                                # self-append: the ordering is rebuilt
    si, sd = gpu.nn(5, 0)
    assert np.array_equal(si, oi) and np.array_equal(bits(sd), bits(od))      # duplicates sit at higher indices: the lowest index wins


def test_posed_copy_into_a_former_shard_slot_drops_its_segments(gpu, orc):
    """A slot that was a SHARD (mvr_cloud_set_global_base / append_range: the target-sharded sequential mode) and is then reused
    as the destination of a plain posed copy must lose its segment table with the old points (the batch path of
    mvr_cloud_transform kept it).  Otherwise every append takes the shard branch -- a segment per append until 'too many
    segments' -- and the growing model's ordering is rebuilt at every align.  Here: shard slot 3, pose a plain scan into it,
    append scans to it more often than a segment table has room for, and search it like a fresh upload of the same points."""
    rng = np.random.default_rng(1206)
    a = rand_cloud(rng, 2500, scale=20)
    gpu.upload(0, a)
    gpu.upload(3, rand_cloud(rng, 700, scale=20)); gpu.set_global_base(3, 12345)          # slot 3 is a shard now
    gpu.append_range(3, 0, 100, 50, 99000)
    T = np.eye(4); T[:3, 3] = (0.25, -0.5, 0.125)
    gpu.transform(3, 0, T)                                                                # plain posed copy into the former shard
    whole = gpu.download(3)
    assert np.array_equal(whole, orc.transform_f64(T, a))
    gpu.reserve(3, 2500 + 80 * 40)
    for k in range(70):                                                                   # more appends than kMaxSegs
        p = rand_cloud(rng, 40, scale=20)
        gpu.upload(1, p); gpu.append(3, 1)
        whole = np.concatenate([whole, p])
    q = rand_cloud(rng, 3000, scale=20); gpu.upload(5, q)
    gi, gd = gpu.nn(5, 3)
    gpu.upload(2, whole)
    fi, fd = gpu.nn(5, 2)
    assert gpu.size(3) == len(whole) and np.array_equal(gi, fi) and np.array_equal(bits(gd), bits(fd))
    oi, od = orc.nn(q, whole, kdtree=True)
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))


def test_transforms_bit_exact(gpu, orc):
    rng = np.random.default_rng(101)
    pts = rand_cloud(rng, 5000)
    ang = 0.3
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [3.25, -1.5, 0.75]
    gpu.upload(0, pts)
    gpu.transform_f32(1, 0, T)
    assert np.array_equal(bits(gpu.download(1)), bits(orc.transform_f32(T, pts)))
    gpu.transform(2, 0, T)                              # getTransformedPoints (f64 pose)
    assert np.array_equal(bits(gpu.download(2)), bits(orc.transform_f64(T, pts)))
    gpu.transform(0, 0, T)                              # in place
    assert np.array_equal(bits(gpu.download(0)), bits(orc.transform_f64(T, pts)))


# ---------------------------------------------------------------------- NN

@pytest.mark.parametrize("fma", [False, True])
@pytest.mark.parametrize("nq,nt", [(1, 1), (63, 31), (1000, 1024), (1025, 1025), (5000, 7777), (4096, 33), (300, 20000)])
def test_nn_bit_exact_ragged_sizes(gpu, orc, nq, nt, fma):
    rng = np.random.default_rng(nq * 7 + nt)
    q, t = rand_cloud(rng, nq, scale=30), rand_cloud(rng, nt, scale=30)
    gpu.upload(0, q); gpu.upload(1, t)
    gi, gd = gpu.nn(0, 1, fma=fma)
    oi, od = orc.nn(q, t, fma=fma, kdtree=nt > 2000)
    assert np.array_equal(gi, oi)
    assert np.array_equal(bits(gd), bits(od))


def test_nn_tie_break_lowest_index(gpu, orc):
    rng = np.random.default_rng(102)
    t = rand_cloud(rng, 6000, scale=20)
    # exact duplicates far apart in the array: inside one sub-tile, across
    # sub-tiles, across LDS tiles and across blocks
    for a, b in [(5, 9), (40, 1000), (100, 1030), (7, 5999), (2048, 2049), (3000, 4100)]:
        t[b] = t[a]
    q = t[[5, 9, 40, 1000, 100, 1030, 7, 5999, 2048, 3000, 4100]].copy()
    # equidistant (mirror) targets: query in the middle of two points
    t[10, :3] = (500, 0, 0); t[5500, :3] = (502, 0, 0)
    q = np.concatenate([q, np.array([[501, 0, 0, 1]], np.float32)])
    gpu.upload(0, q); gpu.upload(1, t)
    gi, gd = gpu.nn(0, 1)
    oi, od = orc.nn(q, t, kdtree=False)
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))
    assert list(gi[:11]) == [5, 5, 40, 40, 100, 100, 7, 7, 2048, 3000, 3000] and gi[11] == 10


def test_nn_empty_inputs(gpu):
    rng = np.random.default_rng(103)
    q = rand_cloud(rng, 10)
    gpu.upload(0, q); gpu.upload(1, np.empty((0, 4), np.float32))
    i, d = gpu.nn(0, 1)
    assert np.all(i == 0xFFFFFFFF) and np.all(np.isinf(d))
    i, d = gpu.nn(1, 0)
    assert len(i) == 0


def test_golden_pair(gpu, mvr):
    g = load_golden("pair_2x10000.npz")
    gpu.upload(0, g["tgt"]); gpu.upload(1, g["raw"])
    gpu.transform(1, 1, g["prior"])
    assert np.array_equal(bits(gpu.download(1)), bits(g["src"]))
    for fma in (0, 1):
        i, d = gpu.nn(1, 0, fma=bool(fma))
        assert np.array_equal(i, g["nn_idx_fma%d" % fma]) and np.array_equal(bits(d), bits(g["nn_d2_fma%d" % fma]))
    for name, rec in (("oneway", False), ("recip", True)):
        q, m, d = gpu.correspondences(1, 0, 4.0, reciprocal=rec)
        c = g["corr_" + name]
        assert np.array_equal(q, c["query"]) and np.array_equal(m, c["match"]) and np.array_equal(bits(d), bits(c["dist2"]))
    pm = gpu.pair_moments(1, 0, 4.0)
    mom = g["moments"]
    assert pm.n == mom[0]
    assert np.allclose(np.array(pm.mean_src), mom[1:4], rtol=1e-13) and np.allclose(np.array(pm.mean_tgt), mom[4:7], rtol=1e-13)
    assert abs(pm.mse - mom[7]) < 1e-12 and np.allclose(np.array(pm.sigma), mom[8:17], rtol=1e-10, atol=1e-10)
    T, st, rc = gpu.icp_align(1, 0, 2, mvr.icp_params())
    assert rc == 0 and st["iterations"] == g["align1_stats"][0] == 1 and st["n_corr"] == g["align1_stats"][1]
    assert abs(st["mse"] - g["align1_stats"][2]) < 1e-12
    assert_pose_close(T, g["align1_T"])
    assert abs(gpu.fitness(1, 0, T) - g["fitness1"][0]) < 1e-6
    p5 = mvr.icp_params(max_iter=5, teps=0.0, feps=-np.finfo(np.float64).max)
    T5, st5, rc = gpu.icp_align(1, 0, 2, p5)
    assert st5["iterations"] == 5 and st5["state"] == "ITERATIONS"
    assert_pose_close(T5, g["align5_T"])
    assert st5["n_corr"] == g["align5_stats"][1] and abs(st5["mse"] - g["align5_stats"][2]) < 1e-9


# --------------------------------------------------------- correspondences

@pytest.mark.parametrize("reciprocal", [False, True])
def test_correspondences_bit_exact_20k(gpu, orc, mvr, reciprocal):
    sp = mvr.synth_params(12, 7)
    tgt, raw = mvr.synth_view(sp, 0, 20000), mvr.synth_view(sp, 11, 17001)
    piv, ax = mvr.synth_prior(sp)
    src = orc.transform_f64(mvr.axis_rotation(piv, ax, mvr.turntable_angle(11, 12)), raw)
    gpu.upload(0, tgt); gpu.upload(1, src)
    q, m, d = gpu.correspondences(1, 0, 4.0, reciprocal=reciprocal)
    c = orc.correspondences(src, tgt, 4.0, reciprocal=reciprocal, kdtree=True)
    assert len(c) > 3000
    assert np.array_equal(q, c["query"]) and np.array_equal(m, c["match"]) and np.array_equal(bits(d), bits(c["dist2"]))


def test_pair_moments_and_umeyama(gpu, orc, mvr):
    sp = mvr.synth_params(12, 8)
    tgt, raw = mvr.synth_view(sp, 0, 30000), mvr.synth_view(sp, 1, 30000)
    piv, ax = mvr.synth_prior(sp)
    src = orc.transform_f64(mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 12)), raw)
    gpu.upload(0, tgt); gpu.upload(1, src)
    c = orc.correspondences(src, tgt, 4.0, kdtree=True)
    To, mom = orc.umeyama(src, tgt, c)
    pm = gpu.pair_moments(1, 0, 4.0)
    assert pm.n == len(c)
    assert np.allclose(np.array(pm.sigma), mom[8:17], rtol=1e-10, atol=1e-10)
    T, sv = mvr.umeyama_from_moments(pm)
    assert_pose_close(T, To)
    # raw second moments: whole range == sum of sub-ranges (the multi-GPU split), and
    # they reproduce the centred moments
    origin = np.array(sp.pivot)
    full = gpu.pair_moments2(1, 0, 4.0, origin)
    assert full.n == len(c)
    pm2 = mvr.moments_from_moments2(full)
    assert np.allclose(np.array(pm2.sigma), np.array(pm.sigma), rtol=1e-8, atol=1e-8)
    T2, _ = mvr.umeyama_from_moments(pm2)
    assert_pose_close(T2, To)
    parts = [gpu.pair_moments2(1, 0, 4.0, origin, q_begin=b, q_count=n) for b, n in ((0, 10000), (10000, 7), (10007, 19993))]
    def row(m2):
        return np.concatenate([[m2.n], m2.sp, m2.sq, m2.spp, m2.sqq, m2.spq])
    total = sum(row(p) for p in parts)
    assert total[0] == full.n and np.allclose(total, row(full), rtol=1e-12, atol=1e-9)
    # determinism: bitwise identical moments run to run (no float atomics)
    again = gpu.pair_moments(1, 0, 4.0)
    assert bytes(again) == bytes(pm)


# ---------------------------------------------------------------------- ICP

def test_icp_align_multi_iteration_and_alias(gpu, orc, mvr):
    sp = mvr.synth_params(12, 9)
    tgt, raw = mvr.synth_view(sp, 0, 15000), mvr.synth_view(sp, 1, 15000)
    piv, ax = mvr.synth_prior(sp)
    src = orc.transform_f64(mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 12)), raw)
    gpu.upload(0, tgt); gpu.upload(1, src)
    for kw in (dict(), dict(max_iter=4, teps=0.0, feps=-1e300), dict(reciprocal=False, max_iter=3, teps=0.0, feps=-1e300),
               dict(fma=True)):
        T, st, rc = gpu.icp_align(1, 0, 2, mvr.icp_params(**kw))
        okw = dict(kw); okw.setdefault("kdtree", True)
        out, To, sto, _ = orc.icp_align(src, tgt, orc.make_params(**okw))
        assert rc == 0 and st["iterations"] == sto["iterations"] and st["state"] == sto["state"]
        assert st["n_corr"] == sto["n_corr"]
        assert_pose_close(T, To)
        got = gpu.download(2)
        assert np.abs(got[:, :3] - out[:, :3]).max() < 2e-4
        assert np.array_equal(bits(got), bits(orc.transform_f32(T, src)))   # out = final * input
    # aliased align(*source_) of registrator.cpp:920: output slot == input slot
    gpu.copy(3, 1)
    T, st, rc = gpu.icp_align(3, 0, 3, mvr.icp_params())
    assert np.array_equal(bits(gpu.download(3)), bits(orc.transform_f32(T, src)))


def test_icp_not_enough_correspondences(gpu, mvr):
    rng = np.random.default_rng(104)
    gpu.upload(0, rand_cloud(rng, 500)); gpu.upload(1, rand_cloud(rng, 500, centre=(1e4, 0, 0)))
    T, st, rc = gpu.icp_align(1, 0, 2, mvr.icp_params(max_dist=1.0))
    assert rc == mvr.E_NOCORR and st["state"] == "NO_CORRESPONDENCES" and not st["converged"]
    assert np.array_equal(T, np.eye(4, dtype=np.float32))
    gpu.upload(2, np.empty((0, 4), np.float32))
    T, st, rc = gpu.icp_align(2, 0, -1, mvr.icp_params())
    assert rc == mvr.E_NOCORR
    T, st, rc = gpu.icp_align(1, 2, -1, mvr.icp_params())
    assert rc == mvr.E_NOCORR


def test_fitness(gpu, orc):
    rng = np.random.default_rng(105)
    tgt = rand_cloud(rng, 9000, scale=10); src = tgt[:4000].copy(); src[:, 0] += np.float32(0.3)
    gpu.upload(0, tgt); gpu.upload(1, src)
    T = np.eye(4); T[0, 3] = -0.1
    assert abs(gpu.fitness(1, 0, T) - orc.fitness(src, tgt, T)) < 1e-12
    assert abs(gpu.fitness(1, 0, T, max_range=0.03) - orc.fitness(src, tgt, T, max_range=0.03)) < 1e-12
    assert gpu.fitness(1, 0, T, max_range=-1.0) == np.finfo(np.float64).max


# ------------------------------------------------------ BASELINE full size

def test_full_size_200k_pair(gpu, orc, mvr):
    """BASELINE configs[1] shape: 2 scans x 200k points.  Oracle = kd-tree
    (seconds), plus size-independent properties."""
    sp = mvr.synth_params(12, 2)
    tgt, raw = mvr.synth_view(sp, 0, 200000), mvr.synth_view(sp, 1, 200000)
    piv, ax = mvr.synth_prior(sp)
    prior = mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 12))
    gpu.upload(0, tgt); gpu.upload(1, raw); gpu.transform(1, 1, prior)
    src = gpu.download(1)
    assert np.array_equal(bits(src), bits(orc.transform_f64(prior, raw)))
    gi, gd = gpu.nn(1, 0)
    oi, od = orc.nn(src, tgt, kdtree=True)
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))
    # property: a cloud's NN in itself is the identity with d2 = 0 (distinct points)
    si, sd = gpu.nn(0, 0)
    assert np.array_equal(si, np.arange(200000, dtype=np.uint32)) and np.all(sd == 0)
    q, m, d = gpu.correspondences(1, 0, 4.0)
    c = orc.correspondences(src, tgt, 4.0, kdtree=True)
    assert np.array_equal(q, c["query"]) and np.array_equal(m, c["match"]) and np.array_equal(bits(d), bits(c["dist2"]))
    # property: reciprocal correspondences are a one-to-one matching, symmetric under role swap
    assert len(np.unique(m)) == len(m)
    q2, m2, d2 = gpu.correspondences(0, 1, 4.0)
    assert set(zip(q.tolist(), m.tolist())) == set(zip(m2.tolist(), q2.tolist()))
    T, st, rc = gpu.icp_align(1, 0, 2, mvr.icp_params())
    out, To, sto, _ = orc.icp_align(src, tgt, orc.make_params())
    assert st["n_corr"] == sto["n_corr"] == len(c)
    assert_pose_close(T, To)
    if gpu.mode == "brute":
        assert st["evals"] >= 200000.0 * 200000.0
    else:       # the culled kernel evaluates a small fraction of the pairs, with identical results
        assert 200000.0 * 256 <= st["evals"] < 0.2 * 200000.0 * 200000.0


# ------------------------------------------------- point-to-plane (extension)

def test_point_to_plane_extension(gpu, orc, mvr):
    """BASELINE config 2 shape (2 scans, point-to-plane ICP).  No counterpart in
    the reference: parity is GPU vs this repo's own oracle."""
    sp = mvr.synth_params(12, 11)
    tgt, tn = mvr.synth_view(sp, 0, 30000, normals=True)
    raw = mvr.synth_view(sp, 1, 30000)
    piv, ax = mvr.synth_prior(sp)
    prior = mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 12))
    src = orc.transform_f64(prior, raw)
    gpu.upload(0, tgt); gpu.upload_normals(0, tn); gpu.upload(1, src)
    assert np.array_equal(gpu.download_normals(0)[:, :3], tn[:, :3])
    for kw in (dict(), dict(max_iter=4, teps=0.0, feps=-1e300)):
        T, st, rc = gpu.icp_align(1, 0, 2, mvr.icp_params(point_to_plane=True, **kw))
        out, To, sto, _ = orc.icp_align_p2plane(src, tgt, tn, orc.make_params(**kw))
        assert rc == 0 and st["iterations"] == sto["iterations"] and st["n_corr"] == sto["n_corr"]
        assert_pose_close(T, To)
    # point-to-plane converges faster than point-to-point on this surface
    Tp, stp, _ = gpu.icp_align(1, 0, 2, mvr.icp_params(max_iter=4, teps=0.0, feps=-1e300))
    assert st["mse"] < stp["mse"]
    # normals follow the cloud: rotate with a transform, drop on a fresh upload
    gpu.transform(3, 0, prior)
    rn = gpu.download_normals(3)
    assert np.abs(rn[:, :3] - tn[:, :3].astype(np.float64) @ prior[:3, :3].T).max() < 1e-6
    gpu.upload(3, src)
    assert len(gpu.download_normals(3)) == 0
    # without target normals the estimator is refused loudly
    gpu.upload(4, tgt)
    with pytest.raises(mvr.MvrError):
        gpu.icp_align(1, 4, 2, mvr.icp_params(point_to_plane=True))


def test_one_million_point_pair_properties(gpu, mvr):
    """BASELINE configs[4] point count (1M per scan): too big for the oracle in a test, so size-independent properties --
    a cloud's NN in itself is the identity at d2 = 0, reciprocal correspondences are a one-to-one matching that is
    symmetric under a role swap, every accepted pair is within max_dist, and the raw moments of the pair equal the
    moments recomputed on the host from the correspondence list (the culled kernel takes its 128-query path here)."""
    if gpu.mode in ("culled_w1", "culled_w2", "culled_w4"):
        pytest.skip("one wave-count variant is enough at this size")
    n = 1_000_000
    sp = mvr.synth_params(36, 4)
    tgt, raw = mvr.synth_view(sp, 0, n), mvr.synth_view(sp, 1, n)
    piv, ax = mvr.synth_prior(sp)
    gpu.upload(0, tgt); gpu.upload(1, raw)
    gpu.transform(1, 1, mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 36)))
    src = gpu.download(1)
    if gpu.mode != "brute":                         # 1e12 evaluations per search: the culled kernel only
        si, sd = gpu.nn(0, 0)
        assert np.array_equal(si, np.arange(n, dtype=np.uint32)) and np.all(sd == 0)
    q, m, d = gpu.correspondences(1, 0, 4.0)
    assert len(q) > 100000 and len(np.unique(m)) == len(m) and np.all(np.diff(q) > 0)
    assert np.all(d <= np.float32(16.0))
    dd = ((src[q, :3].astype(np.float32) - tgt[m, :3].astype(np.float32)) ** 2)
    assert np.array_equal(bits(((dd[:, 0] + dd[:, 1]) + dd[:, 2]).astype(np.float32)), bits(d))     # d2 = (dx2 + dy2) + dz2, rounded per op
    q2, m2, _ = gpu.correspondences(0, 1, 4.0)
    assert set(zip(q.tolist(), m.tolist())) == set(zip(m2.tolist(), q2.tolist()))
    origin = np.array(sp.pivot)
    mom = gpu.pair_moments2(1, 0, 4.0, origin)
    p = src[q, :3].astype(np.float64) - origin
    t = tgt[m, :3].astype(np.float64) - origin
    assert mom.n == len(q)
    assert np.allclose(np.ctypeslib.as_array(mom.sp), p.sum(0), rtol=1e-10) and np.allclose(np.ctypeslib.as_array(mom.sq), t.sum(0), rtol=1e-10)
    assert np.allclose(np.ctypeslib.as_array(mom.spq).reshape(3, 3), p.T @ t, rtol=1e-9)


@pytest.mark.gpu
def test_freed_blocks_serve_the_next_context_and_trim_gives_them_back(mvr):
    """csrc/mvr_pool.cpp: what a context frees stays in the library's process-wide cache (by device and size class) and serves
    the next context's requests of the same sizes; mvr_pool_trim hands every idle block back to the runtime.  Results do not
    depend on where a buffer came from: the second context's search equals the first's."""
    rng = np.random.default_rng(11)
    a = np.c_[rng.normal(size=(5000, 3)).astype(np.float32) * 30, np.ones(5000, np.float32)]
    b = a.copy(); b[:, :3] += rng.normal(size=(5000, 3)).astype(np.float32) * 0.2
    mvr.pool_trim()
    before = mvr.pool_trim()
    outs = []
    for _ in range(2):
        with mvr.Context(0) as ctx:
            ctx.upload(0, a); ctx.upload(1, b)
            outs.append(bytes(ctx.pair_moments2(0, 1, 4.0, np.zeros(3))))
    assert outs[0] == outs[1]
    after = mvr.pool_trim()
    if os.environ.get("MVR_POOL", "1") != "0":
        assert after["hits"] > before["hits"]                 # the second context found the first one's blocks
        assert after["freed_bytes"] > 0 and after["cached_bytes"] == 0
    assert mvr.pool_trim()["freed_bytes"] == 0                # nothing idle is left
