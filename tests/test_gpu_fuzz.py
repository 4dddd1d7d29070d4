"""Adversarial inputs for the exact NN kernels (GPU): degenerate geometry, mass
ties, clusters far apart, extreme sizes.  Both kernels (culled / brute force)
must agree bit for bit with the oracle's brute-force definition (lowest index
on ties), with and without a distance cap."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def cloud(xyz):
    p = np.ones((len(xyz), 4), np.float32)
    p[:, :3] = np.asarray(xyz, np.float32)
    return p


def cases():
    rng = np.random.default_rng(7)
    out = {}
    out["all_identical"] = (cloud(np.tile([1.0, 2.0, 900.0], (700, 1))), cloud(np.tile([1.0, 2.0, 900.0], (1500, 1))))
    t = rng.standard_normal((3000, 3)) * 5 + [0, 0, 900]
    out["every_target_duplicated"] = (cloud(rng.standard_normal((900, 3)) * 5 + [0, 0, 900]), cloud(np.concatenate([t, t[::-1]])))
    s = np.linspace(0, 100, 2000)
    out["collinear"] = (cloud(np.c_[s[::3] + 0.01, 0 * s[::3], 0 * s[::3] + 900]), cloud(np.c_[s, 0 * s, 0 * s + 900]))
    g = np.stack(np.meshgrid(np.arange(40.0), np.arange(40.0), [900.0]), -1).reshape(-1, 3)
    out["regular_grid_midpoints"] = (cloud(g[:1200] + [0.5, 0.5, 0]), cloud(g))          # 4-way exact ties
    a = rng.standard_normal((2500, 3)) + [0, 0, 900]
    b = rng.standard_normal((2500, 3)) + [5000, -3000, 900]
    out["two_far_clusters"] = (cloud(np.concatenate([a[:800], b[:800]])), cloud(np.concatenate([a, b])))
    out["queries_far_from_everything"] = (cloud(rng.standard_normal((500, 3)) + [1e4, 1e4, 1e4]), cloud(a))
    out["one_target"] = (cloud(rng.standard_normal((300, 3)) * 50), cloud([[3.0, -2.0, 1.0]]))
    out["one_query"] = (cloud([[0.3, 0.2, 900.1]]), cloud(a))
    out["planar_sheet"] = (cloud(np.c_[rng.uniform(0, 60, (1500, 2)), np.full(1500, 900.0)]),
                           cloud(np.c_[rng.uniform(0, 60, (5000, 2)), np.full(5000, 900.0)]))
    out["huge_coordinates"] = (cloud(rng.standard_normal((600, 3)) * 1e5), cloud(rng.standard_normal((2200, 3)) * 1e5))
    out["exactly_257_and_1025"] = (cloud(rng.standard_normal((257, 3)) * 3 + [0, 0, 900]), cloud(rng.standard_normal((1025, 3)) * 3 + [0, 0, 900]))
    return out


CASES = cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_nn_adversarial(gpu, orc, name):
    q, t = CASES[name]
    gpu.upload(0, q); gpu.upload(1, t)
    for fma in (False, True):
        gi, gd = gpu.nn(0, 1, fma=fma)
        oi, od = orc.nn(q, t, fma=fma, kdtree=False)
        assert np.array_equal(gi, oi), name
        assert np.array_equal(bits(gd), bits(od)), name


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("max_dist", [0.0, 0.75, 4.0, 1e9])
def test_correspondences_adversarial(gpu, orc, name, max_dist):
    q, t = CASES[name]
    gpu.upload(0, q); gpu.upload(1, t)
    for rec in (False, True):
        gq, gm, gd = gpu.correspondences(0, 1, max_dist, reciprocal=rec)
        c = orc.correspondences(q, t, max_dist, reciprocal=rec, kdtree=False)
        assert np.array_equal(gq, c["query"]) and np.array_equal(gm, c["match"]), (name, max_dist, rec)
        assert np.array_equal(bits(gd), bits(c["dist2"]))


def test_random_shapes_and_caps(gpu, orc):
    rng = np.random.default_rng(11)
    for trial in range(25):
        nq, nt = int(rng.integers(1, 4000)), int(rng.integers(1, 6000))
        scale = float(rng.choice([0.5, 5.0, 50.0]))
        q = cloud(rng.standard_normal((nq, 3)) * scale + [0, 0, 900])
        t = cloud(rng.standard_normal((nt, 3)) * scale + [0, 0, 900])
        if trial % 3 == 0:                      # sprinkle duplicates
            t[rng.integers(0, nt, nt // 10)] = t[rng.integers(0, nt, nt // 10)]
        gpu.upload(0, q); gpu.upload(1, t)
        gi, gd = gpu.nn(0, 1)
        oi, od = orc.nn(q, t, kdtree=False)
        assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od)), trial
        md = float(rng.choice([0.1, 1.0, 4.0]))
        gq, gm, gdd = gpu.correspondences(0, 1, md)
        c = orc.correspondences(q, t, md, kdtree=False)
        assert np.array_equal(gq, c["query"]) and np.array_equal(gm, c["match"]) and np.array_equal(bits(gdd), bits(c["dist2"])), trial


def test_two_query_groups_per_set(gpu, orc):
    """The 128-query variant of the culled kernel (picked automatically above 524k queries) on small inputs:
    forced with cull_q=2, for whichever wave count the fixture selected."""
    if gpu.mode == "brute":
        pytest.skip("knob of the culled kernel")
    rng = np.random.default_rng(23)
    gpu.tune(cull_q=2)
    try:
        for name in ("regular_grid_midpoints", "two_far_clusters", "every_target_duplicated", "exactly_257_and_1025"):
            q, t = CASES[name]
            gpu.upload(0, q); gpu.upload(1, t)
            gi, gd = gpu.nn(0, 1)
            oi, od = orc.nn(q, t, kdtree=False)
            assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od)), name
        for trial in range(6):
            nq, nt = int(rng.integers(1, 9000)), int(rng.integers(1, 12000))
            q = cloud(rng.standard_normal((nq, 3)) * 20 + [0, 0, 900])
            t = cloud(rng.standard_normal((nt, 3)) * 20 + [0, 0, 900])
            gpu.upload(0, q); gpu.upload(1, t)
            gq, gm, gdd = gpu.correspondences(0, 1, 2.0)
            c = orc.correspondences(q, t, 2.0, kdtree=False)
            assert np.array_equal(gq, c["query"]) and np.array_equal(gm, c["match"]) and np.array_equal(bits(gdd), bits(c["dist2"])), trial
    finally:
        gpu.tune(cull_q=0)


def test_fused_batch_random_graphs(gpu, mvr):
    """The fused pass (keys by Hilbert position, seeded reverse searches, pair groups, posed index refresh) against
    one-pair calls on random view graphs: random cloud sizes (empty ones included), duplicated points, self pairs,
    repeated pairs, query sub-ranges, caps.  Bit for bit.  MVR_FUZZ_TRIALS scales it (default 12)."""
    import os
    trials = int(os.environ.get("MVR_FUZZ_TRIALS", "12"))
    rng = np.random.default_rng(101)
    try:
        for trial in range(trials):
            V = int(rng.integers(2, 8))
            scale = float(rng.choice([0.5, 5.0, 40.0]))
            big = int(os.environ.get("MVR_FUZZ_MAX_POINTS", "5000"))          # > 16384: several ballot blocks / super boxes per cloud
            sizes = [int(rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 257, int(rng.integers(1, 5000)), int(rng.integers(1, big))]))
                     for _ in range(V)]
            if all(s == 0 for s in sizes):
                sizes[0] = 100
            raw = []
            for v in range(V):
                p = cloud(rng.standard_normal((sizes[v], 3)) * scale + [0, 0, 900])
                if sizes[v] > 10 and trial % 3 == 0:
                    p[rng.integers(0, sizes[v], sizes[v] // 8)] = p[rng.integers(0, sizes[v], sizes[v] // 8)]
                raw.append(p)
                gpu.upload(20 + v, p)
            poses = []
            for v in range(V):
                T = np.eye(4); T[:3, 3] = rng.standard_normal(3) * 0.2 * scale
                a = rng.standard_normal() * 0.05
                T[0, 0], T[0, 1], T[1, 0], T[1, 1] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
                poses.append(T)
            gpu.tune(pair_groups=int(rng.choice([1, 2, 4])), posed_refresh=int(rng.integers(0, 2)))
            gpu.transform_batch(list(range(V)), [20 + v for v in range(V)], poses)
            npairs = int(rng.integers(1, 20))
            pairs = [(int(rng.integers(0, V)), int(rng.integers(0, V))) for _ in range(npairs)]
            ranges = None
            if trial % 2:
                ranges = []
                for s, _ in pairs:
                    b = int(rng.integers(0, sizes[s] + 1))
                    ranges.append((b, None if rng.integers(0, 3) == 0 else int(rng.integers(0, sizes[s] - b + 1))))
            md = float(rng.choice([0.05, 0.5, 4.0])) * scale
            origin = np.array([0.0, 0.0, 900.0])
            rec = bool(rng.integers(0, 4))
            batch = gpu.pair_moments2_batch(pairs, md, origin, ranges=ranges, reciprocal=rec)
            for k, (s, t) in enumerate(pairs):
                one = gpu.pair_moments2(s, t, md, origin, q_begin=0 if ranges is None else ranges[k][0],
                                        q_count=None if ranges is None else ranges[k][1], reciprocal=rec)
                assert bytes(one) == bytes(batch[k]), (trial, k, pairs[k], sizes, one.n, batch[k].n)
    finally:
        gpu.tune(pair_groups=2, posed_refresh=1)
