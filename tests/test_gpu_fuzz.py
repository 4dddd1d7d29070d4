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
