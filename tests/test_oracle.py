"""CPU tests of the oracle (oracle/mvr_oracle.c) -- the checker itself.

The reference holds no tests or golden vectors for this path (SURVEY.md
section 4/8c: parity unpinned), so the oracle is validated against independent
implementations available offline (scipy cKDTree for exact 1-NN, numpy SVD for
Kabsch) and against analytic known-answer tests, the KAT list of SURVEY 8(c).
"""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from conftest import rand_cloud

REF = dict(reciprocal=True, max_dist=4.0, max_iter=10, teps=1e-6, feps=64.0)  # registrator.cpp:551-560


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def rigid(R, t):
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


# ----------------------------------------------------------------- distances

def test_dist2_spec_is_uncontracted(orc):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((2000, 3)).astype(np.float32) * 100
    b = rng.standard_normal((2000, 3)).astype(np.float32) * 100
    n_diff = 0
    for p, q in zip(a, b):
        d = p - q                                    # float32
        exp = np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])) + np.float32(d[2] * d[2])
        got = orc.dist2(p, q)
        assert np.float32(got) == np.float32(exp)
        ff = np.float32(orc.dist2(p, q, fma=True))
        exact = float(d[0]) ** 2 + float(d[1]) ** 2 + float(d[2]) ** 2
        assert abs(float(ff) - exact) <= 2 * np.spacing(np.float32(exact))
        n_diff += ff != np.float32(got)
    assert n_diff > 0      # the two modes are genuinely different roundings


# ---------------------------------------------------------------------- NN

@pytest.mark.parametrize("fma", [False, True])
def test_nn_kdtree_equals_brute_with_ties(orc, fma):
    rng = np.random.default_rng(1)
    t = rand_cloud(rng, 3000)
    # engineered exact ties: duplicates and points mirrored about a query
    t[100] = t[7]; t[2000] = t[7]; t[2999] = t[0]
    q = rand_cloud(rng, 1500)
    q[0] = t[7]
    q[1, :3] = (0, 0, 900); t[50, :3] = (1, 0, 900); t[40, :3] = (-1, 0, 900); t[60, :3] = (0, 1, 900)
    bi, bd = orc.nn(q, t, fma=fma, kdtree=False)
    ki, kd = orc.nn(q, t, fma=fma, kdtree=True)
    assert np.array_equal(bi, ki)
    assert np.array_equal(bd.view(np.uint32), kd.view(np.uint32))
    assert bi[0] == 7 and bd[0] == 0.0          # lowest index among 3 identical points
    assert bi[1] == 40                           # lowest index among equidistant points


def test_nn_against_scipy(orc):
    rng = np.random.default_rng(2)
    t, q = rand_cloud(rng, 5000), rand_cloud(rng, 2000)
    bi, bd = orc.nn(q, t, kdtree=True)
    d, i = cKDTree(t[:, :3].astype(np.float64)).query(q[:, :3].astype(np.float64))
    same = bi == i
    # scipy works in f64; indices may differ only on float32 near-ties
    assert same.mean() > 0.999
    assert np.allclose(bd[same], d[same] ** 2, rtol=1e-5, atol=1e-6)
    assert np.allclose(np.sqrt(bd[~same]), d[~same], rtol=1e-5)


def test_nn_edge_cases(orc):
    rng = np.random.default_rng(3)
    q = rand_cloud(rng, 5)
    i, d = orc.nn(q, np.empty((0, 4), np.float32))
    assert np.all(i == 0xFFFFFFFF) and np.all(np.isinf(d))
    i, d = orc.nn(np.empty((0, 4), np.float32), q)
    assert len(i) == 0
    for kd in (False, True):
        i, d = orc.nn(q, q[:1], kdtree=kd)
        assert np.all(i == 0)


# --------------------------------------------------------- correspondences

def test_reciprocal_kat_hand_built(orc):
    # source s0..s3, target t0..t2 on a line; non-mutual NNs are rejected
    s = np.array([[0, 0, 0, 1], [1.0, 0, 0, 1], [1.4, 0, 0, 1], [10, 0, 0, 1]], np.float32)
    t = np.array([[0.1, 0, 0, 1], [1.3, 0, 0, 1], [30, 0, 0, 1]], np.float32)
    one = orc.correspondences(s, t, 4.0, reciprocal=False, kdtree=False)
    assert list(one["query"]) == [0, 1, 2] and list(one["match"]) == [0, 1, 1]   # s3 is > 4 away
    rec = orc.correspondences(s, t, 4.0, reciprocal=True, kdtree=False)
    # t1's NN in source is s2 (0.1) not s1 (0.3): s1 dropped
    assert list(rec["query"]) == [0, 2] and list(rec["match"]) == [0, 1]
    assert np.allclose(rec["dist2"], [0.01, 0.01], atol=1e-6)                  # squared distances
    # threshold is on the squared distance, inclusive
    rec2 = orc.correspondences(s, t, 0.1 + 1e-6, reciprocal=True, kdtree=False)
    assert len(rec2) == 2
    assert len(orc.correspondences(s, t, 0.05, reciprocal=True, kdtree=False)) == 0


@pytest.mark.parametrize("reciprocal", [False, True])
def test_correspondences_kdtree_equals_brute(orc, reciprocal):
    rng = np.random.default_rng(4)
    t = rand_cloud(rng, 4000, scale=20)
    s = t[rng.permutation(4000)[:2500]].copy()
    s[:, :3] += rng.standard_normal((2500, 3)).astype(np.float32) * 0.5
    a = orc.correspondences(s, t, 1.0, reciprocal=reciprocal, kdtree=False)
    b = orc.correspondences(s, t, 1.0, reciprocal=reciprocal, kdtree=True)
    assert len(a) > 500 and a.tobytes() == b.tobytes()
    assert np.all(np.diff(a["query"]) > 0)          # ascending source order


# ---------------------------------------------------------------- Umeyama

def test_svd3_against_numpy(orc):
    rng = np.random.default_rng(5)
    mats = [rng.standard_normal((3, 3)) for _ in range(200)]
    mats += [np.outer(rng.standard_normal(3), rng.standard_normal(3)),          # rank 1
             rng.standard_normal((3, 2)) @ rng.standard_normal((2, 3)),         # rank 2
             np.zeros((3, 3)), np.eye(3), -np.eye(3), np.diag([3.0, 3.0, 1e-9])]
    for A in mats:
        U, S, V = orc.svd3(A)
        assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-12 * max(1, np.abs(A).max()))
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-12) and np.allclose(V.T @ V, np.eye(3), atol=1e-12)
        assert S[0] >= S[1] >= S[2] >= 0
        assert np.allclose(S, np.linalg.svd(A, compute_uv=False), atol=1e-12 * max(1, np.abs(A).max()))


def kabsch_numpy(P, Q):
    mp, mq = P.mean(0), Q.mean(0)
    H = (Q - mq).T @ (P - mp) / len(P)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(U) * np.linalg.det(Vt))])
    R = U @ D @ Vt
    return rigid(R, mq - R @ mp)


def test_umeyama_recovers_known_transform(orc):
    rng = np.random.default_rng(6)
    src = rand_cloud(rng, 500, scale=50)
    R, t = rot([0.3, -1, 0.5], 0.2), np.array([2.0, -1.0, 3.0])
    tgt = src.copy()
    tgt[:, :3] = (src[:, :3].astype(np.float64) @ R.T + t).astype(np.float32)
    corr = np.zeros(500, orc.CORR_DTYPE)
    corr["query"] = corr["match"] = np.arange(500)
    T, mom = orc.umeyama(src, tgt, corr)
    assert np.abs(T[:3, :3] - R).max() < 1e-6 and np.abs(T[:3, 3] - t).max() < 1e-3
    Tn = kabsch_numpy(src[:, :3].astype(np.float64), tgt[:, :3].astype(np.float64))
    assert np.abs(T - Tn).max() < 2e-4 and np.abs(T[:3, :3] - Tn[:3, :3]).max() < 1e-6
    assert mom[0] == 500
    # < 3 correspondences -> refused (PCL min_number_correspondences_ = 3)
    assert orc.umeyama(src, tgt, corr[:2])[0] is None


def test_umeyama_reflection_branch(orc):
    # a mirrored target: the best ORTHOGONAL map has det -1, Umeyama must
    # return a proper rotation (S(2) = -1 branch)
    rng = np.random.default_rng(7)
    src = rand_cloud(rng, 200, scale=10, centre=(0, 0, 0))
    tgt = src.copy(); tgt[:, 2] *= -1
    corr = np.zeros(200, orc.CORR_DTYPE); corr["query"] = corr["match"] = np.arange(200)
    T, _ = orc.umeyama(src, tgt, corr)
    R = T[:3, :3].astype(np.float64)
    assert abs(np.linalg.det(R) - 1) < 1e-5 and np.allclose(R @ R.T, np.eye(3), atol=1e-5)
    assert np.abs(T - kabsch_numpy(src[:, :3].astype(float), tgt[:, :3].astype(float))).max() < 1e-4


# -------------------------------------------------------------- transforms

def test_transform_semantics(orc):
    rng = np.random.default_rng(8)
    pts = rand_cloud(rng, 1000)
    T = rigid(rot([1, 2, 3], 0.4), [5, -3, 2])
    o32 = orc.transform_f32(T, pts)
    Tf = T.astype(np.float32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    for r in range(3):   # ((m0 x + m1 y) + m2 z) + m3 with every op rounded to f32
        exp = ((Tf[r, 0] * x + Tf[r, 1] * y) + Tf[r, 2] * z) + Tf[r, 3]
        assert np.array_equal(o32[:, r], exp)
    assert np.all(o32[:, 3] == 1)
    o64 = orc.transform_f64(T, pts)
    exp64 = (pts[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    assert np.abs(o64[:, :3] - exp64).max() <= np.spacing(np.float32(1000))
    # in-place
    assert np.array_equal(orc.transform_f32(np.eye(4), pts)[:, :3], pts[:, :3])


def test_turntable_prior(orc):
    # point_cloud.cpp:409: ((view<7)?(-view):(12-view))*pi/6
    for v in range(12):
        exp = (-v if v < 7 else 12 - v) * np.pi / 6
        assert abs(orc.turntable_angle(v, 12) - exp) < 1e-15
    piv, ax = np.array([1.0, 2, 3]), np.array([0, 0, 2.0])
    T = orc.axis_rotation(piv, ax, np.pi / 2)
    assert np.allclose(T @ np.array([2, 2, 3, 1.0]), [1, 3, 3, 1])     # right-handed about +z through pivot
    assert np.allclose(T @ np.append(piv, 1), np.append(piv, 1))


# ---------------------------------------------------------------------- ICP

def test_icp_one_iteration_under_reference_settings(orc):
    """with euclidean_fitness_epsilon = 64 the
    relative-MSE test fires on iteration 1 (previous MSE starts at DBL_MAX)."""
    rng = np.random.default_rng(9)
    tgt = rand_cloud(rng, 3000, scale=30)
    src = tgt[:2000].copy()
    src[:, :3] = (src[:, :3].astype(np.float64) @ rot([0, 1, 0], 0.01).T + [0.3, 0.2, -0.1]).astype(np.float32)
    out, T, st, rc = orc.icp_align(src, tgt, orc.make_params(max_iter=1000, **{k: REF[k] for k in ("reciprocal", "max_dist", "teps", "feps")}))
    assert rc == 0 and st["iterations"] == 1 and st["converged"] and st["state"] in ("REL_MSE", "TRANSFORM")
    # eps = -DBL_MAX, transformation eps 0 -> runs to max_iterations
    p = orc.make_params(max_iter=7, teps=0.0, feps=-np.finfo(np.float64).max, max_dist=4.0)
    out, T7, st7, rc = orc.icp_align(src, tgt, p)
    assert st7["iterations"] == 7 and st7["state"] == "ITERATIONS"
    assert st7["mse"] < st["mse"]
    # output is final * input recomputed from the original input
    assert np.array_equal(out, orc.transform_f32(T7, src))


def test_icp_no_correspondences(orc):
    rng = np.random.default_rng(10)
    src, tgt = rand_cloud(rng, 100), rand_cloud(rng, 100, centre=(1e4, 0, 0))
    out, T, st, rc = orc.icp_align(src, tgt, orc.make_params(max_dist=1.0))
    assert rc != 0 and st["state"] == "NO_CORRESPONDENCES" and not st["converged"]
    assert np.array_equal(T, np.eye(4, dtype=np.float32))


def test_icp_kdtree_equals_brute(orc):
    rng = np.random.default_rng(11)
    tgt = rand_cloud(rng, 1500, scale=20)
    src = tgt[200:1200].copy(); src[:, :3] += np.float32(0.2)
    a = orc.icp_align(src, tgt, orc.make_params(max_iter=3, feps=-1e300, teps=0, kdtree=True))
    b = orc.icp_align(src, tgt, orc.make_params(max_iter=3, feps=-1e300, teps=0, kdtree=False))
    assert np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[0], b[0])


def test_fitness_score(orc):
    rng = np.random.default_rng(12)
    tgt = rand_cloud(rng, 800, scale=10)
    src = tgt[:300].copy(); src[:, 0] += np.float32(0.5)
    f = orc.fitness(src, tgt, np.eye(4))
    i, d = orc.nn(src, tgt)
    assert abs(f - d.astype(np.float64).mean()) < 1e-12
    f2 = orc.fitness(src, tgt, np.eye(4), max_range=0.2)
    sel = d <= 0.2
    assert abs(f2 - d[sel].astype(np.float64).mean()) < 1e-12
    assert orc.fitness(src, tgt, np.eye(4), max_range=-1.0) == np.finfo(np.float64).max


# ---------------------------------------------------------------------- LUM

def test_pose_to_mat4(orc):
    pose = np.array([1, 2, 3, 0.1, -0.2, 0.3])
    T = orc.pose_to_mat4(pose)
    Rx, Ry, Rz = rot([1, 0, 0], 0.1), rot([0, 1, 0], -0.2), rot([0, 0, 1], 0.3)
    assert np.allclose(T[:3, :3], Rz @ Ry @ Rx) and np.allclose(T[:3, 3], [1, 2, 3])


def test_lum_kat_recovers_pose_offsets(orc):
    """SURVEY 8c KAT (6): 3 clouds with exact correspondences and known small
    pose offsets; LUM must bring them back into one frame (vertex 0 fixed)."""
    rng = np.random.default_rng(13)
    base = rand_cloud(rng, 400, scale=40, centre=(0, 0, 0))
    offs = [np.zeros(6), np.array([0.5, -0.3, 0.2, 0.01, -0.02, 0.015]), np.array([-0.4, 0.2, 0.1, -0.015, 0.01, 0.02])]
    clouds = []
    for o in offs:
        Ti = np.linalg.inv(orc.pose_to_mat4(o))          # cloud_v = T(o)^-1 * base  =>  pose o realigns it
        c = base.copy(); c[:, :3] = (base[:, :3].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        clouds.append(c)
    corr = np.zeros(400, orc.CORR_DTYPE); corr["query"] = corr["match"] = np.arange(400)
    edges = [(0, 1), (1, 2), (2, 0)]
    P, its = orc.lum_compute(clouds, edges, [corr] * 3, max_iterations=30)
    for v in range(3):
        Tv = orc.pose_to_mat4(P[v])
        moved = clouds[v][:, :3].astype(np.float64) @ Tv[:3, :3].T + Tv[:3, 3]
        assert np.abs(moved - base[:, :3]).max() < 2e-3, (v, np.abs(moved - base[:, :3]).max())
    assert np.all(P[0] == 0)


def test_lum_linearisation_is_the_jacobian_of_the_pose_map(orc):
    """Pins the sign pattern of computeEdge's M and of incidenceCorrection (tests/lum_kat.py)."""
    import lum_kat
    corr = np.zeros(4, orc.CORR_DTYPE); corr["query"] = corr["match"] = np.arange(4)

    def edge(src, tgt):
        a, b = np.ones((4, 4)), np.ones((4, 4))
        a[:, :3], b[:, :3] = src, tgt
        # (the oracle takes float32 clouds: p is rounded at the 3e-5 level, far inside the bound)
        n, MM, MZ, ss = orc.lum_edge(a.astype(np.float32), b.astype(np.float32), corr, np.zeros(6), np.zeros(6))
        assert n == 4
        return MM, MZ

    rng = np.random.default_rng(77)
    worst = lum_kat.check(lambda s, t: edge(s, t), orc.lum_incidence, rng)
    assert worst < 1e-2
    # the form SURVEY App. A.6 recalled (pitch sin/cos swapped in column 5 of the top block) is NOT the Jacobian
    X = np.array([3.0, -2.0, 5.0, 0.2, -0.3, 0.25]); p = np.array([10.0, -20.0, 917.0])
    H = orc.lum_incidence(X).copy()
    cx, sx, cy, sy = np.cos(X[3]), np.sin(X[3]), np.cos(X[4]), np.sin(X[4])
    H[0, 5] = X[1] * cx * sy + X[2] * sx * sy; H[1, 5] = -X[0] * cx * sy + X[2] * cy; H[2, 5] = -X[0] * sx * sy - X[1] * cy
    pp = lum_kat.pose_map(X, p)
    assert np.abs(lum_kat.numeric_jacobian(X, p) - lum_kat.M_full(pp) @ H).max() > 1.0


def test_umeyama_f64_oracle_vs_eigen_float_arithmetic(orc, mvr=None):
    """How far is the oracle's Umeyama (moments accumulated in f64, cast to f32 at the end) from the SAME estimate in
    Eigen's own arithmetic (float sums; orc_umeyama_f32 models sequential means and a sequential or blocked product)?
    Measured on synthetic turntable pairs: the ROTATION agrees far inside the 1e-5 bar; the TRANSLATION of the float
    arithmetic wanders by ~1e-4 mm at 4k pairs and ~4e-3 mm at 50k pairs (sum of M values ~917 in float: ulp 8-16 at
    1e8) -- i.e. the reference's own result is only defined to that level, and depends on Eigen's summation order.
    The f64 oracle is the exact-arithmetic centre of that cloud; the GPU path matches IT to 1e-4 mm."""
    import importlib
    mvr = importlib.import_module("multi-view-registration_amd")
    for n, trans_bound in ((10000, 1e-3), (200000, 2e-2)):
        sp = mvr.synth_params(12, 2)
        tgt, raw = mvr.synth_view(sp, 0, n), mvr.synth_view(sp, 1, n)
        piv, ax = mvr.synth_prior(sp)
        src = orc.transform_f64(mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 12)), raw)
        c = orc.correspondences(src, tgt, 4.0, kdtree=True)
        T64, _ = orc.umeyama(src, tgt, c)
        devs = []
        for block in (0, 8, 256, 1024):
            T32 = orc.umeyama_f32(src, tgt, c, block)
            dr, dt = np.abs(T32[:3, :3] - T64[:3, :3]).max(), np.abs(T32[:3, 3] - T64[:3, 3]).max()
            assert dr < 1e-5 and dt < trans_bound, (n, block, dr, dt)
            devs.append(dt)
        if n == 200000:
            assert max(devs) > 1e-4        # the float arithmetic's own noise exceeds the 1e-4 mm bar at this size


def test_refine_axis_restatement(orc):
    """orc_refine_axis (registrator.cpp:402-455) == numpy's least squares on the same stacked systems, and it
    recovers a known axis / pivot from exact turntable poses."""
    true_axis = np.array([-0.054323, -0.814921, -0.577020]); true_axis /= np.linalg.norm(true_axis)
    true_pivot = np.array([-13.382786, 50.223461, 917.4776])
    poses = [orc.axis_rotation(true_pivot, true_axis, orc.turntable_angle(v, 12)) for v in range(1, 12)]
    rng = np.random.default_rng(5)
    noisy = []
    for P in poses:
        Q = P.copy(); Q[:3, 3] += rng.normal(0, 0.05, 3)
        noisy.append(orc.mat4d_mul(orc.axis_rotation(true_pivot, rng.normal(0, 1, 3), rng.normal(0, 2e-4)), Q))
    for ps in (poses, noisy):
        rc, ax, pv = orc.refine_axis(ps, np.float32(true_pivot[1]))
        assert rc == 0
        A = np.concatenate([P[:3, :3] - np.eye(3) for P in ps] + [np.ones((1, 3))])
        b = np.zeros(len(A)); b[-1] = 1
        x = np.linalg.lstsq(A, b, rcond=None)[0]
        assert np.allclose(ax, (x / np.linalg.norm(x)).astype(np.float32), atol=2e-7)
        A[-1] = (0, 1, 0)
        b = np.concatenate([-P[:3, 3] for P in ps] + [[np.float32(true_pivot[1])]])
        x = np.linalg.lstsq(A, b, rcond=None)[0]
        assert np.allclose(pv, x.astype(np.float32), rtol=1e-6, atol=1e-4)
    rc, ax, pv = orc.refine_axis(poses, np.float32(true_pivot[1]))
    assert abs(abs(ax @ true_axis) - 1) < 1e-6
    # the pivot is only defined up to a shift along the axis; with p_y pinned it is the true pivot
    assert np.abs(pv - true_pivot).max() < 1e-3
    assert orc.refine_axis([], 0.0)[0] == -1


def test_lum_edge_structure(orc):
    rng = np.random.default_rng(14)
    a, b = rand_cloud(rng, 50, scale=5), rand_cloud(rng, 50, scale=5)
    corr = np.zeros(50, orc.CORR_DTYPE); corr["query"] = corr["match"] = np.arange(50)
    n, MM, MZ, ss = orc.lum_edge(a, b, corr, np.zeros(6), np.zeros(6))
    assert n == 50 and np.allclose(MM, MM.T) and MM[0, 0] == 50 and ss > 0
    av = 0.5 * (a[:, :3].astype(float) + b[:, :3].astype(float)); df = a[:, :3].astype(float) - b[:, :3].astype(float)
    assert np.allclose(MZ[:3], df.sum(0))
    assert np.isclose(MM[3, 3], (av[:, 1] ** 2 + av[:, 2] ** 2).sum())
    assert np.isclose(MM[0, 4], -av[:, 1].sum()) and np.isclose(MM[1, 3], -av[:, 2].sum())
    # < 3 pairs -> reported, no sums
    n2, _, _, _ = orc.lum_edge(a, b, corr[:2], np.zeros(6), np.zeros(6))
    assert n2 == 2


def test_solve_dense(orc):
    rng = np.random.default_rng(15)
    A = rng.standard_normal((20, 20)) + 5 * np.eye(20); b = rng.standard_normal(20)
    assert np.allclose(orc.solve_dense(A, b), np.linalg.solve(A, b))
    assert orc.solve_dense(np.zeros((3, 3)), np.ones(3)) is None


# ------------------------------------------------- point-to-plane (extension)

def test_p2plane_kat_recovers_small_motion(orc, mvr):
    """EXTENSION (no reference counterpart): PCL's linearised point-to-plane
    estimator recovers a small known rigid motion from exact correspondences."""
    sp = mvr.synth_params(12, 4)
    tgt, nrm = mvr.synth_view(sp, 0, 4000, normals=True)
    ang, t = 0.004, np.array([0.05, -0.03, 0.02])
    R = rot([0.2, 1.0, -0.4], ang)
    src = tgt.copy()
    src[:, :3] = ((tgt[:, :3].astype(np.float64) - t) @ R).astype(np.float32)      # src = R^T (tgt - t)
    corr = np.zeros(4000, orc.CORR_DTYPE); corr["query"] = corr["match"] = np.arange(4000)
    T, sums = orc.p2plane(src, tgt, nrm, corr)
    assert sums[27] == 4000
    assert np.abs(T[:3, :3] - R).max() < 2e-5 and np.abs(T[:3, 3] - t).max() < 2e-2    # linearisation error ~ angle^2/2 * |p| (|p| ~ 920 mm)
    moved = src[:, :3].astype(np.float64) @ T[:3, :3].T.astype(np.float64) + T[:3, 3]
    d = ((moved - tgt[:, :3]) * nrm[:, :3]).sum(1)
    assert np.abs(d).max() < 2e-2
    assert orc.p2plane(src, tgt, nrm, corr[:2])[0] is None


def test_multithreaded_correspondences_equal_single_thread(orc):
    """orc_correspondences_mt (OpenMP over queries, the courtesy CPU baseline of bench.py) == the serial kd-tree path."""
    rng = np.random.default_rng(31)
    src = np.ones((6000, 4), np.float32); src[:, :3] = rng.standard_normal((6000, 3)) * 5 + [0, 0, 900]
    tgt = np.ones((7000, 4), np.float32); tgt[:, :3] = rng.standard_normal((7000, 3)) * 5 + [0, 0, 900]
    for rec in (True, False):
        a = orc.correspondences(src, tgt, 0.8, reciprocal=rec, kdtree=True)
        for th in (1, 3, 8):
            b = orc.correspondences_mt(src, tgt, 0.8, th, reciprocal=rec)
            assert np.array_equal(a, b)


def noisy_surface(rng, n, outliers, clusters=()):
    """a dense sheet (one big component at r = 2.5), isolated outliers and a few small far-away clusters"""
    sheet = np.c_[rng.uniform(0, 60, (n, 2)), 900 + rng.normal(0, 0.1, n)]
    far = rng.uniform(-200, 200, (outliers, 3)) + [0, 0, 1500]
    parts = [sheet, far]
    for k, m in enumerate(clusters):
        parts.append(rng.normal(0, 0.3, (m, 3)) + [300 + 40 * k, -100, 700])
    p = np.concatenate(parts)
    p = p[rng.permutation(len(p))]
    out = np.ones((len(p), 4), np.float32); out[:, :3] = p
    return out


def test_denoise_equals_radius_graph_components_scipy(orc):
    """The oracle's denoise against an independent construction: scipy's cKDTree.query_pairs(r) edges +
    scipy.sparse.csgraph.connected_components, then the reference's rule (keep components >= threshold, ordered by
    their smallest index, members ascending) -- and, on a small cloud, against the Delaunay construction itself
    (scipy.spatial.Delaunay edges <= r), which is what the reference builds with CGAL."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(77)
    for n, outliers, clusters, thr, r in ((4000, 60, (9, 10, 25), 10, 2.5), (1500, 10, (3, 12), 4, 1.0), (300, 5, (), 1, 2.5)):
        pts = noisy_surface(rng, n, outliers, clusters)
        keep, lab, ncomp = orc.denoise(pts, thr, r)
        pairs = cKDTree(pts[:, :3].astype(np.float64)).query_pairs(r, output_type="ndarray")
        d = np.sqrt(((pts[pairs[:, 0], :3].astype(np.float64) - pts[pairs[:, 1], :3].astype(np.float64)) ** 2).sum(1))
        pairs = pairs[d <= r]
        g = coo_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(len(pts), len(pts)))
        nc, cl = connected_components(g, directed=False)
        assert nc == ncomp
        first = np.full(nc, len(pts)); np.minimum.at(first, cl, np.arange(len(pts)))
        assert np.array_equal(lab, first[cl])                         # label = smallest index of the component
        sizes = np.bincount(cl)
        exp = np.array(sorted(np.nonzero(sizes[cl] >= thr)[0], key=lambda i: (first[cl[i]], i)), np.uint32)
        assert np.array_equal(keep, exp)
    # Delaunay edges <= r give the same components (EMST is a subgraph of the Delaunay triangulation)
    pts = noisy_surface(rng, 800, 20, (6, 11))
    keep, lab, ncomp = orc.denoise(pts, 10, 2.5)
    tri = Delaunay(pts[:, :3].astype(np.float64))
    e = np.concatenate([tri.simplices[:, [a, b]] for a in range(4) for b in range(a + 1, 4)])
    d = np.sqrt(((pts[e[:, 0], :3].astype(np.float64) - pts[e[:, 1], :3].astype(np.float64)) ** 2).sum(1))
    e = e[d <= 2.5]
    nc, cl = connected_components(coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(len(pts), len(pts))), directed=False)
    first = np.full(nc, len(pts)); np.minimum.at(first, cl, np.arange(len(pts)))
    assert nc == ncomp and np.array_equal(lab, first[cl])
