"""CPU tests of the product's host side: the C-ABI library loads and exports
every symbol include/mvr_hip.h declares, the host solves (Umeyama from moments,
LUM from raw second moments) agree with the oracle, and the synthetic turntable
generator has the properties SURVEY 8(d) asks for.  No GPU compute here."""
import os
import re

import numpy as np

from conftest import ROOT, rand_cloud


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mvr_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mvr_[a-z0-9_]+)\s*\(", txt)))


def test_cabi_exports_every_declared_symbol(mvr):
    import ctypes
    lib = ctypes.CDLL(mvr.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), "libmvr_hip.so does not export " + s
    assert set(mvr.SIGNATURES) == set(syms)        # the Python view binds exactly the header


def test_no_oracle_in_product():
    """The product must never route through the oracle or a CPU fallback."""
    bad = []
    for base in (os.path.join(ROOT, "multi-view-registration_amd"), os.path.join(ROOT, "include")):
        for dp, _, fs in os.walk(base):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")):
                    src = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(import|from)\s+oracle|mvr_oracle\.h|liboracle|orc_[a-z]", src, re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_ctx_create_fails_loudly_without_gpu(mvr):
    import ctypes as C
    h = C.c_void_p()
    rc = mvr._lib.mvr_ctx_create(C.byref(h), 0)
    if rc == 0:                      # a GPU is present (GPU box): fine
        mvr._lib.mvr_ctx_destroy(h)
    else:
        assert rc == mvr.E_HIP and not h.value
        try:
            mvr.Context(0)
            assert False, "Context() must raise without a GPU"
        except mvr.MvrError as e:
            assert e.status == mvr.E_HIP
    assert mvr._lib.mvr_strerror(mvr.E_NOCORR).decode().startswith("not enough")


def test_entry_points_reject_null_arguments_before_touching_the_gpu(mvr):
    """argument errors are reported as MVR_E_ARG, not as a crash -- checked without a device for the calls a binder
    is most likely to get wrong (the one-call ring step, the batch calls)"""
    import ctypes as C
    L, NULL = mvr._lib, None
    i2 = (C.c_int * 2)(0, 1)
    d3, d32 = (C.c_double * 3)(), (C.c_double * 32)()
    assert L.mvr_ring_step(NULL, 2, i2, i2, 1, i2, i2, 4.0, 1, 0, d3, 16, d32, d32, NULL, NULL, NULL, NULL, NULL, NULL) == mvr.E_ARG
    assert L.mvr_pair_moments2_batch(NULL, 1, i2, i2, 4.0, 1, 0, NULL, NULL, d3, NULL, d32) == mvr.E_ARG
    assert L.mvr_cloud_transform_batch(NULL, 1, i2, i2, d32) == mvr.E_ARG
    assert L.mvr_ctx_tune(NULL, b"pair_groups", 2) == mvr.E_ARG
    # the host-only step validates its graph: an edge that names a view outside [0, n)
    es, et = (C.c_int * 1)(0), (C.c_int * 1)(5)
    poses = (C.c_double * 32)(); lum = (C.c_double * 12)(); pn = (C.c_double * 1)(); pm = (C.c_double * 1)(); its = C.c_int()
    rc = L.mvr_ring_host_step(2, 1, es, et, d32, d3, 16, poses, lum, NULL, pn, pm, C.byref(its))
    assert rc != 0


def _moments_numpy(src, tgt, q, m, origin):
    p = src[q, :3].astype(np.float64) - origin
    t = tgt[m, :3].astype(np.float64) - origin
    iu = np.triu_indices(3)
    row = np.concatenate([[len(q)], origin, p.sum(0), t.sum(0), (p.T @ p)[iu], (t.T @ t)[iu], (p.T @ t).ravel(), [0]])
    return row


def test_umeyama_from_moments_matches_oracle(mvr, orc):
    rng = np.random.default_rng(20)
    for _ in range(20):
        A = rng.standard_normal((3, 3)) * rng.uniform(0.1, 100)
        ms, mt = rng.standard_normal(3) * 100, rng.standard_normal(3) * 100
        pm = mvr.PairMoments()
        pm.n = 100
        pm.mean_src[:] = ms; pm.mean_tgt[:] = mt; pm.sigma[:] = A.ravel()
        T, sv = mvr.umeyama_from_moments(pm)
        To, svo = orc.umeyama_from_moments(ms, mt, A)
        assert np.abs(T - To).max() <= 1e-6 * max(1, np.abs(To).max()) and np.allclose(sv, svo, rtol=1e-12, atol=1e-12)
        R = T[:3, :3].astype(np.float64)
        assert abs(np.linalg.det(R) - 1) < 1e-5
    pm.n = 2
    assert mvr.umeyama_from_moments(pm)[0] is None           # MVR_E_NOCORR


def test_moments2_to_moments_and_lum_edge(mvr, orc):
    rng = np.random.default_rng(21)
    src, tgt = rand_cloud(rng, 600, scale=40), rand_cloud(rng, 600, scale=40)
    tgt[:, :3] = src[:, :3] + rng.standard_normal((600, 3)).astype(np.float32) * 0.3
    q = np.sort(rng.permutation(600)[:450]); m = q.copy()
    origin = np.array([0.0, 0.0, 900.0])
    m2 = mvr.moments2_from_row(_moments_numpy(src, tgt, q, m, origin))
    corr = np.zeros(len(q), orc.CORR_DTYPE); corr["query"], corr["match"] = q, m
    corr["dist2"] = ((src[q, :3] - tgt[m, :3]) ** 2).sum(1)
    # (a) centred moments + Umeyama
    pm = mvr.moments_from_moments2(m2)
    T, _ = mvr.umeyama_from_moments(pm)
    To, mom = orc.umeyama(src, tgt, corr)
    assert np.abs(T[:3, :3] - To[:3, :3]).max() < 1e-6 and np.abs(T[:3, 3] - To[:3, 3]).max() < 1e-4
    assert np.allclose(np.array(pm.sigma), mom[8:17], rtol=1e-9, atol=1e-9)
    assert abs(pm.mse - mom[7]) < 1e-6
    # (b) LUM::computeEdge from moments == direct sums, for non-trivial poses
    for ps, pt in [(np.zeros(6), np.zeros(6)),
                   (np.array([0.3, -0.2, 0.1, 0.01, -0.02, 0.03]), np.array([-0.1, 0.2, 0.05, -0.02, 0.01, 0.015]))]:
        n, MM, MZ, ss = orc.lum_edge(src, tgt, corr, ps, pt)
        rc, MMg, MZg, ssg = mvr.lum_edge_from_moments(m2, ps, pt)
        assert rc == 0 and n == len(q)
        assert np.allclose(MMg, MM, rtol=1e-10, atol=1e-6)
        assert np.allclose(MZg, MZ, rtol=1e-9, atol=1e-6)
        assert abs(ssg - ss) <= 1e-7 * ss


_LUM_BAND_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
import importlib
mvr = importlib.import_module("multi-view-registration_amd")
out = []
for V, chain in ((3, False), (4, False), (12, False), (12, True), (36, False)):
    rng = np.random.default_rng(V)
    origin = np.array([-13.4, 50.2, 917.5])
    edges = [(i, (i + 1) %% V) for i in range(V if not chain else V - 1)]
    m2 = []
    for e, (s_, t_) in enumerate(edges):
        p = rng.normal(size=(3000, 3)) * 40 + origin
        a = 0.002 * (e + 1)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        q = (p - origin) @ R.T + origin + rng.normal(size=p.shape) * 0.1 + 0.01 * e
        pc, qc = p - origin, q - origin
        row = np.zeros(32); row[0] = len(p); row[1:4] = origin; row[4:7] = pc.sum(0); row[7:10] = qc.sum(0)
        sym = lambda m: [m[0, 0], m[0, 1], m[0, 2], m[1, 1], m[1, 2], m[2, 2]]
        row[10:16] = sym(pc.T @ pc); row[16:22] = sym(qc.T @ qc); row[22:31] = (pc.T @ qc).ravel()
        m2.append(mvr.moments2_from_row(row))
    rc, P, its = mvr.lum_compute(V, edges, m2, max_iterations=16)
    out.append("%%d %%d %%d %%s %%s" %% (V, rc, its, hashlib.sha256(np.asarray(P).tobytes()).hexdigest(), np.asarray(P, np.float64).tobytes().hex()))
print("\n".join(out))
"""


def test_lum_routes_agree():
    """rings and chains of views: mvr_lum_compute solves the block-tridiagonal normal equations by a block Cholesky
    (solve_chain6); MVR_LUM_BAND=1 forces the row-wise elimination on band storage (solve_spd_band), MVR_LUM_DENSE=1 the
    dense matrix and solve_spd.  The two row-wise routes perform the same operations in the same order: the same BYTES.
    The block route orders the updates differently: the same poses to rounding (1e-9 of their size after 16 iterations)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = []
    for var in (None, "MVR_LUM_BAND", "MVR_LUM_DENSE"):
        env = dict(os.environ)
        env.pop("MVR_LUM_DENSE", None); env.pop("MVR_LUM_BAND", None)
        if var:
            env[var] = "1"
        r = subprocess.run([sys.executable, "-c", _LUM_BAND_SCRIPT % root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append([ln.split() for ln in r.stdout.strip().splitlines()])
    assert len(runs[0]) == 5
    for blk, band, dense in zip(*runs):
        assert blk[1] == band[1] == dense[1] == "0", (blk[:3], band[:3], dense[:3])
        assert band[3] == dense[3], band[0]                                   # byte for byte
        pb = np.frombuffer(bytes.fromhex(blk[4]), np.float64); pr = np.frombuffer(bytes.fromhex(band[4]), np.float64)
        assert np.abs(pb - pr).max() <= 1e-9 * max(1.0, np.abs(pr).max()), (blk[0], np.abs(pb - pr).max())


def test_lum_view_without_correspondences_is_singular_on_every_route():
    """a view whose two edges carry no usable correspondences has an all-zero block row: the block route's pivot test
    refuses it, the row-wise routes refuse it, the pivoted elimination finds the zero column -- MVR_E_SINGULAR whichever
    route was asked for, poses untouched."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, importlib
import numpy as np
sys.path.insert(0, %r)
mvr = importlib.import_module("multi-view-registration_amd")
V = 6
rng = np.random.default_rng(5); origin = np.array([-13.4, 50.2, 917.5])
edges = [(i, (i + 1) %% V) for i in range(V)]
m2 = []
for e in range(V):
    n = 3000 if e not in (2, 3) else 2
    p = rng.normal(size=(n, 3)) * 40 + origin
    q = p + rng.normal(size=p.shape) * 0.1
    pc, qc = p - origin, q - origin
    row = np.zeros(32); row[0] = n; row[1:4] = origin; row[4:7] = pc.sum(0); row[7:10] = qc.sum(0)
    sym = lambda m: [m[0, 0], m[0, 1], m[0, 2], m[1, 1], m[1, 2], m[2, 2]]
    row[10:16] = sym(pc.T @ pc); row[16:22] = sym(qc.T @ qc); row[22:31] = (pc.T @ qc).ravel()
    m2.append(mvr.moments2_from_row(row))
rc, P, its = mvr.lum_compute(V, edges, m2, max_iterations=4)
print(rc, its, float(np.abs(np.asarray(P)).max()))
""" % root
    for var in (None, "MVR_LUM_BAND", "MVR_LUM_DENSE"):
        env = dict(os.environ)
        env.pop("MVR_LUM_DENSE", None); env.pop("MVR_LUM_BAND", None)
        if var:
            env[var] = "1"
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        rc, its, pmax = r.stdout.split()
        assert int(rc) == -5 and int(its) == 0 and float(pmax) == 0.0, (var, r.stdout)


def test_lum_edge_four_at_once_is_bit_identical_to_the_scalar_function(mvr):
    """mvr_lum_compute sends its edges four at a time through an AVX2 pass (one edge per lane): every lane must give
    the bytes of the scalar LUM::computeEdge -- and a group with a degenerate edge must be refused (-> scalar path)."""
    rng = np.random.default_rng(77)
    origin = np.array([-13.4, 50.2, 917.5])
    for trial in range(25):
        m2s, ps, pt = [], [], []
        for l in range(4):
            n = int(rng.integers(3, 4000))
            src = rand_cloud(rng, n, scale=float(rng.uniform(5, 120)), centre=tuple(origin + rng.standard_normal(3) * 30))
            tgt = src.copy(); tgt[:, :3] += rng.standard_normal((n, 3)).astype(np.float32) * float(rng.uniform(0.01, 2.0))
            idx = np.arange(n)
            m2s.append(mvr.moments2_from_row(_moments_numpy(src, tgt, idx, idx, origin)))
            ps.append(np.r_[rng.standard_normal(3) * 2.0, rng.standard_normal(3) * 0.05])
            pt.append(np.r_[rng.standard_normal(3) * 2.0, rng.standard_normal(3) * 0.05])
        rc, MM, MZ, ss = mvr.lum_edge_from_moments_x4(m2s, ps, pt)
        assert rc == 0
        for l in range(4):
            rc1, MM1, MZ1, ss1 = mvr.lum_edge_from_moments(m2s[l], ps[l], pt[l])
            assert rc1 == 0
            assert MM[l].tobytes() == MM1.tobytes() and MZ[l].tobytes() == MZ1.tobytes()
            assert np.float64(ss[l]).tobytes() == np.float64(ss1).tobytes()
    # an edge with two pairs only: the group is refused, nothing written
    few = mvr.moments2_from_row(_moments_numpy(src, tgt, np.arange(2), np.arange(2), origin))
    rc, _, _, _ = mvr.lum_edge_from_moments_x4([m2s[0], few, m2s[2], m2s[3]], ps, pt)
    assert rc == mvr.E_NOCORR
    # collinear points: MM is singular, the Cholesky pivot test refuses the group
    line = np.zeros((50, 4), np.float32); line[:, 0] = np.arange(50)
    flat = mvr.moments2_from_row(_moments_numpy(line, line, np.arange(50), np.arange(50), np.zeros(3)))
    rc, _, _, _ = mvr.lum_edge_from_moments_x4([flat, m2s[1], m2s[2], m2s[3]], ps, pt)
    assert rc == mvr.E_NOCORR


def test_lum_compute_from_moments_matches_oracle(mvr, orc):
    rng = np.random.default_rng(22)
    base = rand_cloud(rng, 500, scale=40, centre=(0, 0, 900))
    offs = [np.zeros(6)] + [np.r_[rng.standard_normal(3) * 0.4, rng.standard_normal(3) * 0.01] for _ in range(3)]
    clouds = []
    for o in offs:
        Ti = np.linalg.inv(orc.pose_to_mat4(o))
        c = base.copy(); c[:, :3] = (base[:, :3].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        c[:, :3] += rng.standard_normal((500, 3)).astype(np.float32) * 0.05
        clouds.append(c)
    edges = [(0, 1), (1, 2), (2, 3), (3, 0)]
    idx = np.arange(500)
    corr = np.zeros(500, orc.CORR_DTYPE); corr["query"] = corr["match"] = idx
    origin = np.array([0.0, 0.0, 900.0])
    m2 = [mvr.moments2_from_row(_moments_numpy(clouds[s], clouds[t], idx, idx, origin)) for s, t in edges]
    Po, ito = orc.lum_compute(clouds, edges, [corr] * 4, max_iterations=16)
    rc, Pg, itg = mvr.lum_compute(4, edges, m2, max_iterations=16)
    assert rc == 0 and itg == ito
    assert np.abs(Pg - Po).max() < 1e-7
    for v in range(1, 4):       # and they do realign the clouds
        Tv = mvr.pose_to_mat4(Pg[v])
        moved = clouds[v][:, :3].astype(np.float64) @ Tv[:3, :3].T + Tv[:3, 3]
        assert np.abs(moved - base[:, :3]).max() < 0.5


def test_product_lum_linearisation_is_the_jacobian_of_the_pose_map(mvr, orc):
    """The product's own LUM pieces (mvr_lum_edge_from_moments, mvr_lum_incidence) against a numeric Jacobian of the
    pose map -- no oracle involved (tests/lum_kat.py) -- and then equal to the oracle's."""
    import lum_kat

    def edge(src, tgt):
        idx = np.arange(len(src))
        m2 = mvr.moments2_from_row(_moments_numpy(src, tgt, idx, idx, np.array([0.0, 0.0, 900.0])))
        rc, MM, MZ, ss = mvr.lum_edge_from_moments(m2, np.zeros(6), np.zeros(6))
        assert rc == 0
        return MM, MZ

    rng = np.random.default_rng(78)
    assert lum_kat.check(edge, mvr.lum_incidence, rng) < 1e-3
    for _ in range(10):
        X = np.concatenate([rng.normal(0, 5, 3), rng.normal(0, 0.3, 3)])
        assert np.array_equal(mvr.lum_incidence(X), orc.lum_incidence(X))


def test_refine_axis_matches_oracle(mvr, orc):
    """mvr_refine_axis (streaming Givens QR) == the oracle's restatement of Registrator::refineAxis
    (registrator.cpp:402-455, Householder QR as LAPACK dgels) on exact and on perturbed turntable poses, 12 and 36
    views, any subset of registered views."""
    rng = np.random.default_rng(79)
    true_axis = np.array([-0.054323, -0.814921, -0.577020]); true_axis /= np.linalg.norm(true_axis)
    true_pivot = np.array([-13.382786, 50.223461, 917.4776])
    for V in (12, 36):
        poses = [orc.axis_rotation(true_pivot, true_axis, orc.turntable_angle(v, V)) for v in range(1, V)]
        noisy = []
        for P in poses:
            Q = P.copy(); Q[:3, 3] += rng.normal(0, 0.05, 3)
            noisy.append(orc.mat4d_mul(orc.axis_rotation(true_pivot, rng.normal(0, 1, 3), rng.normal(0, 3e-4)), Q))
        for ps in (poses, noisy, noisy[:3], noisy[4:5]):
            rc, ax, pv = mvr.refine_axis(ps, true_pivot[1])
            rco, axo, pvo = orc.refine_axis(ps, true_pivot[1])
            assert rc == 0 and rco == 0
            assert np.abs(ax - axo).max() <= 2e-7 and np.abs(pv - pvo).max() <= 1e-4 * max(1.0, np.abs(pvo).max() / 1000), (ax, axo, pv, pvo)
        rc, ax, pv = mvr.refine_axis(poses, true_pivot[1])
        assert abs(abs(ax.astype(np.float64) @ true_axis) - 1) < 1e-6 and np.abs(pv - true_pivot).max() < 1e-3
    assert mvr.refine_axis([], 0.0)[0] == mvr.E_ARG
    # identity poses: every direction is fixed -> rank deficient for the pivot (the regulariser rows alone)
    assert mvr.refine_axis([np.eye(4)] * 3, 0.0)[0] == mvr.E_SINGULAR and orc.refine_axis([np.eye(4)] * 3, 0.0)[0] == -1


def test_host_helpers_match_oracle(mvr, orc):
    rng = np.random.default_rng(23)
    for v in range(12):
        assert mvr.turntable_angle(v, 12) == orc.turntable_angle(v, 12)
    for v in range(36):
        assert mvr.turntable_angle(v, 36) == orc.turntable_angle(v, 36)
    piv, ax = rng.standard_normal(3) * 100, rng.standard_normal(3)
    assert np.array_equal(mvr.axis_rotation(piv, ax, 0.7), orc.axis_rotation(piv, ax, 0.7))
    A, B = rng.standard_normal((4, 4)), rng.standard_normal((4, 4))
    assert np.array_equal(mvr.mat4d_mul(A, B), orc.mat4d_mul(A, B))
    assert np.array_equal(mvr.mat4f_mul(A, B), orc.mat4f_mul(A, B))
    pose = rng.standard_normal(6) * 0.1
    assert np.allclose(mvr.pose_to_mat4(pose), orc.pose_to_mat4(pose), atol=1e-15)


def test_synth_generator(mvr):
    sp = mvr.synth_params(12, 1)
    a1 = mvr.synth_view(sp, 3, 2000)
    a2, nrm = mvr.synth_view(sp, 3, 2000, normals=True)
    assert np.array_equal(a1, a2)                                  # deterministic
    assert not np.array_equal(a1, mvr.synth_view(sp, 4, 2000))
    assert np.all(a1[:, 3] == 1) and np.all(np.isfinite(a1))
    piv = np.array(sp.pivot)
    r = np.linalg.norm(a1[:, :3] - piv, axis=1)
    assert 50 < r.min() and r.max() < 110                          # star-shaped, r ~ 80 mm
    assert np.all((nrm[:, :3] * a1[:, :3]).sum(1) < 0)             # sensor-facing only
    assert np.allclose(np.linalg.norm(nrm[:, :3], axis=1), 1, atol=1e-5)
    # view v is view 0's object rotated by +v*30deg about (pivot, axis): undoing the
    # TRUE rotation puts both scans on one surface -> small NN distances
    from scipy.spatial import cKDTree
    v0 = mvr.synth_view(sp, 0, 20000)
    back = mvr.axis_rotation(piv, np.array(sp.axis), mvr.turntable_angle(1, 12))
    v1 = mvr.synth_view(sp, 1, 2000)
    moved = v1[:, :3].astype(np.float64) @ back[:3, :3].T + back[:3, 3]
    d, _ = cKDTree(v0[:, :3]).query(moved)
    assert np.median(d) < 1.5 and np.quantile(d, 0.6) < 2.0        # overlap region is dense
    # the prior handed to initRotation is deliberately mis-calibrated
    pp, pa = mvr.synth_prior(sp)
    assert np.allclose(pp - piv, [1.5, -1.0, 2.0])
    ax = np.array(sp.axis) / np.linalg.norm(sp.axis)
    assert abs(np.degrees(np.arccos(np.clip(pa @ ax, -1, 1))) - 0.5) < 0.2


def test_ring_segments_partition(mvr):
    """mvr_ring_segments (the native multi-GPU host's sharding) == ring.split_queries, covers every query exactly once,
    contiguous and balanced to within one query, for ragged scan sizes and more ranks than edges."""
    import importlib
    ring = importlib.import_module("multi-view-registration_amd.ring")
    rng = np.random.default_rng(31)
    for sizes in ([200000] * 12, [1000000] * 36, list(rng.integers(1, 5000, 7)), [5, 0, 3], [1]):
        total = int(sum(sizes))
        for world in (1, 2, 3, 8, 16):
            seen, per_rank = [], []
            for rank in range(world):
                segs = mvr.ring_segments(sizes, world, rank)
                assert segs == [tuple(int(x) for x in sg) for sg in ring.split_queries(sizes, world, rank)]
                per_rank.append(sum(c for _, _, c in segs))
                for e, b, c in segs:
                    assert c > 0 and b + c <= sizes[e]
                    seen += [(e, q) for q in range(b, b + c)] if total < 100000 else []
                es = [e for e, _, _ in segs]
                assert es == sorted(es) and (not es or es == list(range(es[0], es[-1] + 1)) or 0 in sizes)
            assert sum(per_rank) == total and max(per_rank) - min(per_rank) <= 1
            if total < 100000:
                assert sorted(seen) == [(e, q) for e, n in enumerate(sizes) for q in range(n)]
