"""GPU tests of the C++ layer (include/mvr/*.hpp) through tests/cxx/shim_driver:
  * the reference's own driver loops (Registrator::registrationICP / registrationLUM / computeError /
    automaticRegistration, mvr/src/registrator.cpp), replayed call for call on the PCL-named shim classes
    (tests/cxx/call_surface.hpp -- call-site compatibility), and
  * the product's device-resident drivers (mvr::Registrator::registrationICPDevice / registrationLUMDevice /
    computeErrorDevice / registration),
both compared with the same loops restated on the CPU oracle (tests/ref_driver.py) on identical synthetic scans."""
import json
import os
import subprocess

import numpy as np
import pytest

import ref_driver
from conftest import ROOT

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-5, 1e-4


@pytest.fixture(scope="module")
def driver(built):
    exe = built.build_cxx_tests()
    assert exe and os.path.exists(exe)
    return exe


@pytest.fixture(scope="module", params=["1", "0"], ids=["culled", "brute"], autouse=True)
def nn_mode_env(request):
    """Every driver test runs on both exact search kernels (MVR_NN_MODE)."""
    old = os.environ.get("MVR_NN_MODE")
    os.environ["MVR_NN_MODE"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("MVR_NN_MODE", None)
    else:
        os.environ["MVR_NN_MODE"] = old


def run(exe, mode, V, N, max_d, repeat, config, *extra):
    env = dict(os.environ)
    env.pop("MVR_RCCL_LIB", None)        # (the Python package points it at torch's copy; a C++ process uses the system's)
    r = subprocess.run([exe, mode, str(V), str(N), str(max_d), str(repeat), str(config)] + [str(e) for e in extra], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout), r.stderr


def scene(mvr, orc, V, N, config):
    sp = mvr.synth_params(V, config)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    # the driver stores pivot/axis as osg::Vec3 (float)
    piv, ax = piv.astype(np.float32).astype(np.float64), ax.astype(np.float32).astype(np.float64)
    return sp, scans, ref_driver.init_poses(orc, V, piv, ax)


def assert_poses(got, exp, rot=ROT_TOL, trans=TRANS_TOL):
    for v, (g, e) in enumerate(zip(got, exp)):
        g = np.array(g).reshape(4, 4)
        assert np.abs(g[:3, :3] - e[:3, :3]).max() <= rot, (v, np.abs(g[:3, :3] - e[:3, :3]).max())
        assert np.abs(g[:3, 3] - e[:3, 3]).max() <= trans, (v, np.abs(g[:3, 3] - e[:3, 3]).max())


def test_registration_icp_driver(driver, mvr, orc):
    """registrator.cpp:517-588 through the shim == the oracle-driven restatement."""
    V, N, max_d, repeat = 12, 3000, 8.0, 2
    out, _ = run(driver, "seq", V, N, max_d, repeat, 3)
    sp, scans, poses0 = scene(mvr, orc, V, N, 3)
    poses, log = ref_driver.sequential_icp(orc, scans, poses0, orc.make_params(max_dist=max_d, max_iter=1000), V, repeat=repeat)
    assert [e["view"] for e in out["log"]] == [e["view"] for e in log] == ref_driver.view_order(V) * repeat
    for g, e in zip(out["log"], log):
        assert g["iterations"] == e["iterations"] == 1
        assert g["n_corr"] == e["n_corr"] and abs(g["mse"] - e["mse"]) < 1e-9
        T = np.array(g["T"]).reshape(4, 4)
        assert np.abs(T[:3, :3] - e["T"][:3, :3]).max() <= ROT_TOL and np.abs(T[:3, 3] - e["T"][:3, 3]).max() <= TRANS_TOL
    fit = [e for e in log if "fitness" in e]
    gfit = [g["fitness"] for g in out["log"] if g["fitness"] is not None]
    assert len(gfit) == len(fit) == repeat and np.allclose(gfit, [e["fitness"] for e in fit], atol=1e-4)
    assert_poses(out["poses"], poses)
    # refineAxis (registrator.cpp:402-455) moves the mis-calibrated prior towards the true axis
    true_ax = np.array(sp.axis) / np.linalg.norm(sp.axis)
    prior_ax = mvr.synth_prior(sp)[1]
    ref_ax = np.array(out["refined_axis"])
    ang = lambda a: np.degrees(np.arccos(min(1.0, abs(a @ true_ax))))
    assert ang(ref_ax) < ang(prior_ax)


def test_registration_lum_driver(driver, mvr, orc):
    """registrator.cpp:611-664 through the shim == oracle LUM pass."""
    V, N, max_d = 12, 3000, 8.0
    out, _ = run(driver, "lum", V, N, max_d, 1, 3)
    sp, scans, poses0 = scene(mvr, orc, V, N, 3)
    new, P, corrs, its = ref_driver.lum_pass(orc, scans, poses0, max_d, 16)
    assert out["lum_ncorr"] == [len(c) for c in corrs]
    assert_poses(out["poses"], new)


def test_registration_lum_device_resident(driver, mvr, orc):
    """Registrator::registrationLUMDevice (batched pose, concurrent pairs, LUM from moments; nothing but
    V x 31 doubles reaches the host) == the oracle LUM pass == the PCL-style registrationLUM."""
    V, N, max_d = 12, 3000, 8.0
    out, _ = run(driver, "lumdev", V, N, max_d, 1, 3)
    ref, _ = run(driver, "lum", V, N, max_d, 1, 3)
    sp, scans, poses0 = scene(mvr, orc, V, N, 3)
    new, P, corrs, its = ref_driver.lum_pass(orc, scans, poses0, max_d, 16)
    assert out["lum_ncorr"] == [len(c) for c in corrs] == ref["lum_ncorr"]
    assert_poses(out["poses"], new)
    assert_poses(out["poses"], [np.array(p).reshape(4, 4) for p in ref["poses"]])


def test_scan_cloud_denoise(driver, mvr, orc):
    """ScanCloud::denoise (point_cloud.cpp:423-465 through the shim) == the oracle on the same cloud."""
    V, N = 2, 6000
    out, _ = run(driver, "denoise", V, N, 4.0, 1, 3)
    sp = mvr.synth_params(V, 3)
    scan = mvr.synth_view(sp, 0, N)
    extra = np.ones((N // 40, 4), np.float32)
    k = np.arange(N // 40, dtype=np.float32)
    extra[:, 0] = np.float32(1000.0) + np.float32(7.0) * k; extra[:, 1] = -500.0; extra[:, 2] = np.float32(2000.0) + np.float32(3.0) * (k.astype(np.int64) % 11).astype(np.float32)
    pts = np.concatenate([scan, extra])
    keep, _, _ = orc.denoise(pts, 10, 2.5)
    assert out["kept"] == len(keep) and out["noise"] == len(pts) - len(keep) and out["noise"] >= N // 40
    assert np.allclose(out["sum"], pts[keep, :3].astype(np.float64).sum(0), rtol=1e-12)


def test_compute_error_pairs(driver, mvr, orc):
    """registrator.cpp:466-515: ring pairs + (0, V-1), reciprocal correspondences."""
    V, N, max_d = 12, 2500, 6.0
    out, _ = run(driver, "err", V, N, max_d, 1, 5)
    sp, scans, poses0 = scene(mvr, orc, V, N, 5)
    clouds = [orc.transform_f64(poses0[v], scans[v]) for v in range(V)]
    exp_pairs = [(i, i + 1) for i in range(V - 1)] + [(0, V - 1)]
    assert [(p[0], p[1]) for p in out["pairs"]] == exp_pairs
    for (s, t), p in zip(exp_pairs, out["pairs"]):
        c = orc.correspondences(clouds[s], clouds[t], max_d)
        assert p[2] == len(c) and abs(p[3] - float(c["dist2"].astype(np.float64).sum())) < 1e-6 * max(1.0, p[3])


def test_automatic_registration_driver(driver, mvr, orc):
    """The intent of automaticRegistration (:746-842): incremental add-a-view
    with `repeat` aliased in-place aligns per view."""
    V, N, max_d, repeat = 6, 3000, 8.0, 3
    out, _ = run(driver, "auto", V, N, max_d, repeat, 6)
    sp, scans, poses0 = scene(mvr, orc, V, N, 6)
    poses = [p.copy() for p in poses0]
    target = orc.transform_f64(poses[0], scans[0])
    params = orc.make_params(max_dist=max_d, max_iter=1000, teps=0.0, feps=50.0)    # transformation eps never set (App. C.3)
    k = 0
    for v in range(1, V):
        source = orc.transform_f64(poses[v], scans[v])
        for _ in range(repeat):
            source, T, st, rc = orc.icp_align(source, target, params)              # aliased: source advances in place
            poses[v] = orc.mat4d_mul(T.astype(np.float64), poses[v])
            g = out["log"][k]; k += 1
            assert g["view"] == v and g["n_corr"] == st["n_corr"]
        target = np.concatenate([target, source])
    assert k == len(out["log"])
    assert_poses(out["poses"], poses)


def test_automatic_registration_device_resident(driver, mvr, orc):
    """Registrator::automaticRegistrationDevice (scans resident, the model grown in a device slot, every repeat an in-place
    align: one 4x4 back) == the same restatement: counts per repeat, final poses, and the logged TRUE residual equals the
    oracle's fitness of the advanced source against the model."""
    V, N, max_d, repeat = 12, 3000, 8.0, 3
    out, _ = run(driver, "autodev", V, N, max_d, repeat, 6)
    sp, scans, poses0 = scene(mvr, orc, V, N, 6)
    poses = [p.copy() for p in poses0]
    target = orc.transform_f64(poses[0], scans[0])
    params = orc.make_params(max_dist=max_d, max_iter=1000, teps=0.0, feps=50.0)
    k = 0
    for v in range(1, V):
        source = orc.transform_f64(poses[v], scans[v])
        for _ in range(repeat):
            source, T, st, rc = orc.icp_align(source, target, params)
            poses[v] = orc.mat4d_mul(T.astype(np.float64), poses[v])
            g = out["log"][k]; k += 1
            assert g["view"] == v and g["n_corr"] == st["n_corr"] and g["iterations"] == st["iterations"]
            Tg = np.array(g["T"]).reshape(4, 4)
            assert np.abs(Tg[:3, :3] - T[:3, :3]).max() <= ROT_TOL and np.abs(Tg[:3, 3] - T[:3, 3]).max() <= TRANS_TOL
            if v in (1, V - 1):
                assert abs(g["fitness"] - orc.fitness(source, target, np.eye(4, dtype=np.float32))) < 1e-6 * max(1.0, g["fitness"])
        target = np.concatenate([target, source])
    assert k == len(out["log"]) == (V - 1) * repeat
    assert_poses(out["poses"], poses)


def test_pcl_named_api_surface(driver):
    out, err = run(driver, "api", 12, 3000, 8.0, 1, 3)
    assert out["iters"] == 7 and out["converged"] == 1
    assert out["nocorr_converged"] == 0 and out["nocorr_identity"] == 1
    assert "Not enough correspondences" in err
    assert out["n_oneway"] > out["n_recip"] > 100
    assert np.isfinite(out["fitness_after_alias"])


def test_registration_icp_device_resident(driver, mvr, orc):
    """Registrator::registrationICPDevice (scans uploaded once, target grown in a device slot, one 4x4 per align
    back) == the oracle-driven restatement of registrationICP (registrator.cpp:517-588), align by align."""
    V, N, max_d, repeat = 12, 3000, 8.0, 2
    out, _ = run(driver, "seqdev", V, N, max_d, repeat, 3)
    sp, scans, poses0 = scene(mvr, orc, V, N, 3)
    poses, log = ref_driver.sequential_icp(orc, scans, poses0, orc.make_params(max_dist=max_d, max_iter=1000), V, repeat=repeat)
    assert [e["view"] for e in out["log"]] == [e["view"] for e in log]
    for g, e in zip(out["log"], log):
        assert g["iterations"] == e["iterations"] == 1 and g["n_corr"] == e["n_corr"] and abs(g["mse"] - e["mse"]) < 1e-9
        T = np.array(g["T"]).reshape(4, 4)
        assert np.abs(T[:3, :3] - e["T"][:3, :3]).max() <= ROT_TOL and np.abs(T[:3, 3] - e["T"][:3, 3]).max() <= TRANS_TOL
    gfit = [g["fitness"] for g in out["log"] if g["fitness"] is not None]
    assert np.allclose(gfit, [e["fitness"] for e in log if "fitness" in e], atol=1e-6)
    assert_poses(out["poses"], poses)
    # the refined axis (mvr_refine_axis) equals the oracle's refineAxis on the final poses
    piv32 = np.float32(mvr.synth_prior(sp)[0][1])
    rc, ax, pv = orc.refine_axis(poses[1:], piv32)
    assert rc == 0 and np.abs(np.array(out["refined_axis"]) - ax).max() < 1e-6 and np.abs(np.array(out["refined_pivot"]) - pv).max() < 2e-3


def test_compute_error_device_resident(driver, mvr, orc):
    """Registrator::computeErrorDevice: per ring pair (+ (0, V-1)) the count and the residual sum of the reciprocal
    correspondences -- equal to the oracle's lists (registrator.cpp:466-515)."""
    V, N, max_d = 12, 2500, 6.0
    out, _ = run(driver, "errdev", V, N, max_d, 1, 5)
    sp, scans, poses0 = scene(mvr, orc, V, N, 5)
    clouds = [orc.transform_f64(poses0[v], scans[v]) for v in range(V)]
    exp_pairs = [(i, i + 1) for i in range(V - 1)] + [(0, V - 1)]
    assert [(p[0], p[1]) for p in out["pairs"]] == exp_pairs
    for (s, t), p in zip(exp_pairs, out["pairs"]):
        c = orc.correspondences(clouds[s], clouds[t], max_d)
        assert p[2] == len(c) and abs(p[3] - float(c["dist2"].astype(np.float64).sum())) < 1e-9 * max(1.0, p[3])


def test_registration_writes_the_merged_cloud(driver, mvr, orc, tmp_path):
    """Registrator::registration (registrator.cpp:719-744): every view denoised, posed by its prior, merged into
    points.pcd / points.asc (saveRegisteredPoints, :344-400)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pcd_py
    V, N = 4, 5000
    out, _ = run(driver, "register", V, N, 4.0, 1, 3, tmp_path)
    sp, scans, poses0 = scene(mvr, orc, V, N, 3)
    keeps = [orc.denoise(scans[v], 10, 2.5)[0] for v in range(V)]
    assert out["sizes"] == [len(k) for k in keeps] and out["merged"] == sum(out["sizes"]) == out["reloaded"] and out["ok"] == 1
    hdr, rec = pcd_py.read_pcd(os.path.join(tmp_path, "points.pcd"))
    assert len(rec["x"]) == out["merged"]
    off = 0
    for v in range(V):
        exp = orc.transform_f64(poses0[v], scans[v][keeps[v]])
        got = np.stack([rec["x"][off:off + len(exp)], rec["y"][off:off + len(exp)], rec["z"][off:off + len(exp)]], 1)
        assert np.array_equal(got.astype(np.float32).view(np.uint32), np.ascontiguousarray(exp[:, :3]).view(np.uint32)), v
        rgb = rec["rgb"].view(np.uint32)[off:off + len(exp)]
        assert np.all(rgb == ((v << 16) | ((2 * v) << 8) | (255 - v)))
        off += len(exp)
    assert len(open(os.path.join(tmp_path, "points.asc")).read().strip().split("\n")) == out["merged"]


def test_world_host_from_cxx_without_torch(driver, mvr, orc):
    """mvr_world_create / mvr_world_ring_run from a plain C++ process (system HIP + system RCCL, no torch anywhere):
    a world of one reproduces registrationLUMDevice exactly, and RCCL really carried the table (1 rank reported)."""
    V, N, max_d = 12, 3000, 8.0
    out, _ = run(driver, "world", V, N, max_d, 2, 3)
    ref, _ = run(driver, "lumdev", V, N, max_d, 2, 3)
    assert out["comm"] == [0, 1, 1] and "rccl" in out["rccl"]
    assert out["lum_ncorr"] == ref["lum_ncorr"]
    assert out["poses"] == ref["poses"]                      # printed with %.17g: bit-identical
