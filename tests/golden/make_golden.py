"""Generates the golden fixtures under tests/golden/ (committed as data).

The reference (fanxiaochen/Multi-View-Registration) ships no tests, fixtures or
sample data for this path and cannot be built here (PCL/Eigen/FLANN/Qt/OSG
absent) -- SURVEY.md 8(c), "parity unpinned".  These vectors are therefore
inputs from the synthetic turntable generator and expected outputs
from the CPU oracle (oracle/mvr_oracle.c), the pinned restatement of the PCL
semantics of SURVEY App. A.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc  # noqa: E402
import ref_driver  # noqa: E402

mvr = importlib.import_module("multi-view-registration_amd")   # host-only use: the generator
OUT = os.path.dirname(os.path.abspath(__file__))


def pair_fixture():
    """BASELINE config 1: 2 synthetic turntable scans, 10 000 points each, pairwise point-to-point ICP."""
    sp = mvr.synth_params(12, 1)
    tgt = mvr.synth_view(sp, 0, 10000)
    raw = mvr.synth_view(sp, 1, 10000)
    piv, ax = mvr.synth_prior(sp)
    prior = mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 12))
    src = orc.transform_f64(prior, raw)
    out = dict(tgt=tgt, raw=raw, prior=prior, src=src)
    for fma in (0, 1):
        idx, d2 = orc.nn(src, tgt, fma=bool(fma), kdtree=False)
        out["nn_idx_fma%d" % fma], out["nn_d2_fma%d" % fma] = idx, d2
    for name, rec in (("oneway", False), ("recip", True)):
        c = orc.correspondences(src, tgt, 4.0, reciprocal=rec, kdtree=False)
        out["corr_" + name] = c
    T1, mom = orc.umeyama(src, tgt, out["corr_recip"])
    out["moments"] = mom
    p1 = orc.make_params()                                              # reference settings: 1 iteration
    o1, T, st, rc = orc.icp_align(src, tgt, p1)
    out["align1_T"], out["align1_stats"] = T, np.array([st["iterations"], st["n_corr"], st["mse"]])
    p5 = orc.make_params(max_iter=5, teps=0.0, feps=-np.finfo(np.float64).max)
    o5, T5, st5, rc = orc.icp_align(src, tgt, p5)
    out["align5_T"], out["align5_stats"] = T5, np.array([st5["iterations"], st5["n_corr"], st5["mse"]])
    out["fitness1"] = np.array([orc.fitness(src, tgt, T)])
    np.savez_compressed(os.path.join(OUT, "pair_2x10000.npz"), **out)
    print("pair: n_corr", st["n_corr"], "mse", st["mse"], "-> 5 it mse", st5["mse"])


def ring_fixture():
    """12-view turntable ring, 2048 pts/scan: sequential driver + one LUM pass."""
    V, N = 12, 2048
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = ref_driver.init_poses(orc, V, piv, ax)
    params = orc.make_params(max_dist=8.0)
    poses, log = ref_driver.sequential_icp(orc, scans, poses0, params, V, repeat=2)
    out = dict(scans=np.stack(scans), prior_pivot=piv, prior_axis=ax, poses0=np.stack(poses0),
               seq_poses=np.stack(poses), seq_T=np.stack([e["T"] for e in log]),
               seq_ncorr=np.array([e["n_corr"] for e in log]), seq_mse=np.array([e["mse"] for e in log]),
               seq_view=np.array([e["view"] for e in log]),
               seq_fitness=np.array([e["fitness"] for e in log if "fitness" in e]))
    lum_poses, P, corrs, its = ref_driver.lum_pass(orc, scans, poses0, 8.0, 16)
    out.update(lum_poses=np.stack(lum_poses), lum_P=P, lum_ncorr=np.array([len(c) for c in corrs]),
               lum_its=np.array([its]))
    np.savez_compressed(os.path.join(OUT, "ring_12x2048.npz"), **out)
    print("ring: seq n_corr", out["seq_ncorr"][:11], "mse", np.round(out["seq_mse"][:11], 3))
    print("      last sweep mse", np.round(out["seq_mse"][11:], 3), "fitness", out["seq_fitness"])
    print("      lum n_corr", out["lum_ncorr"], "its", its)
    # how far from the truth are the poses?  truth: rotation about the TRUE axis
    for name, ps in (("prior", poses0), ("seq", poses), ("lum", lum_poses)):
        err = []
        for v in range(1, V):
            true = mvr.axis_rotation(np.array(sp.pivot), np.array(sp.axis), mvr.turntable_angle(v, V))
            c = np.append(np.array(sp.pivot), 1.0)
            err.append(np.linalg.norm((ps[v] @ c - true @ c)[:3]))
        print("      %-5s pivot-point error (mm): max %.3f mean %.3f" % (name, max(err), np.mean(err)))


def ring36_fixture():
    """BASELINE configs[4]'s shape in small: 36-view turntable ring (10 degrees apart), 768 pts/scan, THREE outer passes of
    registrationLUM (registrator.cpp:625-664) from the mis-calibrated prior -- the poses, the LUM poses and the per-edge
    correspondence counts after every pass."""
    V, N, passes = 36, 768, 3
    sp = mvr.synth_params(V, 5)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses = ref_driver.init_poses(orc, V, piv, ax)
    out = dict(scans=np.stack(scans), prior_pivot=piv, prior_axis=ax, poses0=np.stack(poses), origin=np.array(sp.pivot))
    P_all, n_all, poses_all, its_all = [], [], [], []
    for _ in range(passes):
        poses, P, corrs, its = ref_driver.lum_pass(orc, scans, poses, 8.0, 16)
        P_all.append(P); n_all.append([len(c) for c in corrs]); poses_all.append(np.stack(poses)); its_all.append(its)
    out.update(lum_poses=np.stack(poses_all), lum_P=np.stack(P_all), lum_ncorr=np.array(n_all), lum_its=np.array(its_all))
    np.savez_compressed(os.path.join(OUT, "ring_36x768.npz"), **out)
    print("ring36: n_corr per pass", [int(np.sum(n)) for n in n_all], "its", its_all)


if __name__ == "__main__":
    pair_fixture()
    ring_fixture()
    ring36_fixture()
