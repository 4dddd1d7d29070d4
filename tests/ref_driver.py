"""Test-side restatement of the reference's DRIVER loops on top of the oracle
(test infrastructure; never imported by the product).

  sequential_icp  <- Registrator::registrationICP(max_it, max_d, object)
                     mvr/src/registrator.cpp:526-588 (+ repeat wrapper :517-524)
  ring_edges      <- the pair lists of Registrator::computeError (:482-487) and
                     Registrator::registrationLUM (:640-643)
  lum_pass        <- one outer pass of Registrator::registrationLUM (:625-664)

Poses are 4x4 float64 in column-vector convention (the transpose of the
osg::Matrix the reference stores); `pose <- T_icp * pose` is
setMatrix(getMatrix()*result_matrix) of registrator.cpp:574.
"""
import numpy as np


def view_order(n_views=12):
    """registrator.cpp:530-541: 1, 11, 2, 10, ..., then the centre view."""
    half = n_views // 2
    order = []
    for i in range(1, half):
        order += [i, n_views - i]
    order.append(half)
    return order


def init_poses(orc, n_views, pivot, axis):
    """PointCloud::initRotation (point_cloud.cpp:400-413) for every view."""
    poses = [np.eye(4)]
    for v in range(1, n_views):
        poses.append(orc.axis_rotation(pivot, axis, orc.turntable_angle(v, n_views)))
    return poses


def sequential_icp(orc, scans, poses, params, n_views=None, repeat=1, fitness_last=True):
    """Returns (poses, log) where log holds one dict per align."""
    n_views = n_views or len(scans)
    poses = [p.copy() for p in poses]
    log = []
    for _ in range(repeat):
        target = orc.transform_f64(poses[0], scans[0])
        order = view_order(n_views)
        for k, v in enumerate(order):
            source = orc.transform_f64(poses[v], scans[v])
            out, T, st, rc = orc.icp_align(source, target, params)
            entry = dict(view=v, T=T.copy(), n_corr=st["n_corr"], mse=st["mse"],
                         iterations=st["iterations"], state=st["state"], nt=len(target))
            if fitness_last and k == len(order) - 1:
                entry["fitness"] = orc.fitness(source, target, T, fma=bool(params.fma_dist))
            log.append(entry)
            poses[v] = orc.mat4d_mul(T.astype(np.float64), poses[v])
            target = np.concatenate([target, out])
    return poses, log


def ring_edges(n_views=12):
    """registrator.cpp:640-643: (i, (i+1) % V)."""
    return [(i, (i + 1) % n_views) for i in range(n_views)]


def lum_pass(orc, scans, poses, max_dist, lum_iterations=16, fma=False):
    """One outer pass of registrationLUM: transformed clouds, ring reciprocal
    correspondences, LUM::compute, pose_i <- LUM_i * pose_i."""
    n = len(scans)
    clouds = [orc.transform_f64(poses[v], scans[v]) for v in range(n)]
    edges = ring_edges(n)
    corrs = [orc.correspondences(clouds[s], clouds[t], max_dist, reciprocal=True, fma=fma) for s, t in edges]
    P, its = orc.lum_compute(clouds, edges, corrs, max_iterations=lum_iterations)
    # lum.getTransformation(i) is an Eigen::Affine3f (float) before it is cast to osg::Matrix
    new = [orc.mat4d_mul(orc.pose_to_mat4(P[v]).astype(np.float32).astype(np.float64), poses[v]) for v in range(n)]
    return new, P, corrs, its
