"""ctypes binding of the CPU oracle (oracle/mvr_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/mvr_oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package, and only as the checker.  The product
(multi-view-registration_amd/) never does.

Point clouds are float32 arrays of shape (n, 4) (16-byte PointXYZ records);
poses are float32/float64 arrays of shape (4, 4) in the usual math layout
(T[r, c]); the binding converts to/from the column-major C layout.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "mvr_oracle.c")
    hdr = os.path.join(_HERE, "mvr_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Corr(C.Structure):
    _fields_ = [("query", C.c_int32), ("match", C.c_int32), ("dist2", C.c_float)]


class IcpParams(C.Structure):
    _fields_ = [("use_reciprocal", C.c_int), ("max_corr_dist", C.c_double),
                ("max_iterations", C.c_int), ("transformation_epsilon", C.c_double),
                ("euclidean_fitness_eps", C.c_double), ("fma_dist", C.c_int),
                ("use_kdtree", C.c_int)]


class IcpStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("state", C.c_int),
                ("n_corr", C.c_int), ("mse", C.c_double), ("evals", C.c_double)]


CORR_DTYPE = np.dtype([("query", "<i4"), ("match", "<i4"), ("dist2", "<f4")])
CONV_STATES = ("NOT_CONVERGED", "ITERATIONS", "TRANSFORM", "ABS_MSE", "REL_MSE",
               "NO_CORRESPONDENCES")

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, dp, u32p, vp = (C.POINTER(C.c_float), C.POINTER(C.c_double),
                            C.POINTER(C.c_uint32), C.c_void_p)
        L.orc_dist2.restype = C.c_float
        L.orc_dist2.argtypes = [fp, fp, C.c_int]
        L.orc_transform_f32.argtypes = [fp, fp, fp, C.c_size_t]
        L.orc_transform_f64.argtypes = [dp, fp, fp, C.c_size_t]
        for f in (L.orc_nn_brute, L.orc_nn_kdtree):
            f.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_int, u32p, fp]
        L.orc_correspondences.restype = C.c_size_t
        L.orc_correspondences.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_double,
                                          C.c_int, C.c_int, C.c_int, vp]
        L.orc_correspondences_mt.restype = C.c_size_t
        L.orc_correspondences_mt.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_double, C.c_int, C.c_int, C.c_int, vp]
        L.orc_denoise.restype = C.c_size_t
        L.orc_denoise.argtypes = [fp, C.c_size_t, C.c_int, C.c_double, u32p, u32p, C.POINTER(C.c_size_t)]
        L.orc_umeyama.restype = C.c_int
        L.orc_umeyama.argtypes = [fp, fp, vp, C.c_size_t, fp, dp]
        L.orc_umeyama_from_moments.argtypes = [dp, dp, dp, fp, dp]
        L.orc_svd3.argtypes = [dp, dp, dp, dp]
        L.orc_icp_align.restype = C.c_int
        L.orc_icp_align.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.POINTER(IcpParams),
                                    fp, fp, C.POINTER(IcpStats)]
        L.orc_p2plane.restype = C.c_int
        L.orc_p2plane.argtypes = [fp, fp, fp, vp, C.c_size_t, fp, dp]
        L.orc_icp_align_p2plane.restype = C.c_int
        L.orc_icp_align_p2plane.argtypes = [fp, C.c_size_t, fp, fp, C.c_size_t, C.POINTER(IcpParams),
                                            fp, fp, C.POINTER(IcpStats)]
        L.orc_fitness.restype = C.c_double
        L.orc_fitness.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.c_double,
                                  C.c_int, C.c_int]
        L.orc_mat4f_mul.argtypes = [fp, fp, fp]
        L.orc_mat4d_mul.argtypes = [dp, dp, dp]
        L.orc_axis_rotation.argtypes = [dp, dp, C.c_double, dp]
        L.orc_turntable_angle.restype = C.c_double
        L.orc_turntable_angle.argtypes = [C.c_int, C.c_int]
        L.orc_pose_to_mat4.argtypes = [dp, dp]
        L.orc_lum_edge.restype = C.c_size_t
        L.orc_lum_edge.argtypes = [fp, fp, vp, C.c_size_t, dp, dp, dp, dp, dp]
        L.orc_lum_compute.restype = C.c_int
        L.orc_lum_compute.argtypes = [C.c_int, C.POINTER(fp), C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(vp),
                                      C.POINTER(C.c_size_t), C.c_int, C.c_double, dp]
        L.orc_solve_dense.restype = C.c_int
        L.orc_solve_dense.argtypes = [C.c_int, dp, dp]
        L.orc_invert6.restype = C.c_int
        L.orc_invert6.argtypes = [dp, dp]
        L.orc_umeyama_f32.restype = C.c_int
        L.orc_umeyama_f32.argtypes = [fp, fp, vp, C.c_size_t, C.c_int, fp]
        L.orc_lum_incidence.argtypes = [dp, dp]
        L.orc_refine_axis.restype = C.c_int
        L.orc_refine_axis.argtypes = [C.c_int, dp, C.c_float, fp, fp]
        _lib = L
    return _lib


# ----------------------------------------------------------------- helpers

def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _pts(a):
    a = _f32(a)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("points must have shape (n, 4) float32")
    return a


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def to_cm(T, dtype):
    """(4,4) math layout -> 16 contiguous column-major values."""
    return np.ascontiguousarray(np.asarray(T, dtype=dtype).T).reshape(16)


def from_cm(v):
    return np.array(v).reshape(4, 4).T.copy()


def make_params(reciprocal=True, max_dist=4.0, max_iter=10, teps=1e-6, feps=64.0,
                fma=False, kdtree=True) -> IcpParams:
    """Defaults = the reference's configuration, mvr/src/registrator.cpp:551-560."""
    return IcpParams(int(reciprocal), float(max_dist), int(max_iter), float(teps),
                     float(feps), int(fma), int(kdtree))


# --------------------------------------------------------------- functions

def dist2(a, b, fma=False) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dist2(_p(a, C.c_float), _p(b, C.c_float), int(fma)))


def transform_f32(T, pts):
    pts = _pts(pts)
    out = np.empty_like(pts)
    t = to_cm(T, np.float32)
    lib().orc_transform_f32(_p(t, C.c_float), _p(pts, C.c_float), _p(out, C.c_float), len(pts))
    return out


def transform_f64(T, pts):
    pts = _pts(pts)
    out = np.empty_like(pts)
    t = to_cm(T, np.float64)
    lib().orc_transform_f64(_p(t, C.c_double), _p(pts, C.c_float), _p(out, C.c_float), len(pts))
    return out


def nn(q, t, fma=False, kdtree=False):
    q, t = _pts(q), _pts(t)
    idx = np.empty(len(q), np.uint32)
    d2 = np.empty(len(q), np.float32)
    f = lib().orc_nn_kdtree if kdtree else lib().orc_nn_brute
    f(_p(q, C.c_float), len(q), _p(t, C.c_float), len(t), int(fma), _p(idx, C.c_uint32),
      _p(d2, C.c_float))
    return idx, d2


def correspondences(src, tgt, max_dist, reciprocal=True, fma=False, kdtree=True):
    src, tgt = _pts(src), _pts(tgt)
    out = np.empty(max(len(src), 1), CORR_DTYPE)
    m = lib().orc_correspondences(_p(src, C.c_float), len(src), _p(tgt, C.c_float), len(tgt),
                                  float(max_dist), int(reciprocal), int(fma), int(kdtree),
                                  out.ctypes.data)
    return out[:m].copy()


def correspondences_mt(src, tgt, max_dist, threads, reciprocal=True, fma=False):
    """kd-tree correspondences with the per-query searches on `threads` OpenMP threads (same output)."""
    src, tgt = _pts(src), _pts(tgt)
    out = np.empty(max(len(src), 1), CORR_DTYPE)
    m = lib().orc_correspondences_mt(_p(src, C.c_float), len(src), _p(tgt, C.c_float), len(tgt), float(max_dist),
                                     int(reciprocal), int(fma), int(threads), out.ctypes.data)
    return out[:m].copy()


def denoise(pts, segment_threshold=10, triangle_length=2.5):
    """PointCloud::denoise: returns (kept original indices in output order, label per point, number of components)."""
    pts = _pts(pts)
    out = np.empty(max(len(pts), 1), np.uint32)
    lab = np.empty(max(len(pts), 1), np.uint32)
    nc = C.c_size_t()
    k = lib().orc_denoise(_p(pts, C.c_float), len(pts), int(segment_threshold), float(triangle_length),
                          _p(out, C.c_uint32), _p(lab, C.c_uint32), C.byref(nc))
    return out[:k].copy(), lab[:len(pts)].copy(), nc.value


def umeyama(src, tgt, corr):
    """Returns (T (4,4) float32, moments[20]) or (None, None) if < 3 pairs."""
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, dtype=CORR_DTYPE)
    T = np.empty(16, np.float32)
    mom = np.empty(20, np.float64)
    rc = lib().orc_umeyama(_p(src, C.c_float), _p(tgt, C.c_float), corr.ctypes.data, len(corr),
                           _p(T, C.c_float), _p(mom, C.c_double))
    if rc != 0:
        return None, None
    return from_cm(T), mom


def umeyama_from_moments(mean_src, mean_tgt, sigma):
    ms = np.ascontiguousarray(mean_src, np.float64)
    mt = np.ascontiguousarray(mean_tgt, np.float64)
    sg = np.ascontiguousarray(sigma, np.float64).reshape(9)
    T = np.empty(16, np.float32)
    sv = np.empty(3, np.float64)
    lib().orc_umeyama_from_moments(_p(ms, C.c_double), _p(mt, C.c_double), _p(sg, C.c_double),
                                   _p(T, C.c_float), _p(sv, C.c_double))
    return from_cm(T), sv


def svd3(A):
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    U, S, V = np.empty(9), np.empty(3), np.empty(9)
    lib().orc_svd3(_p(A, C.c_double), _p(U, C.c_double), _p(S, C.c_double), _p(V, C.c_double))
    return U.reshape(3, 3), S, V.reshape(3, 3)


def icp_align(src, tgt, params: IcpParams):
    """Returns (out_points, T (4,4) float32, stats dict, rc)."""
    src, tgt = _pts(src), _pts(tgt)
    out = np.empty_like(src)
    T = np.empty(16, np.float32)
    st = IcpStats()
    rc = lib().orc_icp_align(_p(src, C.c_float), len(src), _p(tgt, C.c_float), len(tgt),
                             C.byref(params), _p(out, C.c_float), _p(T, C.c_float), C.byref(st))
    stats = dict(iterations=st.iterations, converged=bool(st.converged),
                 state=CONV_STATES[st.state], n_corr=st.n_corr, mse=st.mse, evals=st.evals)
    return out, from_cm(T), stats, rc


def p2plane(src, tgt, tnrm, corr):
    """EXTENSION: point-to-plane LLS estimate -> (T (4,4) float32, sums[29]) or (None, None)."""
    src, tgt, tnrm = _pts(src), _pts(tgt), _pts(tnrm)
    corr = np.ascontiguousarray(corr, dtype=CORR_DTYPE)
    T, sums = np.empty(16, np.float32), np.empty(29, np.float64)
    rc = lib().orc_p2plane(_p(src, C.c_float), _p(tgt, C.c_float), _p(tnrm, C.c_float), corr.ctypes.data, len(corr),
                           _p(T, C.c_float), _p(sums, C.c_double))
    return (from_cm(T), sums) if rc == 0 else (None, None)


def icp_align_p2plane(src, tgt, tnrm, params: IcpParams):
    src, tgt, tnrm = _pts(src), _pts(tgt), _pts(tnrm)
    out = np.empty_like(src)
    T = np.empty(16, np.float32)
    st = IcpStats()
    rc = lib().orc_icp_align_p2plane(_p(src, C.c_float), len(src), _p(tgt, C.c_float), _p(tnrm, C.c_float), len(tgt),
                                     C.byref(params), _p(out, C.c_float), _p(T, C.c_float), C.byref(st))
    stats = dict(iterations=st.iterations, converged=bool(st.converged), state=CONV_STATES[st.state],
                 n_corr=st.n_corr, mse=st.mse, evals=st.evals)
    return out, from_cm(T), stats, rc


def fitness(inp, tgt, T, max_range=np.finfo(np.float64).max, fma=False, kdtree=True) -> float:
    inp, tgt = _pts(inp), _pts(tgt)
    t = to_cm(T, np.float32)
    return float(lib().orc_fitness(_p(inp, C.c_float), len(inp), _p(tgt, C.c_float), len(tgt),
                                   _p(t, C.c_float), float(max_range), int(fma), int(kdtree)))


def mat4f_mul(A, B):
    a, b = to_cm(A, np.float32), to_cm(B, np.float32)
    c = np.empty(16, np.float32)
    lib().orc_mat4f_mul(_p(a, C.c_float), _p(b, C.c_float), _p(c, C.c_float))
    return from_cm(c)


def mat4d_mul(A, B):
    a, b = to_cm(A, np.float64), to_cm(B, np.float64)
    c = np.empty(16, np.float64)
    lib().orc_mat4d_mul(_p(a, C.c_double), _p(b, C.c_double), _p(c, C.c_double))
    return from_cm(c)


def axis_rotation(pivot, axis, angle):
    p = np.ascontiguousarray(pivot, np.float64)
    a = np.ascontiguousarray(axis, np.float64)
    T = np.empty(16, np.float64)
    lib().orc_axis_rotation(_p(p, C.c_double), _p(a, C.c_double), float(angle), _p(T, C.c_double))
    return from_cm(T)


def turntable_angle(view, n_views=12) -> float:
    return float(lib().orc_turntable_angle(int(view), int(n_views)))


def pose_to_mat4(pose):
    p = np.ascontiguousarray(pose, np.float64)
    T = np.empty(16, np.float64)
    lib().orc_pose_to_mat4(_p(p, C.c_double), _p(T, C.c_double))
    return from_cm(T)


def lum_edge(src, tgt, corr, pose_s, pose_t):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, dtype=CORR_DTYPE)
    ps = np.ascontiguousarray(pose_s, np.float64)
    pt = np.ascontiguousarray(pose_t, np.float64)
    MM, MZ, ss = np.empty(36), np.empty(6), C.c_double()
    n = lib().orc_lum_edge(_p(src, C.c_float), _p(tgt, C.c_float), corr.ctypes.data, len(corr),
                           _p(ps, C.c_double), _p(pt, C.c_double), _p(MM, C.c_double),
                           _p(MZ, C.c_double), C.byref(ss))
    return int(n), MM.reshape(6, 6), MZ, ss.value


def lum_compute(clouds, edges, corrs, max_iterations=5, threshold=0.0, poses=None):
    """clouds: list of (n,4) arrays; edges: list of (s,t); corrs: list of CORR arrays."""
    n, ne = len(clouds), len(edges)
    clouds = [_pts(c) for c in clouds]
    corrs = [np.ascontiguousarray(c, dtype=CORR_DTYPE) for c in corrs]
    cl = (C.POINTER(C.c_float) * n)(*[_p(c, C.c_float) for c in clouds])
    es = (C.c_int * ne)(*[e[0] for e in edges])
    et = (C.c_int * ne)(*[e[1] for e in edges])
    cp = (C.c_void_p * ne)(*[c.ctypes.data for c in corrs])
    nc = (C.c_size_t * ne)(*[len(c) for c in corrs])
    P = np.zeros((n, 6), np.float64) if poses is None else np.array(poses, np.float64).reshape(n, 6)
    P = np.ascontiguousarray(P)
    its = lib().orc_lum_compute(n, cl, ne, es, et, cp, nc, int(max_iterations), float(threshold),
                                _p(P, C.c_double))
    return P, int(its)


def solve_dense(A, b):
    A = np.array(A, np.float64, order="C")
    x = np.array(b, np.float64)
    rc = lib().orc_solve_dense(len(x), _p(A, C.c_double), _p(x, C.c_double))
    return x if rc == 0 else None


def umeyama_f32(src, tgt, corr, block=0):
    """The estimate in Eigen's own float arithmetic (a model; see orc_umeyama_f32) -> T (4,4) float32 or None."""
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, dtype=CORR_DTYPE)
    T = np.empty(16, np.float32)
    rc = lib().orc_umeyama_f32(_p(src, C.c_float), _p(tgt, C.c_float), corr.ctypes.data, len(corr), int(block), _p(T, C.c_float))
    return from_cm(T) if rc == 0 else None


def lum_incidence(pose):
    p, out = np.ascontiguousarray(pose, np.float64), np.empty(36)
    lib().orc_lum_incidence(_p(p, C.c_double), _p(out, C.c_double))
    return out.reshape(6, 6)


def refine_axis(poses, pivot_y):
    """Registrator::refineAxis on a list of registered (4,4) column-vector poses -> (rc, axis float32[3], pivot float32[3])."""
    P = np.ascontiguousarray(np.asarray(poses, np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)).reshape(-1, 16)
    ax, pv = np.zeros(3, np.float32), np.zeros(3, np.float32)
    rc = lib().orc_refine_axis(len(P), _p(P, C.c_double), float(np.float32(pivot_y)), _p(ax, C.c_float), _p(pv, C.c_float))
    return rc, ax, pv
