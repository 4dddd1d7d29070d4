"""multi-view-registration_amd -- MI355X-native ICP hot path of
fanxiaochen/Multi-View-Registration (`mvr`).

This package is a thin ctypes view of the C-ABI in include/mvr_hip.h
(libmvr_hip.so: hand-written HIP kernels for gfx950 + the host-side solves).
The product host code is C++ (include/mvr/*.hpp, the PCL-style shim the
reference's Registrator call sites compile against); Python is only used to
drive tests and bench.py.  There is no CPU fallback: without the built
extension the import fails, without a GPU Context() raises.

Import with importlib (the directory name has hyphens):
    mvr = importlib.import_module("multi-view-registration_amd")
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmvr_hip.so")
if os.environ.get("MVR_LIB_VARIANT"):      # tuning experiments only (tools/build_variant.sh): build/libmvr_hip_<name>.so
    LIB_PATH = os.path.join(os.path.dirname(_HERE), "build", "libmvr_hip_%s.so" % os.environ["MVR_LIB_VARIANT"])

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libmvr_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). There is no fallback implementation.")



def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (SONAME libamdhip64.so.7, the name libmvr_hip.so needs) but
    ask for it as "libamdhip64.so"; if libmvr_hip.so were loaded first against
    /opt/rocm, a later `import torch` would bring a SECOND runtime into the
    process and find no GPU.  Loading torch's copy first makes both import
    orders resolve to the same runtime.  MVR_HIP_RUNTIME=system skips this."""
    if os.environ.get("MVR_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
            # ... and one RCCL: the copy built for that runtime (loaded only when a multi-GPU entry point is used)
            r = os.path.join(os.path.dirname(p), "librccl.so")
            if os.path.exists(r):
                os.environ.setdefault("MVR_RCCL_LIB", r)


_share_hip_runtime_with_torch()
_lib = C.CDLL(LIB_PATH)

MAX_SLOTS = 256
OK, E_ARG, E_HIP, E_NOCORR, E_NOMEM, E_SINGULAR, E_RCCL = 0, -1, -2, -3, -4, -5, -6
CONV_STATES = ("NOT_CONVERGED", "ITERATIONS", "TRANSFORM", "ABS_MSE", "REL_MSE",
               "NO_CORRESPONDENCES")
K_NN, K_REDUCE, K_XFORM, K_GLUE, K_NN_GRID, K_NN_WIDE = 0, 1, 2, 3, 4, 5


class MvrError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = _lib.mvr_strerror(status).decode()
        super().__init__("mvr status %d (%s)%s" % (status, msg, (": " + detail) if detail else ""))


class IcpParams(C.Structure):
    _fields_ = [("use_reciprocal", C.c_int), ("max_corr_dist", C.c_double),
                ("max_iterations", C.c_int), ("transformation_epsilon", C.c_double),
                ("euclidean_fitness_eps", C.c_double), ("fma_dist", C.c_int), ("point_to_plane", C.c_int)]


class IcpStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("state", C.c_int),
                ("n_corr", C.c_int), ("mse", C.c_double), ("evals", C.c_double),
                ("fwd_queries", C.c_double), ("ms", C.c_double)]


class PairMoments(C.Structure):
    _fields_ = [("n", C.c_double), ("mean_src", C.c_double * 3), ("mean_tgt", C.c_double * 3),
                ("mse", C.c_double), ("sigma", C.c_double * 9)]


class PairMoments2(C.Structure):
    _fields_ = [("n", C.c_double), ("origin", C.c_double * 3), ("sp", C.c_double * 3),
                ("sq", C.c_double * 3), ("spp", C.c_double * 6), ("sqq", C.c_double * 6),
                ("spq", C.c_double * 9), ("sum_d2", C.c_double)]


class SynthParams(C.Structure):
    _fields_ = [("n_views", C.c_int), ("seed", C.c_uint64), ("noise_sigma", C.c_double),
                ("pivot", C.c_double * 3), ("axis", C.c_double * 3)]


_fp, _dp, _u32p, _i32p = (C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint32),
                          C.POINTER(C.c_int32))
_vp, _sz = C.c_void_p, C.c_size_t

# name -> (restype, argtypes); also the list the symbol-export test checks
SIGNATURES = {
    "mvr_ctx_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "mvr_ctx_create_on_stream": (C.c_int, [C.POINTER(_vp), C.c_int, _vp]),
    "mvr_ctx_destroy": (C.c_int, [_vp]),
    "mvr_ctx_sync": (C.c_int, [_vp]),
    "mvr_strerror": (C.c_char_p, [C.c_int]),
    "mvr_last_error": (C.c_char_p, [_vp]),
    "mvr_device_info": (C.c_int, [_vp, C.c_char_p, _sz, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mvr_cloud_upload": (C.c_int, [_vp, C.c_int, _fp, _sz, _sz]),
    "mvr_cloud_download": (C.c_int, [_vp, C.c_int, _fp, _sz, _sz, C.POINTER(_sz)]),
    "mvr_cloud_size": (C.c_int, [_vp, C.c_int, C.POINTER(_sz)]),
    "mvr_cloud_reserve": (C.c_int, [_vp, C.c_int, _sz]),
    "mvr_cloud_copy": (C.c_int, [_vp, C.c_int, C.c_int]),
    "mvr_cloud_append": (C.c_int, [_vp, C.c_int, C.c_int]),
    "mvr_cloud_denoise": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _u32p]),
    "mvr_cloud_set_global_base": (C.c_int, [_vp, C.c_int, _sz]),
    "mvr_cloud_append_range": (C.c_int, [_vp, C.c_int, C.c_int, _sz, _sz, _sz]),
    "mvr_nn_forward_keys": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int, _vp]),
    "mvr_pair_moments2_from_keys": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_double, C.c_int, C.c_int, _dp, _vp]),
    "mvr_cloud_clear": (C.c_int, [_vp, C.c_int]),
    "mvr_cloud_transform": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "mvr_cloud_transform_f32": (C.c_int, [_vp, C.c_int, C.c_int, _fp]),
    "mvr_cloud_upload_normals": (C.c_int, [_vp, C.c_int, _fp, _sz, _sz]),
    "mvr_cloud_download_normals": (C.c_int, [_vp, C.c_int, _fp, _sz, _sz, C.POINTER(_sz)]),
    "mvr_nn": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _u32p, _fp]),
    "mvr_correspondences": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                      _i32p, _i32p, _fp, _sz, C.POINTER(_sz)]),
    "mvr_pair_moments": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                   C.POINTER(PairMoments)]),
    "mvr_pair_moments2": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _sz, _sz,
                                    _dp, C.POINTER(PairMoments2)]),
    "mvr_pair_moments2_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _sz,
                                        _sz, _dp, _vp]),
    "mvr_cloud_transform_batch": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp]),
    "mvr_pair_moments2_batch": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_double, C.c_int,
                                          C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _dp,
                                          C.POINTER(PairMoments2), _vp]),
    "mvr_seq_align_sharded": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(IcpParams), _dp, _fp, C.POINTER(IcpStats)]),
    "mvr_seq_run_sharded": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(IcpParams), _dp, C.c_int, _dp,
                                      C.POINTER(C.c_int), _fp, C.POINTER(IcpStats), C.POINTER(C.c_int)]),
    "mvr_seq_run": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(IcpParams), C.c_int, _dp,
                              C.POINTER(C.c_int), _fp, C.POINTER(IcpStats), C.POINTER(C.c_int)]),
    "mvr_pair_batch_correspondences": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _fp, _sz, C.POINTER(_sz)]),
    "mvr_pair_moments2_from_corr": (C.c_int, [_vp, C.c_int, C.c_int, _i32p, _i32p, _sz, _dp,
                                              C.POINTER(PairMoments2)]),
    "mvr_umeyama_from_moments": (C.c_int, [C.POINTER(PairMoments), _fp, _dp]),
    "mvr_moments_from_moments2": (C.c_int, [C.POINTER(PairMoments2), C.POINTER(PairMoments)]),
    "mvr_icp_align": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(IcpParams), _fp,
                                C.POINTER(IcpStats)]),
    "mvr_fitness": (C.c_int, [_vp, C.c_int, C.c_int, _fp, C.c_double, C.c_int, _dp]),
    "mvr_lum_edge_from_moments": (C.c_int, [C.POINTER(PairMoments2), _dp, _dp, _dp, _dp, _dp]),
    "mvr_lum_edge_from_moments_x4": (C.c_int, [C.POINTER(PairMoments2), _dp, _dp, _dp, _dp, _dp]),
    "mvr_lum_compute": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(PairMoments2), C.c_int, C.c_double, _dp,
                                  C.POINTER(C.c_int)]),
    "mvr_ring_host_step": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, _dp, C.c_int,
                                     _dp, _dp, _fp, _dp, _dp, C.POINTER(C.c_int)]),
    "mvr_ring_step": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                                C.POINTER(C.c_int), C.c_double, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, _fp, _dp, _dp,
                                C.POINTER(C.c_int), _dp, _dp]),
    "mvr_ring_run": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                               C.POINTER(C.c_int), C.c_double, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, _fp, _dp, _dp,
                               C.POINTER(C.c_int), _dp, _dp]),
    "mvr_comm_unique_id": (C.c_int, [C.c_char_p]),
    "mvr_ctx_comm_init": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    "mvr_ctx_comm_destroy": (C.c_int, [_vp]),
    "mvr_ctx_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mvr_rccl_library": (C.c_char_p, []),
    "mvr_pool_trim": (C.c_uint64, [C.POINTER(C.c_uint64)]),
    "mvr_ring_segments": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "mvr_ring_run_sharded": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.c_double, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, _fp, _dp, _dp,
                                       C.POINTER(C.c_int), _dp, _dp]),
    "mvr_ctx_project": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_int]),
    "mvr_debug_order": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mvr_ring_rows_sharded": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.c_double, C.c_int, C.c_int, _dp, _dp, _dp]),
    "mvr_world_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(C.c_int)]),
    "mvr_world_destroy": (C.c_int, [_vp]),
    "mvr_world_size": (C.c_int, [_vp]),
    "mvr_world_ctx": (_vp, [_vp, C.c_int]),
    "mvr_world_last_error": (C.c_char_p, [_vp]),
    "mvr_world_upload": (C.c_int, [_vp, C.c_int, _fp, _sz, _sz]),
    "mvr_world_ring_run": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.c_double, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, _fp, _dp, _dp,
                                     C.POINTER(C.c_int), _dp, _dp]),
    "mvr_pose_to_mat4": (None, [_dp, _dp]),
    "mvr_lum_incidence": (None, [_dp, _dp]),
    "mvr_refine_axis": (C.c_int, [C.c_int, _dp, C.c_float, _fp, _fp]),
    "mvr_turntable_angle": (C.c_double, [C.c_int, C.c_int]),
    "mvr_axis_rotation": (None, [_dp, _dp, C.c_double, _dp]),
    "mvr_mat4d_mul": (None, [_dp, _dp, _dp]),
    "mvr_mat4f_mul": (None, [_fp, _fp, _fp]),
    "mvr_ctx_tune": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "mvr_ctx_stat": (C.c_int, [_vp, C.c_char_p, _dp]),
    "mvr_ctx_pass_log": (C.c_int, [_vp, _dp, C.c_int, C.POINTER(C.c_int)]),
    "mvr_debug_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int]),
    "mvr_prof_enable": (C.c_int, [_vp, C.c_int]),
    "mvr_prof_reset": (C.c_int, [_vp]),
    "mvr_prof_get": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint64), _dp, _dp]),
    "mvr_synth_default": (None, [C.POINTER(SynthParams), C.c_int, C.c_int]),
    "mvr_synth_view": (C.c_int, [C.POINTER(SynthParams), C.c_int, _sz, _fp, _fp]),
    "mvr_synth_prior": (None, [C.POINTER(SynthParams), _dp, _dp]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(_lib, _name)
    _f.restype, _f.argtypes = _res, _args


def _chk(rc, ctx=None, allow=()):
    if rc != OK and rc not in allow:
        detail = _lib.mvr_last_error(ctx).decode() if ctx else ""
        raise MvrError(rc, detail)
    return rc


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def to_cm(T, dtype):
    """(4,4) math layout T[r,c] -> 16 column-major values (Eigen::Matrix4f memory)."""
    return np.ascontiguousarray(np.asarray(T, dtype=dtype).T).reshape(16)


def from_cm(v):
    return np.array(v).reshape(4, 4).T.copy()


# --------------------------------------------------------------- host helpers

def icp_params(reciprocal=True, max_dist=4.0, max_iter=10, teps=1e-6, feps=64.0, fma=False, point_to_plane=False):
    """Defaults = the reference's settings at mvr/src/registrator.cpp:551-560
    (max_iterations is the caller's; with feps=64 every align is one iteration)."""
    return IcpParams(int(reciprocal), float(max_dist), int(max_iter), float(teps), float(feps),
                     int(fma), int(point_to_plane))


def synth_params(n_views=12, config_id=0) -> SynthParams:
    sp = SynthParams()
    _lib.mvr_synth_default(C.byref(sp), n_views, config_id)
    return sp


def synth_view(sp: SynthParams, view: int, n: int, normals=False):
    pts = np.empty((n, 4), np.float32)
    nrm = np.empty((n, 4), np.float32) if normals else None
    _chk(_lib.mvr_synth_view(C.byref(sp), view, n, _p(pts, C.c_float),
                             _p(nrm, C.c_float) if normals else None))
    return (pts, nrm) if normals else pts


def synth_prior(sp: SynthParams):
    piv, ax = np.empty(3), np.empty(3)
    _lib.mvr_synth_prior(C.byref(sp), _p(piv, C.c_double), _p(ax, C.c_double))
    return piv, ax


def turntable_angle(view, n_views=12) -> float:
    return float(_lib.mvr_turntable_angle(view, n_views))


def axis_rotation(pivot, axis, angle):
    p, a = np.ascontiguousarray(pivot, np.float64), np.ascontiguousarray(axis, np.float64)
    T = np.empty(16)
    _lib.mvr_axis_rotation(_p(p, C.c_double), _p(a, C.c_double), float(angle), _p(T, C.c_double))
    return from_cm(T)


def mat4d_mul(A, B):
    a, b, c = to_cm(A, np.float64), to_cm(B, np.float64), np.empty(16)
    _lib.mvr_mat4d_mul(_p(a, C.c_double), _p(b, C.c_double), _p(c, C.c_double))
    return from_cm(c)


def mat4f_mul(A, B):
    a, b, c = to_cm(A, np.float32), to_cm(B, np.float32), np.empty(16, np.float32)
    _lib.mvr_mat4f_mul(_p(a, C.c_float), _p(b, C.c_float), _p(c, C.c_float))
    return from_cm(c)


def pose_to_mat4(pose):
    p, T = np.ascontiguousarray(pose, np.float64), np.empty(16)
    _lib.mvr_pose_to_mat4(_p(p, C.c_double), _p(T, C.c_double))
    return from_cm(T)


def lum_incidence(pose):
    p, H = np.ascontiguousarray(pose, np.float64), np.empty(36)
    _lib.mvr_lum_incidence(_p(p, C.c_double), _p(H, C.c_double))
    return H.reshape(6, 6)


def refine_axis(poses, pivot_y):
    """Registrator::refineAxis on the registered views' (4,4) column-vector poses -> (rc, axis f32[3], pivot f32[3])."""
    P = np.ascontiguousarray(np.asarray(poses, np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)).reshape(-1, 16)
    ax, pv = np.zeros(3, np.float32), np.zeros(3, np.float32)
    rc = _lib.mvr_refine_axis(len(P), _p(P, C.c_double), float(np.float32(pivot_y)), _p(ax, C.c_float), _p(pv, C.c_float))
    return rc, ax, pv


def moments_to_dict(m: PairMoments):
    return dict(n=m.n, mean_src=np.array(m.mean_src), mean_tgt=np.array(m.mean_tgt), mse=m.mse,
                sigma=np.array(m.sigma).reshape(3, 3))


def umeyama_from_moments(m: PairMoments):
    T, sv = np.empty(16, np.float32), np.empty(3)
    rc = _lib.mvr_umeyama_from_moments(C.byref(m), _p(T, C.c_float), _p(sv, C.c_double))
    if rc != OK:
        return None, None
    return from_cm(T), sv


def moments_from_moments2(m2: PairMoments2) -> PairMoments:
    out = PairMoments()
    _chk(_lib.mvr_moments_from_moments2(C.byref(m2), C.byref(out)))
    return out


def moments2_from_row(row) -> PairMoments2:
    """32-double device row {n, origin, sp, sq, spp, sqq, spq, 0} -> struct."""
    m2 = PairMoments2()
    C.memmove(C.byref(m2), np.ascontiguousarray(row, np.float64).ctypes.data, C.sizeof(m2))
    return m2


def lum_edge_from_moments(m2: PairMoments2, pose_s, pose_t):
    ps, pt = np.ascontiguousarray(pose_s, np.float64), np.ascontiguousarray(pose_t, np.float64)
    MM, MZ, ss = np.empty(36), np.empty(6), C.c_double()
    rc = _lib.mvr_lum_edge_from_moments(C.byref(m2), _p(ps, C.c_double), _p(pt, C.c_double),
                                        _p(MM, C.c_double), _p(MZ, C.c_double), C.byref(ss))
    return rc, MM.reshape(6, 6), MZ, ss.value


def lum_edge_from_moments_x4(m2s, poses_s, poses_t):
    """Four edges through the AVX2 pass of LUM::computeEdge: (rc, MM[4,6,6], MZ[4,6], ss[4])."""
    arr = (PairMoments2 * 4)(*m2s)
    ps, pt = np.ascontiguousarray(poses_s, np.float64).reshape(4, 6), np.ascontiguousarray(poses_t, np.float64).reshape(4, 6)
    MM, MZ, ss = np.empty((4, 36)), np.empty((4, 6)), np.empty(4)
    rc = _lib.mvr_lum_edge_from_moments_x4(arr, _p(ps, C.c_double), _p(pt, C.c_double), _p(MM, C.c_double), _p(MZ, C.c_double), _p(ss, C.c_double))
    return rc, MM.reshape(4, 6, 6), MZ, ss


_edge_cache = {}


def ring_host_step(n_views, edges, rows, origin, poses, lum_iterations=16):
    """One call for the host side of a global step.  rows: (ne, 32) float64 edge
    table; poses: list of (4,4) float64.  Returns (rc, new_poses, info dict)."""
    ne = len(edges)
    key = tuple(map(tuple, edges))
    if _edge_cache.get("key") != key:                       # the graph of a registration does not change between steps
        _edge_cache.update(key=key, es=(C.c_int * ne)(*[e[0] for e in edges]), et=(C.c_int * ne)(*[e[1] for e in edges]))
    es, et = _edge_cache["es"], _edge_cache["et"]
    R = np.ascontiguousarray(rows, np.float64).reshape(ne, 32)
    o = np.ascontiguousarray(origin, np.float64)
    P = np.ascontiguousarray(np.asarray(poses, np.float64).reshape(n_views, 4, 4).transpose(0, 2, 1)).reshape(n_views, 16)
    lum = np.zeros((n_views, 6))
    pT, pn, pm, its = np.empty((ne, 16), np.float32), np.empty(ne), np.empty(ne), C.c_int()
    rc = _lib.mvr_ring_host_step(n_views, ne, es, et, _p(R, C.c_double), _p(o, C.c_double), int(lum_iterations),
                                 _p(P, C.c_double), _p(lum, C.c_double), _p(pT, C.c_float), _p(pn, C.c_double),
                                 _p(pm, C.c_double), C.byref(its))
    new = list(np.ascontiguousarray(P.reshape(n_views, 4, 4).transpose(0, 2, 1)))
    info = dict(pair_T=list(np.ascontiguousarray(pT.reshape(ne, 4, 4).transpose(0, 2, 1))), pair_n=pn.tolist(),
                pair_mse=pm.tolist(), lum_pose=lum, lum_iterations=its.value)
    return rc, new, info


def lum_compute(n, edges, moments2, max_iterations=5, threshold=0.0, poses=None):
    ne = len(edges)
    es = (C.c_int * ne)(*[e[0] for e in edges])
    et = (C.c_int * ne)(*[e[1] for e in edges])
    arr = (PairMoments2 * ne)(*moments2)
    P = np.zeros((n, 6)) if poses is None else np.array(poses, np.float64).reshape(n, 6)
    P = np.ascontiguousarray(P)
    its = C.c_int()
    rc = _lib.mvr_lum_compute(n, ne, es, et, arr, int(max_iterations), float(threshold),
                              _p(P, C.c_double), C.byref(its))
    return rc, P, its.value


# ------------------------------------------------------------------- context

class Context:
    """One GPU context (mvr_ctx): device clouds in numbered slots + the hot path."""

    def __init__(self, device=0, stream=None):
        h = _vp()
        rc = _lib.mvr_ctx_create_on_stream(C.byref(h), int(device), _vp(stream) if stream else None)
        if rc != OK:
            raise MvrError(rc, "mvr_ctx_create: no usable GPU or HIP failure; there is no CPU fallback")
        self._h = h

    def close(self):
        if self._h:
            _lib.mvr_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- clouds
    def upload(self, slot, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        if pts.ndim != 2 or pts.shape[1] not in (3, 4):
            raise ValueError("points must be (n,4) or (n,3) float32")
        _chk(_lib.mvr_cloud_upload(self._h, slot, _p(pts, C.c_float), len(pts), 4 * pts.shape[1]), self._h)

    def size(self, slot) -> int:
        n = _sz()
        _chk(_lib.mvr_cloud_size(self._h, slot, C.byref(n)), self._h)
        return n.value

    def download(self, slot, packed=False):
        n = self.size(slot)
        out = np.empty((n, 3 if packed else 4), np.float32)
        got = _sz()
        _chk(_lib.mvr_cloud_download(self._h, slot, _p(out, C.c_float), n, 12 if packed else 16,
                                     C.byref(got)), self._h)
        return out

    def upload_normals(self, slot, normals):
        nr = np.ascontiguousarray(normals, np.float32)
        if nr.ndim != 2 or nr.shape[1] not in (3, 4):
            raise ValueError("normals must be (n,4) or (n,3) float32")
        _chk(_lib.mvr_cloud_upload_normals(self._h, slot, _p(nr, C.c_float), len(nr), 4 * nr.shape[1]), self._h)

    def download_normals(self, slot):
        n = self.size(slot)
        out = np.empty((n, 4), np.float32)
        got = _sz()
        _chk(_lib.mvr_cloud_download_normals(self._h, slot, _p(out, C.c_float), n, 16, C.byref(got)), self._h)
        return out[:got.value]

    def reserve(self, slot, cap):
        _chk(_lib.mvr_cloud_reserve(self._h, slot, cap), self._h)

    def copy(self, dst, src):
        _chk(_lib.mvr_cloud_copy(self._h, dst, src), self._h)

    def append(self, dst, src):
        _chk(_lib.mvr_cloud_append(self._h, dst, src), self._h)

    def clear(self, slot):
        _chk(_lib.mvr_cloud_clear(self._h, slot), self._h)

    def transform(self, dst, src, T):
        """getTransformedPoints semantics (f64 pose)."""
        t = to_cm(T, np.float64)
        _chk(_lib.mvr_cloud_transform(self._h, dst, src, _p(t, C.c_double)), self._h)

    def transform_f32(self, dst, src, T):
        """pcl::transformPointCloud semantics (f32 pose)."""
        t = to_cm(T, np.float32)
        _chk(_lib.mvr_cloud_transform_f32(self._h, dst, src, _p(t, C.c_float)), self._h)

    # -- hot path
    def nn(self, q_slot, t_slot, fma=False):
        n = self.size(q_slot)
        idx, d2 = np.empty(n, np.uint32), np.empty(n, np.float32)
        _chk(_lib.mvr_nn(self._h, q_slot, t_slot, int(fma), _p(idx, C.c_uint32), _p(d2, C.c_float)), self._h)
        return idx, d2

    def denoise(self, slot, segment_threshold=10, triangle_length=2.5):
        """PointCloud::denoise in place.  Returns (kept original indices in output order, number of components)."""
        n = self.size(slot)
        idx = np.empty(max(n, 1), np.uint32)
        kept, comps = C.c_size_t(), C.c_size_t()
        _chk(_lib.mvr_cloud_denoise(self._h, slot, int(segment_threshold), float(triangle_length), C.byref(kept), C.byref(comps),
                                    _p(idx, C.c_uint32)), self._h)
        return idx[:kept.value].copy(), comps.value

    # ---- target sharding over ranks (sequential mode): see seq.py
    def set_global_base(self, slot, global_begin):
        _chk(_lib.mvr_cloud_set_global_base(self._h, slot, int(global_begin)), self._h)

    def append_range(self, dst, src, src_begin, count, global_begin):
        _chk(_lib.mvr_cloud_append_range(self._h, dst, src, int(src_begin), int(count), int(global_begin)), self._h)

    def nn_forward_keys(self, s, t, max_dist, dev_ptr, fma=False):
        """forward NN of slot s in the (shard) slot t -> Ns signed 64-bit keys with GLOBAL target indices at dev_ptr"""
        _chk(_lib.mvr_nn_forward_keys(self._h, s, t, float(max_dist), int(fma), _vp(dev_ptr)), self._h)

    def pair_moments2_from_keys(self, s, t, keys_dev_ptr, max_dist, origin, out_dev_ptr, reciprocal=True, fma=False):
        o = np.ascontiguousarray(origin, np.float64)
        _chk(_lib.mvr_pair_moments2_from_keys(self._h, s, t, _vp(keys_dev_ptr), float(max_dist), int(reciprocal), int(fma),
                                              _p(o, C.c_double), _vp(out_dev_ptr)), self._h)

    def correspondences(self, s, t, max_dist, reciprocal=True, fma=False):
        n = self.size(s)
        q, m, d = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32)
        cnt = _sz()
        _chk(_lib.mvr_correspondences(self._h, s, t, float(max_dist), int(reciprocal), int(fma),
                                      _p(q, C.c_int32), _p(m, C.c_int32), _p(d, C.c_float), n,
                                      C.byref(cnt)), self._h)
        k = cnt.value
        return q[:k].copy(), m[:k].copy(), d[:k].copy()

    def pair_batch_correspondences(self, k, cap):
        """(query, match, d2) of pair k of the last fused batch / ring step (mvr_pair_batch_correspondences)"""
        q, m, d = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)
        cnt = _sz()
        _chk(_lib.mvr_pair_batch_correspondences(self._h, int(k), _p(q, C.c_int32), _p(m, C.c_int32), _p(d, C.c_float), cap, C.byref(cnt)), self._h)
        n = min(cnt.value, cap)
        return q[:n].copy(), m[:n].copy(), d[:n].copy()

    def pair_moments(self, s, t, max_dist, reciprocal=True, fma=False) -> PairMoments:
        out = PairMoments()
        _chk(_lib.mvr_pair_moments(self._h, s, t, float(max_dist), int(reciprocal), int(fma),
                                   C.byref(out)), self._h)
        return out

    def pair_moments2(self, s, t, max_dist, origin, reciprocal=True, fma=False, q_begin=0,
                      q_count=None) -> PairMoments2:
        out = PairMoments2()
        o = np.ascontiguousarray(origin, np.float64)
        qc = (1 << 62) if q_count is None else int(q_count)
        _chk(_lib.mvr_pair_moments2(self._h, s, t, float(max_dist), int(reciprocal), int(fma),
                                    int(q_begin), qc, _p(o, C.c_double), C.byref(out)), self._h)
        return out

    def pair_moments2_dev(self, s, t, max_dist, origin, dev_ptr, reciprocal=True, fma=False,
                          q_begin=0, q_count=None):
        o = np.ascontiguousarray(origin, np.float64)
        qc = (1 << 62) if q_count is None else int(q_count)
        _chk(_lib.mvr_pair_moments2_dev(self._h, s, t, float(max_dist), int(reciprocal), int(fma),
                                        int(q_begin), qc, _p(o, C.c_double), _vp(dev_ptr)), self._h)

    def transform_batch(self, dst_slots, src_slots, poses):
        """slot dst[k] = pose[k] applied to slot src[k] (mvr_cloud_transform semantics), all clouds in one launch."""
        n = len(dst_slots)
        d = (C.c_int * n)(*[int(v) for v in dst_slots])
        s = (C.c_int * n)(*[int(v) for v in src_slots])
        # (4,4) column-vector matrices -> column-major 16-vectors: one transpose for all clouds
        T = (np.ascontiguousarray(np.asarray(poses, np.float64).reshape(n, 4, 4).transpose(0, 2, 1)).reshape(n, 16) if n
             else np.zeros((0, 16)))
        _chk(_lib.mvr_cloud_transform_batch(self._h, n, d, s, _p(T, C.c_double)), self._h)

    def ring_step(self, posed_slots, raw_slots, edges, poses, max_dist, origin, lum_iterations=16, reciprocal=True, fma=False,
                  steps=None, _entry=None):
        """mvr_ring_step: one outer pass of registrationLUM in one native call (single process).  edges: [(src view,
        tgt view)]; poses: list of (4,4) float64.  Returns (new poses, info) like ring_host_step, plus info["rows"]
        (ne x 32 edge table) and info["timing_ms"] = (enqueue, GPU wait + copy, host solve).  poses may be a list of
        (4,4) arrays or one (V,4,4) array; the new poses come back as a (V,4,4) array.  steps=K: K passes in one
        native call (mvr_ring_run)."""
        V, ne = len(posed_slots), len(edges)
        key = (tuple(posed_slots), tuple(raw_slots), tuple(edges))
        st = getattr(self, "_ring_static", None)
        if st is None or st[0] != key:                      # the slot / edge arrays of a registration never change
            st = (key, (C.c_int * V)(*[int(v) for v in posed_slots]), (C.c_int * V)(*[int(v) for v in raw_slots]),
                  (C.c_int * ne)(*[int(e[0]) for e in edges]), (C.c_int * ne)(*[int(e[1]) for e in edges]))
            self._ring_static = st
        _, ps, rs, es, et = st
        # (4,4) column-vector matrices <-> the library's column-major 16-vectors: one transpose for all views
        P = np.ascontiguousarray(np.asarray(poses, np.float64).transpose(0, 2, 1)).reshape(V, 16)
        o = np.ascontiguousarray(origin, np.float64)
        lum, rows, tm = np.zeros((V, 6)), np.empty((ne, 32)), np.zeros(3)
        pT, pn, pm, its = np.empty((ne, 16), np.float32), np.empty(ne), np.empty(ne), C.c_int()
        args = (V, ps, rs, ne, es, et, float(max_dist), int(reciprocal), int(fma), _p(o, C.c_double), int(lum_iterations),
                _p(P, C.c_double), _p(lum, C.c_double), _p(pT, C.c_float), _p(pn, C.c_double), _p(pm, C.c_double), C.byref(its),
                _p(rows, C.c_double), _p(tm, C.c_double))
        if steps is None:
            _chk(_lib.mvr_ring_step(self._h, *args), self._h)
        else:                     # mvr_ring_run: `steps` passes in one call; outputs of the last one, timing summed
            _chk((_entry or _lib.mvr_ring_run)(self._h, int(steps), *args), self._h)
        new = np.ascontiguousarray(P.reshape(V, 4, 4).transpose(0, 2, 1))        # (V,4,4): indexable like the list that came in
        info = dict(pair_T=np.ascontiguousarray(pT.reshape(ne, 4, 4).transpose(0, 2, 1)), pair_n=pn.tolist(),
                    pair_mse=pm.tolist(), lum_pose=lum, lum_iterations=its.value, rows=rows, timing_ms=tuple(tm.tolist()))
        return new, info

    # ---- multi-GPU (mvr_world.cpp): one process per GPU
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """ncclCommInitRank on this context's device (collective: every rank calls it with rank 0's unique id)."""
        assert len(unique_id) == 128
        _torch_before_rccl()
        _chk(_lib.mvr_ctx_comm_init(self._h, unique_id, int(rank), int(world)), self._h)

    def project(self, world, rank=0, peer_rows=None):
        """mvr_ctx_project: ring_run_sharded then runs as `rank` of `world` on this one GPU, the absent peers' rows added behind
        the all-reduce (peer_rows: (ne, 32) float64); world <= 1 switches it off"""
        if world <= 1:
            _chk(_lib.mvr_ctx_project(self._h, 0, 0, None, 0), self._h)
            return
        rows = np.ascontiguousarray(peer_rows, np.float64)
        _chk(_lib.mvr_ctx_project(self._h, int(world), int(rank), _p(rows, C.c_double), rows.shape[0]), self._h)

    def debug_order(self, slot):
        """diagnostics: the ordering of the cloud in `slot` (sorted position -> original index), or None while it has none"""
        n = C.c_size_t()
        _chk(_lib.mvr_debug_order(self._h, int(slot), None, 0, C.byref(n)), self._h)
        if n.value == 0:
            return None
        buf = np.zeros(n.value, np.uint32)
        _chk(_lib.mvr_debug_order(self._h, int(slot), buf.ctypes.data, n.value, C.byref(n)), self._h)
        return buf

    def comm_destroy(self):
        _chk(_lib.mvr_ctx_comm_destroy(self._h), self._h)

    def comm_info(self):
        """(rank, world, ranks RCCL itself reports for the communicator; 0 = no communicator)"""
        r, w, n = C.c_int(), C.c_int(), C.c_int()
        _chk(_lib.mvr_ctx_comm_info(self._h, C.byref(r), C.byref(w), C.byref(n)), self._h)
        return r.value, w.value, n.value

    def ring_run_sharded(self, posed_slots, raw_slots, edges, poses, max_dist, origin, steps=1, lum_iterations=16, reciprocal=True,
                         fma=False):
        """mvr_ring_run_sharded: this rank's share of `steps` outer passes + one RCCL all-reduce of the edge table per
        pass; same arguments and results as ring_step(steps=...) on every rank."""
        return self.ring_step(posed_slots, raw_slots, edges, poses, max_dist, origin, lum_iterations=lum_iterations,
                              reciprocal=reciprocal, fma=fma, steps=steps, _entry=_lib.mvr_ring_run_sharded)

    def ring_rows_sharded(self, rank, world, posed_slots, raw_slots, edges, poses, max_dist, origin, reciprocal=True, fma=False):
        """mvr_ring_rows_sharded: the rows of the edge table that `rank` of `world` contributes to one pass (host, (ne, 32))"""
        V, ne = len(posed_slots), len(edges)
        ps, rs = (C.c_int * V)(*[int(v) for v in posed_slots]), (C.c_int * V)(*[int(v) for v in raw_slots])
        es, et = (C.c_int * ne)(*[int(e[0]) for e in edges]), (C.c_int * ne)(*[int(e[1]) for e in edges])
        P = np.ascontiguousarray(np.asarray(poses, np.float64).transpose(0, 2, 1)).reshape(V, 16)
        o, rows = np.ascontiguousarray(origin, np.float64), np.zeros((ne, 32))
        _chk(_lib.mvr_ring_rows_sharded(self._h, int(rank), int(world), V, ps, rs, ne, es, et, float(max_dist), int(reciprocal), int(fma),
                                        _p(o, C.c_double), _p(P, C.c_double), _p(rows, C.c_double)), self._h)
        return rows

    def pair_moments2_batch(self, pairs, max_dist, origin, dev_ptr=None, reciprocal=True, fma=False, ranges=None):
        """All scan pairs of one global iteration in one call: one launch per stage for all pairs (culled
        search) or one pair per worker stream (brute force).  pairs: [(src, tgt)]; ranges: [(q_begin, q_count)] or None.
        Returns a list of PairMoments2 (host) or, with dev_ptr (device [n][32] float64), None."""
        n = len(pairs)
        src = (C.c_int * n)(*[int(a) for a, _ in pairs])
        dst = (C.c_int * n)(*[int(b) for _, b in pairs])
        qb = qn = None
        if ranges is not None:
            qb = (C.c_size_t * n)(*[int(a) for a, _ in ranges])
            qn = (C.c_size_t * n)(*[(1 << 62) if c is None else int(c) for _, c in ranges])
        o = np.ascontiguousarray(origin, np.float64)
        out = None if dev_ptr else (PairMoments2 * n)()
        _chk(_lib.mvr_pair_moments2_batch(self._h, n, src, dst, float(max_dist), int(reciprocal), int(fma), qb, qn,
                                          _p(o, C.c_double), out, _vp(dev_ptr) if dev_ptr else None), self._h)
        return None if dev_ptr else list(out)

    def pair_moments2_from_corr(self, s, t, query, match, origin) -> PairMoments2:
        q = np.ascontiguousarray(query, np.int32)
        m = np.ascontiguousarray(match, np.int32)
        o = np.ascontiguousarray(origin, np.float64)
        out = PairMoments2()
        _chk(_lib.mvr_pair_moments2_from_corr(self._h, s, t, _p(q, C.c_int32), _p(m, C.c_int32), len(q),
                                              _p(o, C.c_double), C.byref(out)), self._h)
        return out

    def icp_align(self, src, tgt, out, params: IcpParams):
        """Returns (T (4,4) float32, stats dict, rc); rc is OK or E_NOCORR."""
        T, st = np.empty(16, np.float32), IcpStats()
        rc = _chk(_lib.mvr_icp_align(self._h, src, tgt, out, C.byref(params), _p(T, C.c_float),
                                     C.byref(st)), self._h, allow=(E_NOCORR,))
        stats = dict(iterations=st.iterations, converged=bool(st.converged),
                     state=CONV_STATES[st.state], n_corr=st.n_corr, mse=st.mse, evals=st.evals,
                     fwd_queries=st.fwd_queries, ms=st.ms)
        return from_cm(T), stats, rc

    @staticmethod
    def _stats(st):
        return dict(iterations=st.iterations, converged=bool(st.converged), state=CONV_STATES[st.state], n_corr=st.n_corr, mse=st.mse,
                    evals=st.evals, fwd_queries=st.fwd_queries, ms=st.ms)

    def seq_align_sharded(self, src, tgt, out, params: IcpParams, origin):
        """mvr_seq_align_sharded: one align of the full source against this rank's target shard -> (T, stats, rc)"""
        T, st, o = np.empty(16, np.float32), IcpStats(), np.ascontiguousarray(origin, np.float64)
        rc = _chk(_lib.mvr_seq_align_sharded(self._h, src, tgt, out, C.byref(params), _p(o, C.c_double), _p(T, C.c_float), C.byref(st)),
                  self._h, allow=(E_NOCORR,))
        return from_cm(T), self._stats(st), rc

    def seq_run_sharded(self, raw_slots, target_slot, source_slot, out_slot, params: IcpParams, origin, poses, repeat=1):
        """mvr_seq_run_sharded: registrationICP with the growing target sharded over the context's communicator.
        poses: V (4,4) float64 -> (new poses (V,4,4), log: list of dict(view, T, iterations, state, n_corr, mse, ...))"""
        V = len(raw_slots)
        rs = (C.c_int * V)(*[int(v) for v in raw_slots])
        P = np.ascontiguousarray(np.asarray(poses, np.float64).transpose(0, 2, 1)).reshape(V, 16)
        cap = max(1, repeat * (V - 1))
        views, Ts, sts, n = (C.c_int * cap)(), np.zeros((cap, 16), np.float32), (IcpStats * cap)(), C.c_int()
        o = np.ascontiguousarray(origin, np.float64)
        _chk(_lib.mvr_seq_run_sharded(self._h, V, rs, int(target_slot), int(source_slot), int(out_slot), C.byref(params), _p(o, C.c_double),
                                      int(repeat), _p(P, C.c_double), views, _p(Ts, C.c_float), sts, C.byref(n)), self._h)
        log = [dict(self._stats(sts[k]), view=views[k], T=from_cm(Ts[k])) for k in range(n.value)]
        return np.ascontiguousarray(P.reshape(V, 4, 4).transpose(0, 2, 1)), log

    def seq_run(self, raw_slots, target_slot, source_slot, out_slot, params: IcpParams, poses, repeat=1):
        """mvr_seq_run: registrationICP (registrator.cpp:526-588) on one GPU as one native call -- `repeat` sweeps of aligns of
        views 1, V-1, 2, ... against the growing model.  poses: V (4,4) float64 -> (new poses (V,4,4), log as seq_run_sharded)"""
        V = len(raw_slots)
        rs = (C.c_int * V)(*[int(v) for v in raw_slots])
        P = np.ascontiguousarray(np.asarray(poses, np.float64).transpose(0, 2, 1)).reshape(V, 16)
        cap = max(1, repeat * (V - 1))
        views, Ts, sts, n = (C.c_int * cap)(), np.zeros((cap, 16), np.float32), (IcpStats * cap)(), C.c_int()
        _chk(_lib.mvr_seq_run(self._h, V, rs, int(target_slot), int(source_slot), int(out_slot), C.byref(params), int(repeat),
                              _p(P, C.c_double), views, _p(Ts, C.c_float), sts, C.byref(n)), self._h)
        log = [dict(self._stats(sts[k]), view=views[k], T=from_cm(Ts[k])) for k in range(n.value)]
        return np.ascontiguousarray(P.reshape(V, 4, 4).transpose(0, 2, 1)), log

    def fitness(self, inp, tgt, T, max_range=np.finfo(np.float64).max, fma=False) -> float:
        t, s = to_cm(T, np.float32), C.c_double()
        _chk(_lib.mvr_fitness(self._h, inp, tgt, _p(t, C.c_float), float(max_range), int(fma),
                              C.byref(s)), self._h)
        return s.value

    # -- misc
    def sync(self):
        _chk(_lib.mvr_ctx_sync(self._h), self._h)

    def device_info(self):
        name = C.create_string_buffer(64)
        ncu, mhz = C.c_int(), C.c_int()
        _chk(_lib.mvr_device_info(self._h, name, 64, C.byref(ncu), C.byref(mhz)), self._h)
        return name.value.decode(), ncu.value, mhz.value

    def tune(self, **kw):
        """NN launch knobs (nn_q, nn_sub, nn_blocks_per_cu); results do not depend on them."""
        for k, v in kw.items():
            _chk(_lib.mvr_ctx_tune(self._h, k.encode(), int(v)), self._h)

    def stat(self, key) -> float:
        """mvr_ctx_stat: "piped_passes", "fused_passes", "blocking_events", "pipeline" """
        v = C.c_double()
        _chk(_lib.mvr_ctx_stat(self._h, key.encode(), C.byref(v)), self._h)
        return v.value

    def pass_log(self):
        """wall ms of every pass of the last ring run (mvr_ctx_pass_log)"""
        n = C.c_int()
        _chk(_lib.mvr_ctx_pass_log(self._h, None, 0, C.byref(n)), self._h)
        ms = np.zeros(max(n.value, 1))
        _chk(_lib.mvr_ctx_pass_log(self._h, _p(ms, C.c_double), n.value, C.byref(n)), self._h)
        return ms[: n.value].tolist()

    def debug_counters(self, reset=False):
        out = (C.c_uint64 * 4)()
        _chk(_lib.mvr_debug_counters(self._h, out, int(reset)), self._h)
        return list(out)

    def prof_enable(self, on=True):
        _chk(_lib.mvr_prof_enable(self._h, int(on)), self._h)

    def prof_reset(self):
        _chk(_lib.mvr_prof_reset(self._h), self._h)

    def prof_get(self, family):
        n, ms, w = C.c_uint64(), C.c_double(), C.c_double()
        _chk(_lib.mvr_prof_get(self._h, family, C.byref(n), C.byref(ms), C.byref(w)), self._h)
        return n.value, ms.value, w.value


# ------------------------------------------------------------------ multi-GPU

def _torch_before_rccl():
    """The library's collectives run on PyTorch-ROCm's copy of RCCL when this process can import torch (MVR_RCCL_LIB above).  A
    process that initialises that RCCL through the library and imports torch only AFTERWARDS aborts when it exits ("double free or
    corruption": the two libraries' exit handlers then run in the wrong order -- DESIGN.md section 10); imported first, never.  So the
    first multi-GPU entry point imports it, if it is there and this RCCL is the one in use."""
    if "torch" in sys.modules or "torch" not in os.environ.get("MVR_RCCL_LIB", ""):
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def comm_unique_id() -> bytes:
    """ncclGetUniqueId (rank 0); hand the 128 bytes to the other ranks by the launcher's means."""
    _torch_before_rccl()
    buf = C.create_string_buffer(128)
    _chk(_lib.mvr_comm_unique_id(buf))
    return buf.raw


def rccl_library() -> str:
    return _lib.mvr_rccl_library().decode()


def pool_trim():
    """give the idle blocks of the library's allocation cache back to the runtime (mvr_pool_trim); returns
    dict(freed_bytes, cached_bytes, hits, misses)"""
    st = (C.c_uint64 * 3)()
    freed = _lib.mvr_pool_trim(st)
    return dict(freed_bytes=int(freed), cached_bytes=int(st[0]), hits=int(st[1]), misses=int(st[2]))


def ring_segments(edge_queries, world, rank):
    """the sharding of the ring pass: [(edge, first query, count)] of `rank` (mvr_ring_segments)"""
    ne = len(edge_queries)
    q = (C.c_size_t * max(ne, 1))(*[int(v) for v in edge_queries])
    se, sb, sc, n = (C.c_int * max(ne, 1))(), (C.c_size_t * max(ne, 1))(), (C.c_size_t * max(ne, 1))(), C.c_int()
    _chk(_lib.mvr_ring_segments(ne, q, int(world), int(rank), se, sb, sc, C.byref(n)))
    return [(se[k], sb[k], sc[k]) for k in range(n.value)]


class _BorrowedContext(Context):
    """a rank's context of a World: owned by the world, never destroyed from here"""

    def __init__(self, handle):
        self._h = handle

    def close(self):
        self._h = None


class World:
    """One process, n GPUs (mvr_world_create: one context per device + ncclCommInitAll)."""

    def __init__(self, n_dev, device_ids=None):
        h = _vp()
        ids = (C.c_int * n_dev)(*device_ids) if device_ids is not None else None
        _torch_before_rccl()
        rc = _lib.mvr_world_create(C.byref(h), int(n_dev), ids)
        if rc != OK:
            raise MvrError(rc, "mvr_world_create(%d): %s" % (n_dev, rccl_library()))
        self._h, self.n = h, n_dev

    def ctx(self, rank) -> Context:
        return _BorrowedContext(_vp(_lib.mvr_world_ctx(self._h, int(rank))))

    def upload(self, slot, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        rc = _lib.mvr_world_upload(self._h, int(slot), _p(pts, C.c_float), len(pts), 4 * pts.shape[1])
        if rc != OK:
            raise MvrError(rc, _lib.mvr_world_last_error(self._h).decode())

    def ring_run(self, posed_slots, raw_slots, edges, poses, max_dist, origin, steps=1, lum_iterations=16, reciprocal=True, fma=False):
        V, ne = len(posed_slots), len(edges)
        ps, rs = (C.c_int * V)(*[int(v) for v in posed_slots]), (C.c_int * V)(*[int(v) for v in raw_slots])
        es, et = (C.c_int * ne)(*[int(e[0]) for e in edges]), (C.c_int * ne)(*[int(e[1]) for e in edges])
        P = np.ascontiguousarray(np.asarray(poses, np.float64).transpose(0, 2, 1)).reshape(V, 16)
        o = np.ascontiguousarray(origin, np.float64)
        lum, rows, tm = np.zeros((V, 6)), np.empty((ne, 32)), np.zeros(3)
        pT, pn, pm, its = np.empty((ne, 16), np.float32), np.empty(ne), np.empty(ne), C.c_int()
        rc = _lib.mvr_world_ring_run(self._h, int(steps), V, ps, rs, ne, es, et, float(max_dist), int(reciprocal), int(fma),
                                     _p(o, C.c_double), int(lum_iterations), _p(P, C.c_double), _p(lum, C.c_double), _p(pT, C.c_float),
                                     _p(pn, C.c_double), _p(pm, C.c_double), C.byref(its), _p(rows, C.c_double), _p(tm, C.c_double))
        if rc != OK:
            raise MvrError(rc, _lib.mvr_world_last_error(self._h).decode())
        new = np.ascontiguousarray(P.reshape(V, 4, 4).transpose(0, 2, 1))
        return new, dict(pair_T=np.ascontiguousarray(pT.reshape(ne, 4, 4).transpose(0, 2, 1)), pair_n=pn.tolist(), pair_mse=pm.tolist(),
                         lum_pose=lum, lum_iterations=its.value, rows=rows, timing_ms=tuple(tm.tolist()))

    def close(self):
        if self._h:
            _lib.mvr_world_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
