#!/usr/bin/env python3
"""tools/host_step_bench.py -- the HOST side of a global pass (mvr_ring_host_step: LUM's 16 iterations on the all-reduced edge
table, registrator.cpp:650-662), timed alone on this machine's CPU with the table of a real 12 x 200k pass.  The GPU waits for it
between two passes, so it is part of the step (bench.py: step_breakdown_ms.host_solve).

    python tools/host_step_bench.py [views] [points] [calls]        (MVR_LIB_VARIANT=<name> times build/libmvr_hip_<name>.so)
Prints one JSON line: median / min microseconds per call, with and without the per-pair transformations."""
import ctypes as C, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    mvr = importlib.import_module("multi-view-registration_amd")
    sp = mvr.synth_params(V, 3)
    piv, ax = mvr.synth_prior(sp)
    origin = np.array(sp.pivot)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    edges = [(v, (v + 1) % V) for v in range(V)]
    with mvr.Context(0) as ctx:
        for v in range(V):
            ctx.upload(V + v, mvr.synth_view(sp, v, n))
        _, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses0, 4.0, origin)
        rows = np.ascontiguousarray(info["rows"])
    L = mvr._lib
    ne = len(edges)
    es = (C.c_int * ne)(*[e[0] for e in edges]); et = (C.c_int * ne)(*[e[1] for e in edges])
    o = np.ascontiguousarray(origin, np.float64)
    P0 = np.ascontiguousarray(np.asarray(poses0, np.float64).reshape(V, 4, 4).transpose(0, 2, 1)).reshape(V, 16)
    lum = np.zeros((V, 6)); pT = np.empty((ne, 16), np.float32); pn = np.empty(ne); pm = np.empty(ne); its = C.c_int()
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    out = dict(views=V, points=n, variant=os.environ.get("MVR_LIB_VARIANT", ""), lum_iterations=16)
    for name, with_T in (("with_pair_T", True), ("without_pair_T", False)):
        ts = []
        for _ in range(calls):
            P = P0.copy()
            t0 = time.perf_counter()
            L.mvr_ring_host_step(V, ne, es, et, rows.ctypes.data_as(dp), o.ctypes.data_as(dp), 16, P.ctypes.data_as(dp), lum.ctypes.data_as(dp),
                                 pT.ctypes.data_as(fp) if with_T else None, pn.ctypes.data_as(dp), pm.ctypes.data_as(dp), C.byref(its))
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out[name] = dict(median_us=round(1e6 * ts[len(ts) // 2], 2), min_us=round(1e6 * ts[0], 2))
    out["pose_checksum"] = float(np.abs(P).sum())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
