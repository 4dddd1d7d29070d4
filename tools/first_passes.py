#!/usr/bin/env python3
"""tools/first_passes.py [views] [points] -- wall time of each of the first ring passes after the uploads (the first is
unseeded and runs on the culled kernel; the second builds the scans' cell grids), with and without the grid search."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
mvr = importlib.import_module("multi-view-registration_amd")
sp = mvr.synth_params(V, 3)
piv, ax = mvr.synth_prior(sp)
origin = np.array(sp.pivot)
scans = [mvr.synth_view(sp, v, n) for v in range(V)]
for mode in (0, 1, 0, 1):
    with mvr.Context(0) as ctx:
        ctx.tune(ring_search=mode)
        for v in range(V):
            ctx.upload(V + v, scans[v])
        poses = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
        edges = [(v, (v + 1) % V) for v in range(V)]
        ctx.sync()
        ms = []
        for k in range(6):
            t0 = time.perf_counter()
            poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, 4.0, origin)
            ctx.sync()
            ms.append(round(1e3 * (time.perf_counter() - t0), 3))
        print(json.dumps(dict(views=V, n=n, ring_search=mode, ms_per_pass=ms, n_corr=sum(info["pair_n"]))))
