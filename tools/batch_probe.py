#!/usr/bin/env python3
"""tools/batch_probe.py -- the fused global pass (mvr_pair_moments2_batch on the 12 x 200k ring) for rocprofv3 counter
passes: `reps` batches after one warm-up.  Prints one JSON line with the NN launches' HIP-event time."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
knobs = dict(kv.split("=") for kv in sys.argv[4:])
mvr = importlib.import_module("multi-view-registration_amd")
sp = mvr.synth_params(V, 3)
piv, ax = mvr.synth_prior(sp)
with mvr.Context(0) as ctx:
    ctx.tune(**{k: int(v) for k, v in knobs.items()})
    for v in range(V):
        ctx.upload(V + v, mvr.synth_view(sp, v, n))
    poses = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    ctx.transform_batch(list(range(V)), [V + v for v in range(V)], poses)
    pairs = [(v, (v + 1) % V) for v in range(V)]
    ctx.pair_moments2_batch(pairs, 4.0, np.array(sp.pivot))
    ctx.prof_reset(); ctx.prof_enable(1)
    for _ in range(reps):
        out = ctx.pair_moments2_batch(pairs, 4.0, np.array(sp.pivot))
    ctx.prof_enable(False)
    launches, ms, evals = ctx.prof_get(mvr.K_NN)
    print(json.dumps(dict(views=V, n=n, reps=reps, knobs=knobs, nn_launches=launches, nn_ms=ms, nn_evals=evals, ms_per_launch=ms / max(launches, 1),
                          n_corr=[int(m.n) for m in out])))
