#!/usr/bin/env python3
"""tools/step_probe.py -- the bench's timed region without torch, for rocprofv3 counter passes: `warm` ring steps from
the mis-calibrated prior, back to the prior, then `steps` ring steps (mvr_ring_run: posing, all 12 scan pairs in fused
launches, host LUM solve, pose update) -- a real ICP sequence, so that seeded searches see what they see in bench.py.
    python tools/step_probe.py [views] [points] [steps] [warm] [knob=value ...]
Prints one JSON line (ms per step by wall clock, HIP-event time and evaluations of the search launches)."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if "=" not in a]
knobs = dict(kv.split("=") for kv in sys.argv[1:] if "=" in kv)
V = int(args[0]) if len(args) > 0 else 12
n = int(args[1]) if len(args) > 1 else 200000
steps = int(args[2]) if len(args) > 2 else 10
warm = int(args[3]) if len(args) > 3 else 2
if "lib" in knobs:                      # a build variant (tools/build_variant.sh): lib=<name>
    os.environ["MVR_LIB_VARIANT"] = knobs.pop("lib")
mvr = importlib.import_module("multi-view-registration_amd")
sp = mvr.synth_params(V, 3)
piv, ax = mvr.synth_prior(sp)
origin = np.array(sp.pivot)
with mvr.Context(0) as ctx:
    ctx.tune(**{k: int(v) for k, v in knobs.items()})
    for v in range(V):
        ctx.upload(V + v, mvr.synth_view(sp, v, n))
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    edges = [(v, (v + 1) % V) for v in range(V)]
    posed, raw = list(range(V)), [V + v for v in range(V)]
    ctx.ring_step(posed, raw, edges, poses0, 4.0, origin, steps=max(warm, 1))
    ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(int(os.environ.get("MVR_PROBE_PROF", "1")))
    t0 = time.perf_counter()
    new, info = ctx.ring_step(posed, raw, edges, poses0, 4.0, origin, steps=steps)
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    launches, ms, evals = ctx.prof_get(mvr.K_NN)
    print(json.dumps(dict(views=V, n=n, steps=steps, warm=warm, knobs=knobs, lib=os.environ.get("MVR_LIB_VARIANT", "default"), ms_per_step=1e3 * dt / steps, nn_launches=launches, nn_ms=ms,
                          nn_evals=evals, ms_per_launch=ms / max(launches, 1), evals_per_launch=evals / max(launches, 1),
                          n_corr=sum(info["pair_n"]), timing_ms=[t / steps for t in info["timing_ms"]],
                          **({"pass_ms": [round(v, 4) for v in ctx.pass_log()], "piped": ctx.stat("piped_passes")} if os.environ.get("MVR_PROBE_PASSLOG") else {}),
                          **({"stage_waves": {k: ctx.stat("stage_" + k) for k in ("staged", "rows", "width", "points")},
                              "stage_waves_fwd_rev": [[ctx.stat("stage_raw_%d" % (o + k)) for k in (1, 2, 3, 4)] for o in (0, 8)]} if knobs.get("grid_stage_stat") == "1" else {}),
                          **({"stage_clock": {o: [ctx.stat("stage_raw_%d" % (16 + 8 * i + k)) for k in range(8)] for i, o in enumerate(("fallback", "staged"))}} if knobs.get("grid_stage_stat") == "1" and os.environ.get("MVR_LIB_VARIANT") else {}))))
