#!/usr/bin/env python3
"""tools/cold_probe.py [views] [points] [passes] [knob=value ...] -- a COLD registration: fresh context, uploads, then
`passes` ring passes from the prior in ONE native call per pass (so that each pass's wall time is seen) -- what the
first passes of a registration cost before the steady state bench.py times.  One JSON line.
Under rocprofv3 --hip-trace --kernel-trace --stats it shows where the host time of the first two passes goes."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if "=" not in a]
knobs = dict(kv.split("=") for kv in sys.argv[1:] if "=" in kv)
V = int(args[0]) if len(args) > 0 else 12
n = int(args[1]) if len(args) > 1 else 200000
passes = int(args[2]) if len(args) > 2 else 6
one_call = int(knobs.pop("one_call", 0))
mvr = importlib.import_module("multi-view-registration_amd")
sp = mvr.synth_params(V, 3)
piv, ax = mvr.synth_prior(sp)
origin = np.array(sp.pivot)
scans = [mvr.synth_view(sp, v, n) for v in range(V)]
with mvr.Context(0) as warm:          # (the process's first context pays for the runtime's own start-up: not what is measured)
    warm.upload(0, scans[0][:1000]); warm.sync()
reps = int(knobs.pop("reps", 1))          # reps=2: the registration twice in one process, each on a fresh context -- the second one meets a WARM process (kernels loaded)
for rep in range(reps):
  if rep + 1 == reps and os.environ.get("MVR_TRACE_HOST"):
      sys.stderr.write("[mvr host] ---- measured registration starts\n")
  with mvr.Context(0) as ctx:
      ctx.tune(**{k: int(v) for k, v in knobs.items()})
      t0 = time.perf_counter()
      for v in range(V):
          ctx.upload(V + v, scans[v])
      ctx.sync()
      t_up = time.perf_counter() - t0
      poses = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
      edges = [(v, (v + 1) % V) for v in range(V)]
      ms = []
      t_all = time.perf_counter()
      if one_call:
          poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, 4.0, origin, steps=passes)
          ms = [round(v, 3) for v in ctx.pass_log()]
      else:
          for k in range(passes):
              t0 = time.perf_counter()
              poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, 4.0, origin)
              ms.append(round(1e3 * (time.perf_counter() - t0), 3))
      total = 1e3 * (time.perf_counter() - t_all)
      print(json.dumps(dict(views=V, n=n, passes=passes, knobs=knobs, one_call=one_call, upload_ms=round(1e3 * t_up, 3), ms_per_pass=ms, total_ms=round(total, 3),
                            amortised_ms_per_pass=round(total / passes, 3), n_corr=sum(info["pair_n"]), piped=ctx.stat("piped_passes"),
                            blocking_events=ctx.stat("blocking_events"))))
