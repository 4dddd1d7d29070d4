#!/usr/bin/env python3
"""tools/rank_share_bench.py -- what ONE rank of a world of 2 / 4 / 8 does per pass of the global registration
(registrator.cpp:625-664), measured on the one GPU there is: a PROJECTION of the multi-GPU critical path, not a scaling run.

For each world size W and each rank r: the context plays rank r of W (mvr_ctx_project) -- mvr_ring_run_sharded plans r's
contiguous share of the edges' queries (mvr_ring_segments), runs its chain, really issues the pass's ncclAllReduce on a
communicator of ONE rank (library-owned RCCL, on the library's stream), adds the rows the absent peers would contribute (the sum
of their mvr_ring_rows_sharded tables at the converged poses the run starts from) and solves -- K passes in one native call.
The slowest rank is the projected step of the world.  What the figure leaves out: the fabric (an 8-rank all-reduce of 3 KB over
xGMI is latency, ~10-20 us) and the ranks' skew.  `pipeline` 0 / 1: a pass with a collective over more than one rank is NOT
queued ahead of its poses today (not before a recorded two-GPU run), so 0 is what a real world would run; 1 is what
lifting that rule would give.

    python tools/rank_share_bench.py [views] [points] [steps] [worlds, e.g. 1,2,4,8] [knob=value ...]
Prints one JSON line."""
import gc, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def project(mvr, ctx, V, edges, posed, raw, poses_c, origin, max_d, worlds, steps, warm, all_ranks=True):
    out = {}
    for W in worlds:
        per_rank = []
        ranks = range(W) if all_ranks else [0]
        tables = [ctx.ring_rows_sharded(r, W, posed, raw, edges, poses_c, max_d, origin) for r in range(W)] if W > 1 else []
        for r in ranks:
            if W > 1:
                ctx.project(W, r, sum(t for k, t in enumerate(tables) if k != r))
            else:
                ctx.project(1)
            rec = {}
            for pipe in (0, 1):
                ctx.tune(pipeline=pipe)
                gc.disable()      # (a collection of the caller's heap inside the 3 ms window showed as a slow "rank": bench.py holds 100 MB of scans; no collection here either -- it leaves the host's caches cold for the window)
                try:
                    ctx.ring_run_sharded(posed, raw, edges, poses_c, max_d, origin, steps=warm)          # the rank's buffers, seeds, the pipe
                    ctx.sync()
                    t0 = time.perf_counter()
                    new, info = ctx.ring_run_sharded(posed, raw, edges, poses_c, max_d, origin, steps=steps)
                    ctx.sync()
                    dt = time.perf_counter() - t0
                finally:
                    gc.enable()
                rec["ms_per_step_pipeline%d" % pipe] = 1e3 * dt / steps
                rec["timing_ms_pipeline%d" % pipe] = [t / steps for t in info["timing_ms"]]
                pl = sorted(ctx.pass_log())      # (a single slow pass -- an allocation, a stall of the box -- shows here, not in the mean)
                rec["pass_ms_median_max_pipeline%d" % pipe] = [round(pl[len(pl) // 2], 4), round(pl[-1], 4)] if pl else None
            rec["rank"] = r
            per_rank.append(rec)
        ctx.project(1)
        out[str(W)] = {"ranks": per_rank,
                       "ms_per_step_pipeline0": max(x["ms_per_step_pipeline0"] for x in per_rank),
                       "ms_per_step_pipeline1": max(x["ms_per_step_pipeline1"] for x in per_rank)}
    ctx.tune(pipeline=1)
    return out


def main():
    args = [a for a in sys.argv[1:] if "=" not in a]
    knobs = dict(kv.split("=") for kv in sys.argv[1:] if "=" in kv)
    V = int(args[0]) if len(args) > 0 else 12
    n = int(args[1]) if len(args) > 1 else 200000
    steps = int(args[2]) if len(args) > 2 else 20
    worlds = [int(w) for w in (args[3] if len(args) > 3 else "1,2,4,8").split(",")]
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
        rccl = None
        try:
            ctx.comm_init(mvr.comm_unique_id(), 0, 1)          # a world of one: the collective is really issued
            rccl = ctx.comm_info()
        except Exception as e:                                 # no RCCL: the projection then has no collective in it (and says so)
            rccl = "none (%s)" % e
        poses_c, _ = ctx.ring_run_sharded(posed, raw, edges, poses0, 4.0, origin, steps=40)          # converge first
        res = project(mvr, ctx, V, edges, posed, raw, poses_c, origin, 4.0, worlds, steps, 6)
        print(json.dumps(dict(views=V, n=n, steps=steps, knobs=knobs, rccl=rccl, projected=res,
                              note="one rank's share per pass on ONE GPU, slowest rank per world; the fabric and rank skew are not in it")))


if __name__ == "__main__":
    main()
