#!/usr/bin/env python3
"""tools/seq_bench.py -- BASELINE configs[2]: 12-view turntable ring, 200k pts/scan,
SEQUENTIAL pairwise ICP against the growing target (Registrator::registrationICP,
mvr/src/registrator.cpp:526-588, order 1,11,2,10,...,6), device-resident on one
MI355X, with the CPU oracle (kd-tree) timed beside it on the same inputs.

    python tools/seq_bench.py [--points 200000] [--views 12] [--repeat 1] [--no-cpu]

Prints one JSON line: ms per align (GPU, both exact kernels), CPU ms per align,
final-pose deltas GPU vs oracle.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gpu_sweep(mvr, ctx, scans, poses0, params, order, repeat):
    V = len(scans)
    RAW, TARGET, SOURCE, OUT = 16, 0, 1, 2
    if not getattr(ctx, "_seq_bench_uploaded", False):       # the scans are uploaded ONCE (a re-upload is a new point set: new ordering, new grid)
        for v in range(V):
            ctx.upload(RAW + v, scans[v])
        ctx._seq_bench_uploaded = True
    poses = [p.copy() for p in poses0]
    log = []
    ctx.sync()
    t0 = time.perf_counter()
    if os.environ.get("MVR_SEQ_NATIVE", "1") != "0":                 # the whole run as ONE native call (mvr_seq_run)
        nt = len(scans[0])
        poses, nlog = ctx.seq_run([RAW + v for v in range(V)], TARGET, SOURCE, OUT, params, poses, repeat=repeat)
        for k, e in enumerate(nlog):
            nt = len(scans[0]) * (k % len(order) + 2)
            log.append(dict(view=e["view"], n_corr=e["n_corr"], mse=e["mse"], ms=e["ms"], evals=e["evals"], nt=nt))
        poses = [np.array(p) for p in poses]
        repeat = 0
    for _ in range(repeat):
        ctx.transform(TARGET, RAW + 0, poses[0])                       # registrator.cpp:562
        ctx.reserve(TARGET, V * len(scans[0]))
        for v in order:
            ctx.transform(SOURCE, RAW + v, poses[v])                   # :565
            T, st, rc = ctx.icp_align(SOURCE, TARGET, OUT, params)     # :566-569
            poses[v] = mvr.mat4d_mul(T.astype(np.float64), poses[v])   # :573-574
            ctx.append(TARGET, OUT)                                    # :576
            log.append(dict(view=v, n_corr=st["n_corr"], mse=st["mse"], ms=st["ms"], evals=st["evals"], nt=ctx.size(TARGET)))
    ctx.sync()
    return poses, log, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--max-dist", type=float, default=4.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-brute", action="store_true")
    a = ap.parse_args()
    mvr = importlib.import_module("multi-view-registration_amd")
    import ref_driver
    V, N = a.views, a.points
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    order = ref_driver.view_order(V)
    params = mvr.icp_params(max_dist=a.max_dist, max_iter=1000)           # registrator.cpp:551-560
    out = dict(config="%d-view ring, %d pts/scan, sequential pairwise ICP vs growing target, repeat %d" % (V, N, a.repeat),
               aligns=len(order) * a.repeat)
    with mvr.Context(0) as ctx:
        for kv in os.environ.get("MVR_SEQ_KNOBS", "").split():             # e.g. MVR_SEQ_KNOBS="lazy_super=0 align_spin=0"
            ctx.tune(**{kv.split("=")[0]: int(kv.split("=")[1])})
        for name, mode in (("culled", 1),) + ((("brute", 0),) if not a.no_brute else ()):
            ctx.tune(nn_mode=mode)
            gpu_sweep(mvr, ctx, scans, poses0, params, order, 1)                     # warm-up (allocations, sorts)
            # the warm-up aligned the same scans from the same poses: what it left for the next aligns' searches to start from
            # (seq_seed) is forgotten, so that the timed run's FIRST sweep is a first sweep and only its later ones are seeded
            seeds = int(os.environ.get("MVR_SEQ_SEED", "1"))
            ctx.tune(seq_seed=0)
            if seeds:
                ctx.tune(seq_seed=1)
            poses, log, dt = gpu_sweep(mvr, ctx, scans, poses0, params, order, a.repeat)
            per = len(order)
            out["gpu_%s" % name] = dict(total_s=dt, ms_per_align=1e3 * dt / len(log), queries_per_s=N * len(log) / dt,
                                        native_ms_per_align_by_sweep=[round(sum(e["ms"] for e in log[i:i + per]) / per, 4) for i in range(0, len(log), per)],
                                        evals=sum(e["evals"] for e in log), last_nt=log[-1]["nt"],
                                        n_corr=[e["n_corr"] for e in log][:len(order)])
            out["poses_%s" % name] = poses
    if not a.no_cpu:
        import oracle as orc
        t0 = time.perf_counter()
        oposes, olog = ref_driver.sequential_icp(orc, scans, poses0, orc.make_params(max_dist=a.max_dist, max_iter=1000), V,
                                                 repeat=a.repeat, fitness_last=False)
        dt = time.perf_counter() - t0
        out["cpu_oracle_kdtree_1thread"] = dict(total_s=dt, ms_per_align=1e3 * dt / len(olog), queries_per_s=N * len(olog) / dt,
                                                n_corr=[e["n_corr"] for e in olog][:len(order)])
        for name in ("culled", "brute"):
            if "poses_%s" % name in out:
                g = out["poses_%s" % name]
                out["pose_delta_%s_vs_oracle" % name] = dict(
                    rot=max(float(np.abs(g[v][:3, :3] - oposes[v][:3, :3]).max()) for v in range(V)),
                    trans_mm=max(float(np.abs(g[v][:3, 3] - oposes[v][:3, 3]).max()) for v in range(V)))
    for k in [k for k in out if k.startswith("poses_")]:
        del out[k]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
