#!/usr/bin/env python3
"""tools/hang_probe.py -- does the FIRST torch GPU use hang when many library contexts / streams exist already?

Round 1 recorded one GPU test run that "hung for minutes at the first torch use after dozens of library streams
already existed" (tests/conftest.py); no log was kept.  This probe replays that sequence in child processes, each
under its own timeout, and logs what every phase took and which HIP runtime the process mapped:

  scenario   order                                   contexts
  mvr_first  library work, THEN first torch GPU use  5 contexts in a row (closed one after the other), as the
                                                     session-scoped `gpu` fixture creates them
  mvr_alive  the same, contexts kept alive           5 contexts + all their workers alive at the first torch use
  torch_first torch GPU use first                    (the order the round-1 workaround forces)

Each context runs what the parity tests run before the ring tests: a 12-pair culled ring step in 2 groups (worker
stream of the lowest priority), a brute-force batch on 6 worker streams, a single-pair align.

usage: python tools/hang_probe.py [scenario]      (no argument: all scenarios, one child process each)
"""
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hip_libs():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l})


def log(t0, msg):
    print("[%7.2fs] %s" % (time.time() - t0, msg), flush=True)


def library_work(mvr, ctx, t0, tag):
    import numpy as np
    V, N = 12, 20000
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    for v in range(V):
        ctx.upload(V + v, scans[v])
    edges = [(i, (i + 1) % V) for i in range(V)]
    origin = np.array(sp.pivot)
    ctx.tune(nn_mode=1, pair_groups=2)
    new, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, 4.0, origin, steps=3)
    log(t0, "%s: culled ring x3, n_corr %.0f" % (tag, sum(info["pair_n"])))
    ctx.tune(nn_mode=0, pair_streams=6)
    out = ctx.pair_moments2_batch([(s, t) for s, t in edges], 4.0, origin)
    log(t0, "%s: brute batch on 6 worker streams, n %.0f" % (tag, sum(o.n for o in out)))
    ctx.tune(nn_mode=1)
    T, st, rc = ctx.icp_align(1, 0, 40, mvr.icp_params())
    log(t0, "%s: align rc %d n_corr %d" % (tag, rc, st["n_corr"]))


def torch_use(t0, tag):
    import torch
    log(t0, "%s: torch imported" % tag)
    torch.cuda.init()
    log(t0, "%s: torch.cuda.init() done" % tag)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        z = torch.zeros((12, 32), dtype=torch.float64, device="cuda")
        z += 1
    torch.cuda.synchronize()
    log(t0, "%s: torch stream + zeros + sync done, sum %.0f" % (tag, float(z.sum().item())))


def scenario(name):
    t0 = time.time()
    mvr = importlib.import_module("multi-view-registration_amd")
    log(t0, "scenario %s; HIP libs after importing the package: %s" % (name, hip_libs()))
    if name == "torch_first":
        torch_use(t0, "first")
    keep = []
    for k in range(5):
        ctx = mvr.Context(0)
        library_work(mvr, ctx, t0, "ctx %d" % k)
        if name == "mvr_alive":
            keep.append(ctx)
        else:
            ctx.close()
            log(t0, "ctx %d closed" % k)
    torch_use(t0, "after the library")
    # and the library again on a torch stream, as the ring tests do
    import torch
    ts = torch.cuda.Stream()
    ctx = mvr.Context(0, stream=ts.cuda_stream)
    library_work(mvr, ctx, t0, "ctx on a torch stream")
    ctx.close()
    for c in keep:
        c.close()
    log(t0, "scenario %s done; HIP libs: %s" % (name, hip_libs()))


def main():
    if len(sys.argv) > 1:
        scenario(sys.argv[1])
        return
    res = {}
    for name in ("mvr_first", "mvr_alive", "torch_first"):
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), name], timeout=240)
            res[name] = {"rc": r.returncode, "s": round(time.time() - t0, 1)}
        except subprocess.TimeoutExpired:
            res[name] = {"rc": "TIMEOUT (hang reproduced)", "s": round(time.time() - t0, 1)}
            print(json.dumps(res), flush=True)
            sys.exit(3)          # a hung GPU process: start nothing else on this box
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
