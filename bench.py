#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: ICP correspondences/sec (+ ms/iteration)
on the 12-view x 200k-point turntable ring.

One "step" = one global iteration over ALL scan pairs of the view graph, i.e. one
outer pass of Registrator::registrationLUM (mvr/src/registrator.cpp:625-664):
  for each ring edge (i -> i+1): reciprocal nearest-neighbour correspondences
  (brute-force HIP kernel) + per-pair moment accumulation on the GPU;
  [N>1: one RCCL all-reduce of the 12x32 f64 edge table];
  host: per-pair Umeyama (3x3 SVD) + residuals, LUM graph solve, pose update;
  GPU: every scan moved by its new pose.
N ranks shard the 12 x 200k source queries evenly (strong scaling: the job is the
same 12-view ring at every N).  value = forward source queries of all ranks / s.

Contract: python bench.py --gpus N --steps K --warmup W ; one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 vector == FP32 matrix (dense) peak
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s
FLOP_PER_EVAL = 8.0           # 3 sub + 1 mul + 2 fma per point pair


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def visible_gpus():
    """GPUs a child process would see, counted in a child so that THIS process never initialises HIP."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE,
                       stderr=subprocess.DEVNULL, text=True, timeout=600)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, count_gpus=visible_gpus):
    """Parent of `python bench.py --gpus N` (no WORLD_SIZE in the environment): N ranks under torch.distributed.run,
    rank 0's single JSON line forwarded on stdout, non-zero exit if any rank fails.  Refuses when fewer than N GPUs
    are visible.  MVR_BENCH_LAUNCH_CMD (tests): the command a rank runs instead of this file."""
    import subprocess
    have = count_gpus()
    if have < n:
        log("bench.py: --gpus %d asked for but only %d GPU(s) visible: refusing (no oversubscription, no CPU ranks)" % (n, have))
        return 2
    target = os.environ.get("MVR_BENCH_LAUNCH_CMD", os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), target] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("[bench] launching %d ranks: %s" % (n, " ".join(cmd)))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        log("bench.py: the %d-rank run failed (exit %d, %d JSON line(s))" % (n, r.returncode, len(lines)))
        sys.stderr.write(r.stdout)
        return r.returncode or 1
    print(lines[0], flush=True)
    return 0


def main():
    # stdout carries exactly ONE line (the JSON): libraries that print banners to
    # fd 1 (RCCL prints its version there) are diverted to stderr until the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--max-dist", type=float, default=4.0)
    ap.add_argument("--fma", type=int, default=0, help="1: fma distance chain instead of the spec's rounded-per-op form")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nn-mode", type=int, default=1, help="1: exact culled search (default); 0: exact brute force")
    ap.add_argument("--no-bruteforce-pass", action="store_true",
                    help="skip the extra untimed brute-force ring pass that feeds roofline_bruteforce")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip cold_registration and secondary_sequential (they launch the same kernels on other shapes: a rocprofv3 --stats "
                         "average over the run then covers the ring step's launches only)")
    ap.add_argument("--no-projection", action="store_true", help="skip projected_scaling (one rank's share per world size, measured on this GPU)")
    ap.add_argument("--no-stress", action="store_true", help="skip secondary_stress_36x1M (a few passes of the 36-view x 1M-point ring)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed K-step window is repeated this many times after the headline one: min/median/max as extra keys")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process stays a plain parent (it never imports torch nor
        # touches HIP) and starts N fresh ranks, one per GPU
        os.dup2(real_stdout, 1)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node equal to --gpus"
                         % (args.gpus, os.environ["WORLD_SIZE"]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the RCCL path is taken even with one rank,
    # so that the collective code can be rehearsed on a 1-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("MVR_BENCH_FORCE_DIST", "0") == "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g
    if not os.path.exists(g.LIB):
        g.build_hip()
    mvr = importlib.import_module(g.PKG)

    V, N = args.views, args.points
    sp = mvr.synth_params(V, 3)
    t0 = time.time()
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    origin = np.array(sp.pivot)
    ring = importlib.import_module(g.PKG + ".ring")

    # one explicit (non-default) HIP stream carries everything in order: the
    # library's kernels, torch's zero_/D2H copies and the RCCL all-reduce
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream, "need a non-default stream handle"
    h2d0 = time.time()
    backend = ring.HipBackend(scans, device=local_rank, stream=stream, fma=bool(args.fma))
    ctx = backend.ctx
    ctx.sync()
    h2d = time.time() - h2d0
    # N > 1: the RCCL communicator belongs to the LIBRARY (mvr_ctx_comm_init = ncclCommInitRank); torch.distributed only
    # carries rank 0's 128-byte unique id to the other ranks and provides the barrier of the timing contract.  The whole
    # sharded loop -- posing, searches, ncclAllReduce of the edge table, host solve -- is one native call per rank.
    torch_allreduce = os.environ.get("MVR_BENCH_TORCH_ALLREDUCE", "0") == "1"      # the round-1 path, kept for comparison
    rccl_ranks = 1
    if use_dist and not torch_allreduce:
        ids = [mvr.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0, device=torch.device("cuda", local_rank))
        ctx.comm_init(ids[0], rank, world)
        rccl_ranks = ctx.comm_info()[2]
        if rccl_ranks != world:
            raise SystemExit("bench.py: RCCL reports %d ranks, expected %d" % (rccl_ranks, world))
    reg = ring.RingLUM(backend, V, [N] * V, args.max_dist, origin, rank=rank, world=world,
                       all_reduce=(dist.all_reduce if (use_dist and torch_allreduce) else None),
                       native_comm=(use_dist and not torch_allreduce))
    if rank == 0:
        log("[bench] synth %dx%d in %.2fs; rank0 segments %s" % (V, N, time.time() - t0, reg.segments))
    state = {"poses": [p.copy() for p in poses0]}

    def reset():
        state["poses"] = [p.copy() for p in poses0]
        ctx.sync()

    def step():
        # one registrationLUM outer pass over all scan pairs (ring.RingLUM.step)
        state["poses"] = reg.step(state["poses"])

    def run(n):
        # n outer passes, exactly n steps of the kind above.  Single process: the loop itself is native (mvr_ring_run,
        # as in Registrator::registrationLUMDevice of the C++ shim); several ranks: step by step, with the all-reduce
        state["poses"] = reg.run(state["poses"], n)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.tune(nn_mode=args.nn_mode)
    # The timed region carries NO instrumentation by default (per-launch HIP events cost ~5 % of a 0.9 ms step: the
    # roofline comes from the same steps repeated right after it, with every launch timed).  MVR_BENCH_PROF=2 puts
    # events on the search launches of the timed region as well (-> roofline_timed_region).
    prof_level = int(os.environ.get("MVR_BENCH_PROF", "0"))
    reset()
    ctx.prof_enable(prof_level)             # (events are pooled after first use: warm up in the mode of the timed region)
    # set-up, not measurement: the first passes ever allocate work buffers, build the scans' orderings and find the GPU at
    # its idle clocks; 20 untimed passes before the contract's W warm-up steps take that out of short (K = 10) windows
    run(int(os.environ.get("MVR_BENCH_PREWARM", "20")))
    reset()
    # (no collection of Python's heap inside a timed window, as timeit does: a window is a few milliseconds, a collection of this
    # process's heap -- 100 MB of scans -- one of them; it showed as a slow "rank" in the projection leg, twice at the same place.
    # Switched off ahead of the warm-up steps and back on after the windows.)
    import gc
    gc.disable()      # (and no gc.collect() here: a full collection right before the window cost the window 3 % -- the host code ran on cold caches)
    run(max(args.warmup, 1))
    reset()
    ctx.prof_reset(); ctx.prof_enable(prof_level)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    # the same K-step window again, `--repeats` times (each from the prior, like the headline one): spread of the number
    rep_s = []
    for _ in range(max(args.repeats, 0)):
        reset()
        barrier()
        tr = time.perf_counter()
        run(args.steps)
        barrier()
        rep_s.append(time.perf_counter() - tr)
    # ... and K more passes that CONTINUE the last window instead of restarting from the prior: what a pass costs once the
    # registration has settled (the windows above each pay for a pass whose every seed is 2 mm off; `value` stays the first one)
    conv_s = None
    if max(args.repeats, 0) > 0 and not use_dist:
        run(args.steps)
        barrier()
        tc = time.perf_counter()
        run(args.steps)
        barrier()
        conv_s = time.perf_counter() - tc
    gc.enable()
    if use_dist:
        tt = torch.tensor([elapsed] + rep_s, dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, rep_s = float(tt[0].item()), [float(v) for v in tt[1:].tolist()]
    rank_segments = [reg.segments]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, reg.segments)
        rank_segments = gathered

    nn_launches, nn_ms, nn_evals = ctx.prof_get(mvr.K_NN)
    rd_launches, rd_ms, rd_bytes = ctx.prof_get(mvr.K_REDUCE)
    gl_launches, gl_ms, gl_bytes = ctx.prof_get(mvr.K_GLUE)
    name, n_cu, mhz = ctx.device_info()
    culled = args.nn_mode != 0
    # evaluations a brute-force search of the same step performs (Ns*Nt + Nt'*Ns per pair)
    brute_equiv = None

    out = {
        "metric": "ICP correspondences/sec + ms/iteration, 12-view x 200k pts",
        "value": V * N * args.steps / elapsed,
        "unit": "correspondences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ranks": world, "rccl_ranks": rccl_ranks,
        "rank_query_ranges": [[list(map(int, sg)) for sg in segs] for segs in rank_segments],     # per rank: (edge, first query, count)
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d-view turntable ring, %d pts/scan: reciprocal p2p correspondences + per-pair "
                               "rigid solve over all %d scan pairs + LUM global step (registrationLUM outer pass)"
                               % (V, N, V),
                   "views": V, "points_per_scan": N, "max_distance": args.max_dist, "pairs": V,
                   "dist_mode": "fma" if args.fma else "rounded-per-op (spec)",
                   "nn_search": ("exact: bounded queries (seeded forward, every reverse search) by a uniform-grid walk, the rest by the spatially culled kernel (Hilbert tiles)"
                                 if culled else "exact, brute force"),
                   "sharding": "source queries of the %d ring pairs split evenly over %d rank(s)" % (V, world)},
        "accepted_correspondences_per_step": reg.last["n_corr"], "mse": reg.last["mse"],
        "device": name, "n_cu": n_cu,
        "step_breakdown_ms": {"enqueue": reg.last["ms_enqueue"], "gpu_drain": reg.last["ms_drain"],
                              "host_solve": reg.last["ms_host_solve"],
                              "piped_passes": ctx.stat("piped_passes"),
                              "note": "enqueue = host time spent pushing launches (overlaps the GPU: a pass's chain is queued while the previous "
                                      "pass runs, behind a gate); gpu_drain = host time spent waiting for a pass to finish; gpu_idle_ms (below, "
                                      "filled from the profiled re-run) = ms_per_step minus the sum of the step's kernel durations"},
    }
    if rep_s:
        per = sorted(1e3 * t / args.steps for t in rep_s)
        out["repeats"] = {"n": len(per), "ms_per_step_min": per[0], "ms_per_step_median": per[len(per) // 2], "ms_per_step_max": per[-1],
                          "value_median": V * N / (per[len(per) // 2] * 1e-3),
                          "note": "the same K-step window repeated after the headline one (each from the prior); `value` is the first window"}
    if conv_s is not None:
        out["settled_window"] = {"ms_per_step": 1e3 * conv_s / args.steps, "value": V * N * args.steps / conv_s,
                                 "note": "K passes that continue the registration (passes K+1 .. 2K after the last restart from the prior) instead "
                                         "of restarting it: no pass whose seeds are 2 mm off; reported beside `value`, never as it"}

    def traffic_of(fname, kernel_source):
        """HBM-side bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
        tools/measure_traffic.sh -> profiles/<fname>): counters cannot be collected inside this run, so the figure is
        only reported while the kernel source it was measured on is the one in the tree (sha256), else null."""
        import hashlib
        tp = os.path.join(ROOT, "profiles", fname)
        try:
            rec = json.load(open(tp))
            if kernel_source and rec.get("kernel_source_sha256"):
                now = hashlib.sha256(open(os.path.join(ROOT, g.PKG, "csrc", kernel_source), "rb").read()).hexdigest()
                if now != rec["kernel_source_sha256"]:
                    return None, "profiles/%s is stale (measured on another version of %s)" % (fname, kernel_source)
            return rec.get("hbm_bytes_per_launch"), "profiles/%s (PMC, %s)" % (fname, rec.get("source", "")[:120])
        except Exception:
            return None, "no profiles/%s" % fname

    def traffic_of_other(kernel_name):
        """the step's other kernels (posing, tail launch, filter + sums) from the same counter passes: profiles/nn_grid_traffic.json
        ["other_kernels"], reported while the source file each was measured on is unchanged"""
        import hashlib
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "nn_grid_traffic.json")))["other_kernels"][kernel_name]
            now = hashlib.sha256(open(os.path.join(ROOT, g.PKG, "csrc", rec["source_file"]), "rb").read()).hexdigest()
            if now != rec["source_sha256"]:
                return None, "profiles/nn_grid_traffic.json is stale for %s (measured on another version of %s)" % (kernel_name, rec["source_file"])
            return rec["hbm_bytes_per_launch"], "profiles/nn_grid_traffic.json other_kernels[%s] (PMC FETCH_SIZE x 2 + WRITE_SIZE, separate passes)" % kernel_name
        except Exception:
            return None, "no counter pass for %s in profiles/nn_grid_traffic.json" % kernel_name

    def hbm_roofline(kernel, launches, ms, evals, alg_bytes_per_launch, traffic, extra=None):
        """the roof that binds a search over a few candidates per query: SURVEY 8(d)'s compulsory bytes per launch against
        the HBM peak (intensity ~3 flop/B, far left of the 19.7 flop/B ridge); the executed-flop fraction rides along"""
        if not launches:
            return None
        avg_s = ms * 1e-3 / launches
        achieved = alg_bytes_per_launch / avg_s / 1e9
        r = {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
             "traffic": traffic[0], "traffic_source": traffic[1],
             "algorithmic_bytes_per_launch": alg_bytes_per_launch,
             "launches": launches, "avg_launch_ms": ms / launches, "evals_per_launch": evals / launches, "evals_per_s": evals / (ms * 1e-3),
             "frac_of_fp32_peak_on_executed_flops": FLOP_PER_EVAL * (evals / launches) / avg_s / 1e12 / PEAK_FP32_TFLOPS}
        if extra:
            r.update(extra)
        return r

    def nn_roofline(kernel, launches, ms, evals, traffic, extra=None):
        if not launches:
            return None
        avg_s = ms * 1e-3 / launches
        achieved = FLOP_PER_EVAL * (evals / launches) / avg_s / 1e12
        r = {"kernel": kernel, "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
             "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic[0], "traffic_source": traffic[1],
             "note": "priced against the FP32 VECTOR ALUs (157.3 TFLOP/s; gfx950's dense FP32 matrix peak is the same "
                     "figure, but no MFMA is issued: a 3-D distance has no dense contraction); achieved = "
                     "8 flop x point-pair evaluations EXECUTED per launch / mean launch time from HIP events",
             "launches": launches, "avg_launch_ms": ms / launches, "evals_per_launch": evals / launches,
             "evals_per_s": evals / (ms * 1e-3)}
        if extra:
            r.update(extra)
        return r

    # Roofline of the dominant kernel: the same steps once more right after the timed region, with every
    # kernel family timed (per-launch HIP events and evaluation counts) -- the timed region itself carries none.
    # (pair_streams=1 / pair_groups=1: the pairs, or the two groups of
    # pairs of the fused pass, would otherwise overlap on worker streams and a launch's duration would include
    # its neighbours.)
    iso = None
    if world == 1:
        ctx.tune(pair_streams=1, pair_groups=1)
        reset(); run(max(args.warmup, 1)); reset()          # the same warm-up as the timed region (seeded searches see the same history)
        ctx.prof_reset(); ctx.prof_enable(1)
        barrier(); ti = time.perf_counter()
        run(args.steps)
        barrier(); iso_elapsed = time.perf_counter() - ti
        ctx.prof_enable(False)
        iso = dict(nn=ctx.prof_get(mvr.K_NN), rd=ctx.prof_get(mvr.K_REDUCE), gl=ctx.prof_get(mvr.K_GLUE),
                   grid=ctx.prof_get(mvr.K_NN_GRID), wide=ctx.prof_get(mvr.K_NN_WIDE), xf=ctx.prof_get(mvr.K_XFORM),
                   ms_per_step=1e3 * iso_elapsed / args.steps,
                   n_corr=reg.last["n_corr"])       # the accepted correspondences of the profiled steps' last pass (NOT of the brute-force pass below)
        rd_launches, rd_ms, rd_bytes = iso["rd"]
        kern_ms = sum(iso[f][1] for f in ("nn", "rd", "gl", "grid", "wide", "xf")) / args.steps
        out["step_breakdown_ms"]["kernels_sum"] = kern_ms
        out["step_breakdown_ms"]["gpu_idle_ms"] = max(0.0, 1e3 * elapsed / args.steps - kern_ms)
        # (the same difference inside the profiled one-stream re-run: wall time and kernel sum of the SAME steps; the figure above
        # sets the headline window's wall time against the re-run's kernels)
        out["step_breakdown_ms"]["gpu_idle_ms_profiled_run"] = max(0.0, iso["ms_per_step"] - kern_ms)
    # the brute-force kernel (the plain VALU-roofline kernel) on the same pairs: one extra, untimed ring pass
    bf = None
    if world == 1 and not args.no_bruteforce_pass:
        ctx.tune(nn_mode=0, pair_streams=1)
        reset()
        step()
        ctx.prof_reset(); ctx.prof_enable(2)
        reset()
        step()
        ctx.prof_enable(False)
        bl, bms, bev = ctx.prof_get(mvr.K_NN)
        ctx.tune(nn_mode=args.nn_mode)
        if bl:
            brute_equiv = bev
            bf = nn_roofline("nn_kernel<8,32> (brute-force 1-NN, fwd + reciprocal; one untimed ring pass, one stream)", bl, bms,
                             bev, traffic_of("nn_traffic.json", None))
    kname = "nn_cull_kernel (exact culled 1-NN; one fused launch per direction for all scan pairs of a step)" if culled else "nn_kernel<8,32> (brute-force 1-NN, fwd + reciprocal)"
    ktraffic = traffic_of("nn_cull_traffic.json", "mvr_cull.hip") if culled else traffic_of("nn_traffic.json", None)
    grid_dominant = bool(iso and culled and iso["grid"][0] and iso["grid"][1] >= iso["nn"][1])
    if grid_dominant:
        # the fused pass answers its bounded queries (all of them from the second pass on) by the grid search: that
        # launch is the dominant kernel of a step; the culled kernel keeps the flagged query sets (and the first pass)
        gl_, gms, gev = iso["grid"]
        wl_, wms, wev = iso["wide"]
        cl_, cms, cev = iso["nn"]
        # SURVEY 8(d), compulsory bytes (12-byte points, every array once per kernel), per pair: forward search K2 =
        # 12 Ns + 12 Nt + 8 Ns; reverse search K3 = 12 Nt' + 12 Ns + 4 Nt' + 4 Ns with Nt' = the matched targets (taken as the
        # accepted correspondences: a lower bound).  One launch = all V pairs of a step in one direction; mean of the two.
        Ns = Nt = float(N)
        Ntp = iso["n_corr"] / float(V)
        fwd_bytes, rev_bytes = V * (12 * Ns + 12 * Nt + 8 * Ns), V * (12 * Ntp + 12 * Ns + 4 * Ntp + 4 * Ns)
        launches_per_step = gl_ / float(args.steps)          # 2 while one fused launch holds all pairs (V <= 12), more beyond
        alg_bytes = (fwd_bytes + rev_bytes) / launches_per_step
        extra = {"measured": "%d steps right after the timed region, HIP events and evaluation counters per launch; per step one forward "
                             "launch (all %d x %d source queries) and one reverse launch (the matched targets) for all scan pairs" % (args.steps, V, N),
                 "ms_per_step_one_stream_profiled": iso["ms_per_step"],
                 "algorithmic_bytes": {"forward_per_step": fwd_bytes, "reverse_per_step": rev_bytes, "launches_per_step": launches_per_step,
                                       "formula": "SURVEY 8(d) K2 / K3 with Nt' = accepted correspondences per pair; per launch = (forward + reverse) / launches per step"},
                 "limiter": "TA: the L1 gather path (rocprofv3 --pmc: TA_TA_BUSY ~88 % of the launch, ~13 distinct cache lines per vector load; "
                            "DESIGN.md 4.3) -- neither roof is near: HBM by the roofline model (3.4 flop/B against a 19.7 flop/B ridge), a few "
                            "per-lane 16-byte gathers per query in practice",
                 "evals_bruteforce_equivalent_per_step": brute_equiv}
        if brute_equiv:
            tot_ms = gms + wms + cms
            extra["culling_factor"] = brute_equiv / ((gev + wev + cev) / args.steps)
            extra["bruteforce_equivalent_tflops"] = FLOP_PER_EVAL * brute_equiv / (tot_ms * 1e-3 / args.steps) / 1e12
        out["roofline"] = hbm_roofline("nn_grid_kernel (exact 1-NN of the BOUNDED queries of a fused pass over a pose-invariant cell grid, one thread "
                                       "per query; forward and reverse launches of all scan pairs)", gl_, gms, gev, alg_bytes,
                                       traffic_of("nn_grid_traffic.json", "mvr_grid.hip"), extra)
        if wl_:
            out["roofline_stragglers"] = nn_roofline("nn_grid_tail_kernel (wide bounded queries, a wave each, and the listed 64-query sets, a block "
                                                     "each over the grid, in one launch; nn_grid_wide_kernel alone for the reverse pass)",
                                                     wl_, wms, wev, traffic_of_other("nn_grid_tail_kernel"))
        if cl_:
            out["roofline_culled"] = nn_roofline(kname, cl_, cms, cev, ktraffic, {"measured": "the culled launches among the same steps (none once every query has a bound)"})
    elif iso and iso["nn"][0]:
        il, ims, iev = iso["nn"]
        extra = {"measured": "%d steps right after the timed region, HIP events and evaluation counters per launch; one launch = "
                             "the searches of all %d scan pairs of a step" % (args.steps, V),
                 "ms_per_step_one_stream_profiled": iso["ms_per_step"]}
        if culled:
            extra["evals_bruteforce_equivalent_per_step"] = brute_equiv
            if brute_equiv:
                extra["culling_factor"] = brute_equiv / (iev / args.steps)
                extra["bruteforce_equivalent_tflops"] = FLOP_PER_EVAL * brute_equiv / (ims * 1e-3 / args.steps) / 1e12
        out["roofline"] = nn_roofline(kname, il, ims, iev, ktraffic, extra)
    if nn_launches and not grid_dominant:      # the same kernel inside the timed region, overlapped with other pairs' kernels
        r = nn_roofline(kname, nn_launches, nn_ms, nn_evals, ktraffic,
                        {"measured": "timed region (only the NN launches carry HIP events there; evaluations from running totals)"})
        if "roofline" in out:
            out["roofline_timed_region"] = r
        else:
            out["roofline"] = r
    if bf:
        out["roofline_bruteforce"] = bf
    if rd_launches:
        out["roofline_hbm"] = {
            "kernel": "pass1 + moments2 reductions (K5/K8)", "bound": "hbm",
            "achieved": rd_bytes / (rd_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": rd_bytes / (rd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "launches": rd_launches,
            "avg_launch_ms": rd_ms / rd_launches, "traffic": traffic_of_other("accept_moments2_batch_kernel")[0],
            "traffic_source": traffic_of_other("accept_moments2_batch_kernel")[1],
            "traffic_refresh_sorted_kernel": traffic_of_other("refresh_sorted_kernel")[0],
            "note": "`achieved` prices the launchers' MODELLED algorithmic bytes over the family's launches (filter + sums and its final launch); "
                    "`traffic` is the counter figure of the filter + sums launch (accept_moments2_batch_kernel) alone; launch-latency bound at "
                    "200k points per scan; one-stream pass",
        }
    out["pcie"] = {"h2d_s": h2d, "h2d_bytes": V * N * 16}

    # What a registration costs from a standing start, beside the steady state above: a FRESH context, the uploads, then K
    # passes from the prior in one native call -- no prewarm, no seeds, the orderings and grids built on the way (the reference
    # builds its kd-trees inside every align, registrator.cpp:569, and cpu_baseline pays for them too)
    if world == 1 and not args.no_secondary:
        cold_ctx = mvr.Context(local_rank)
        try:
            cold_ctx.tune(nn_mode=args.nn_mode)
            tc0 = time.perf_counter()
            for v in range(V):
                cold_ctx.upload(V + v, scans[v])
            cold_ctx.sync()
            tc1 = time.perf_counter()
            Kc = max(args.steps, 2)
            _, cinfo = cold_ctx.ring_step(list(range(V)), [V + v for v in range(V)], reg.edges, [p.copy() for p in poses0], args.max_dist, origin,
                                          fma=bool(args.fma), steps=Kc)
            tc2 = time.perf_counter()
            plog = cold_ctx.pass_log()
            out["cold_registration"] = {
                "passes": Kc, "upload_ms": 1e3 * (tc1 - tc0), "total_ms": 1e3 * (tc2 - tc1), "amortised_ms_per_pass": 1e3 * (tc2 - tc1) / Kc,
                "ms_per_pass": [round(v, 4) for v in plog], "piped_passes": cold_ctx.stat("piped_passes"),
                "accepted_correspondences_last_pass": float(sum(cinfo["pair_n"])),
                "note": "fresh context -> %d uploads -> %d passes from the mis-calibrated prior in one mvr_ring_run: pass 1 builds the scans' "
                        "Hilbert orderings and searches unseeded (culled kernel) while the cell grids are built on a side stream; from pass 2 "
                        "on the seeded grid search; pipelined once a pass has run without allocating" % (V, Kc)}
        finally:
            cold_ctx.close()

    # What ONE rank of a world of 2 / 4 / 8 would do per pass, measured on this one GPU (tools/rank_share_bench.py): a PROJECTION of
    # the multi-GPU critical path (the rank's chain, the pass's ncclAllReduce really issued on a communicator of one rank, the
    # absent peers' rows added behind it, the replicated solve) -- not a scaling run: the fabric and the ranks' skew are not in it
    if world == 1 and not args.no_secondary and not args.no_projection:
        import importlib.util as _ilu
        spec = _ilu.spec_from_file_location("rank_share_bench", os.path.join(ROOT, "tools", "rank_share_bench.py"))
        rsb = _ilu.module_from_spec(spec); spec.loader.exec_module(rsb)
        pctx = mvr.Context(local_rank)
        try:
            pctx.tune(nn_mode=args.nn_mode)
            for v in range(V):
                pctx.upload(V + v, scans[v])
            try:
                pctx.comm_init(mvr.comm_unique_id(), 0, 1)
                prccl = list(pctx.comm_info())
            except Exception as e:
                prccl = "no communicator (%s): the projection has no collective in it" % e
            posed_p, raw_p = list(range(V)), [V + v for v in range(V)]
            poses_c, _ = pctx.ring_run_sharded(posed_p, raw_p, reg.edges, [p.copy() for p in poses0], args.max_dist, origin, steps=40)
            proj = rsb.project(mvr, pctx, V, reg.edges, posed_p, raw_p, poses_c, origin, args.max_dist, [1, 2, 4, 8], max(args.steps, 10), 6)
            out["projected_scaling"] = {
                "is": "a PROJECTION measured on one GPU, not a multi-GPU run: per world size the slowest rank's ms per pass when this context plays "
                      "each rank in turn (mvr_ctx_project: its share of the queries by mvr_ring_segments, the all-reduce issued on a one-rank "
                      "communicator, the peers' rows added behind it, the replicated solve); passes continue from converged poses",
                "rccl": prccl, "workload": "%d-view ring x %d pts" % (V, N),
                "ms_per_step": {w: {"unpipelined": v["ms_per_step_pipeline0"], "pipelined": v["ms_per_step_pipeline1"]} for w, v in proj.items()},
                "rank0_timing_ms_unpipelined": {w: v["ranks"][0]["timing_ms_pipeline0"] for w, v in proj.items()},
                "pass_ms_median_max_of_slowest_rank": {w: {"unpipelined": max(v["ranks"], key=lambda r: r["ms_per_step_pipeline0"])["pass_ms_median_max_pipeline0"],
                                                           "pipelined": max(v["ranks"], key=lambda r: r["ms_per_step_pipeline1"])["pass_ms_median_max_pipeline1"]} for w, v in proj.items()},
                "note": "unpipelined is what a world of more than one rank runs today (a pass with a multi-rank collective is not queued "
                        "ahead of its poses: ADVICE r3); pipelined is what lifting that rule would give; timing = {enqueue, wait for the GPU, host solve}"}
        finally:
            pctx.close()

    # BASELINE configs[4]'s shape (36 views x 1M points: the working set is far beyond the caches) as a short leg of the default line,
    # so that the driver's record carries a number for it: a few timed passes on one GPU and the roofline fraction of the walk launches
    if world == 1 and not args.no_secondary and not args.no_stress and V == 12 and N == 200000:
        Vs, Ns_ = 36, 1000000
        sctx = mvr.Context(local_rank)
        try:
            sctx.tune(nn_mode=args.nn_mode)
            sps = mvr.synth_params(Vs, 5)
            pivs, axs = mvr.synth_prior(sps)
            for v in range(Vs):
                sctx.upload(Vs + v, mvr.synth_view(sps, v, Ns_))
            sposes0 = [np.eye(4)] + [mvr.axis_rotation(pivs, axs, mvr.turntable_angle(v, Vs)) for v in range(1, Vs)]
            sedges = [(v, (v + 1) % Vs) for v in range(Vs)]
            sposed, sraw = list(range(Vs)), [Vs + v for v in range(Vs)]
            sorigin = np.array(sps.pivot)
            sctx.ring_step(sposed, sraw, sedges, sposes0, args.max_dist, sorigin, fma=bool(args.fma), steps=3)      # orderings, grids, seeds
            sctx.sync()
            Ks = 4
            ts0 = time.perf_counter()
            _, sinfo = sctx.ring_step(sposed, sraw, sedges, sposes0, args.max_dist, sorigin, fma=bool(args.fma), steps=Ks)
            sctx.sync()
            sdt = time.perf_counter() - ts0
            sctx.tune(pair_streams=1, pair_groups=1)
            sctx.ring_step(sposed, sraw, sedges, sposes0, args.max_dist, sorigin, fma=bool(args.fma), steps=2); sctx.sync()
            sctx.prof_reset(); sctx.prof_enable(1)
            _, sinfo2 = sctx.ring_step(sposed, sraw, sedges, sposes0, args.max_dist, sorigin, fma=bool(args.fma), steps=Ks)
            sctx.sync(); sctx.prof_enable(False)
            gl2, gms2, gev2 = sctx.prof_get(mvr.K_NN_GRID)
            ncs = float(sum(sinfo2["pair_n"]))
            sfwd, srev = Vs * (12.0 * Ns_ + 12.0 * Ns_ + 8.0 * Ns_), Vs * (16.0 * (ncs / Vs) + 16.0 * Ns_)
            stress = {"workload": "%d-view ring x %d pts, 1 GPU (BASELINE configs[4]'s shape)" % (Vs, Ns_), "steps": Ks,
                      "ms_per_step": 1e3 * sdt / Ks, "value": Vs * Ns_ * Ks / sdt, "unit": "correspondences/s",
                      "accepted_correspondences": float(sum(sinfo["pair_n"])), "timing_ms": [t / Ks for t in sinfo["timing_ms"]]}
            if gl2:
                lps = gl2 / float(Ks)
                stress["roofline"] = {"kernel": "nn_grid_kernel (walk launches, forward + reverse)", "bound": "hbm", "launches_per_step": lps,
                                      "avg_launch_ms": gms2 / gl2, "algorithmic_bytes_per_launch": (sfwd + srev) / lps,
                                      "achieved": (sfwd + srev) / lps / (gms2 / gl2 * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": (sfwd + srev) / lps / (gms2 / gl2 * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                      "traffic": traffic_of("nn_grid_traffic_36x1M.json", "mvr_grid.hip")[0],
                                      "measured": "%d profiled steps (HIP events per launch, one stream) after the timed ones" % Ks}
            out["secondary_stress_36x1M"] = stress
        finally:
            sctx.close()

    # BASELINE configs[2] beside the headline: the SEQUENTIAL mode (Registrator::registrationICP, registrator.cpp:526-588:
    # views 1, V-1, 2, ... each aligned to the growing merged target), device-resident, same scans -- ms per align
    def sequential_sweeps(repeat):
        order = []
        for i in range(1, V // 2):
            order += [i, V - i]
        order.append(V // 2)
        RAW, TARGET, SOURCE, OUT = V, 2 * V, 2 * V + 1, 2 * V + 2          # the raw scans already sit in slots V .. 2V-1
        params = mvr.icp_params(max_dist=args.max_dist, max_iter=1000, fma=bool(args.fma))
        poses, ncorr, per_sweep, first = [p.copy() for p in poses0], [], [], None
        for r in range(repeat):                                            # registrator.cpp:530: every sweep rebuilds the model from view 0
            ctx.sync()
            t0 = time.perf_counter()
            # one sweep = ONE native call (mvr_seq_run: the loop of registrator.cpp:562-577 around mvr_icp_align)
            poses, log = ctx.seq_run([RAW + v for v in range(V)], TARGET, SOURCE, OUT, params, poses, repeat=1)
            ctx.sync()
            per_sweep.append(time.perf_counter() - t0)
            poses = [np.array(p) for p in poses]
            if r == 0:
                ncorr = [e["n_corr"] for e in log]
                first = [p.copy() for p in poses]
        return first, ncorr, per_sweep, order
    seq = None
    if world == 1 and V >= 4 and not args.no_secondary:
        ctx.tune(nn_mode=args.nn_mode, pair_streams=6, pair_groups=2)
        sequential_sweeps(1)                                            # warm-up: allocations, orderings
        # the warm-up aligned the same scans from the same poses: the seeds it left (seq_seed) are forgotten, so that the timed
        # run's first sweep is a first sweep and only its later ones start from the matches of the sweep before
        ctx.tune(seq_seed=0); ctx.tune(seq_seed=1)
        sweeps = 5                                                      # repeat_times of the reference (registrator.cpp:530, parameter default)
        seq_poses, seq_ncorr, seq_dts, seq_order = sequential_sweeps(sweeps)
        seq_dt = sum(seq_dts)
        seq = {"config": "%d-view ring, %d pts/scan, sequential pairwise ICP against the growing target (%d sweeps of %d aligns as "
                         "registrationICP runs them, target grows to %d points in each)" % (V, N, sweeps, len(seq_order), V * N),
               "ms_per_align": 1e3 * seq_dt / (sweeps * len(seq_order)), "queries_per_s": N * sweeps * len(seq_order) / seq_dt,
               "ms_per_align_by_sweep": [1e3 * t / len(seq_order) for t in seq_dts],
               "ms_per_align_first_sweep": 1e3 * seq_dts[0] / len(seq_order),      # unseeded: the figure that is like for like with cpu_oracle_ms_per_align (one unseeded sweep)
               "ms_per_align_is": "the mean over %d sweeps, %d of them seeded by the sweep before" % (sweeps, sweeps - 1),
               "note": "one native call per sweep (mvr_seq_run); sweep 1 searches unseeded, from sweep 2 on every forward search starts from the "
                       "scan's match of the sweep before (seq_seed); n_corr and the oracle comparison are sweep 1's",
               "n_corr": seq_ncorr}
        out["secondary_sequential"] = seq

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle as orc    # checker only: the CPU restatement timed beside the GPU path
        t0 = time.perf_counter()
        clouds = [orc.transform_f64(poses0[v], scans[v]) for v in range(V)]
        nq, passes = 0, 0
        while time.perf_counter() - t0 < 10.0:            # bounded sample: >= 10 s, whole ring passes
            for s, t in reg.edges:
                c = orc.correspondences(clouds[s], clouds[t], args.max_dist, reciprocal=True, fma=bool(args.fma), kdtree=True)
                orc.umeyama(clouds[s], clouds[t], c)
                nq += len(clouds[s])
                if time.perf_counter() - t0 > 45:
                    break
            passes += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nq / dt, "unit": "correspondences/s", "cores": 1, "kind": "port",
                               "sample": "%d ring pairs (%d pass(es) over the %d pairs of the same workload; kd-tree "
                                         "exact NN built per pair, reciprocal filter, Umeyama), oracle/mvr_oracle.c, "
                                         "1 thread, %.1f s" % (nq // N, passes, V, dt),
                               "host_cpus": os.cpu_count()}
        # courtesy upper bound: the same searches with OpenMP over the queries; the reference is single-threaded
        threads = max(1, min(16, os.cpu_count() or 1))         # a 1-GPU box's CPU share
        t0 = time.perf_counter()
        nq = 0
        while time.perf_counter() - t0 < 5.0:
            for s, t in reg.edges:
                c = orc.correspondences_mt(clouds[s], clouds[t], args.max_dist, threads, reciprocal=True, fma=bool(args.fma))
                orc.umeyama(clouds[s], clouds[t], c)
                nq += len(clouds[s])
        dt_omp, nq_omp = time.perf_counter() - t0, nq
        if seq is not None and V * N <= 3_000_000:
            # the oracle's restatement of the same sequential sweep: its ms per align, and how far the GPU's final poses are from it
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import ref_driver
            ts0 = time.perf_counter()
            oposes, olog = ref_driver.sequential_icp(orc, scans, poses0, orc.make_params(max_dist=args.max_dist, max_iter=1000, fma=bool(args.fma)),
                                                     V, fitness_last=False)
            dt_seq = time.perf_counter() - ts0
            seq["cpu_oracle_ms_per_align"] = 1e3 * dt_seq / len(olog)
            seq["pose_delta_vs_oracle"] = {"rot": max(float(np.abs(seq_poses[v][:3, :3] - oposes[v][:3, :3]).max()) for v in range(V)),
                                           "trans_mm": max(float(np.abs(seq_poses[v][:3, 3] - oposes[v][:3, 3]).max()) for v in range(V)),
                                           "n_corr_equal": seq_ncorr == [e["n_corr"] for e in olog]}
        out["cpu_baseline_openmp"] = {"value": nq_omp / dt_omp, "unit": "correspondences/s", "cores": threads, "kind": "port",
                                      "sample": "%d ring pairs, kd-trees built serially, per-query searches on %d OpenMP threads, %.1f s"
                                                % (nq_omp // N, threads, dt_omp)}
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
