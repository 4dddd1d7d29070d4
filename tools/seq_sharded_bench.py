#!/usr/bin/env python3
"""tools/seq_sharded_bench.py -- BASELINE configs[2] shape (12 views x 200k, sequential registration against the growing
target) with the target SHARDED by points (multi-view-registration_amd/seq.py).

    python tools/seq_sharded_bench.py --parts 2                        one process, the shards walked serially on one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           tools/seq_sharded_bench.py                                  one shard per rank: MIN / SUM all-reduce over RCCL

Prints one JSON line (rank 0): ms per align, source queries per second, the collectives' payload."""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=12); ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--parts", type=int, default=1, help="shards walked serially when not launched under torch.distributed")
    ap.add_argument("--max-dist", type=float, default=4.0); ap.add_argument("--repeat", type=int, default=1)
    args = ap.parse_args()
    import torch
    mvr = importlib.import_module("multi-view-registration_amd")
    seq = importlib.import_module("multi-view-registration_amd.seq")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("MVR_FORCE_DIST") == "1")     # rehearse RCCL with one rank
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)       # "nccl" is RCCL on ROCm
    V, N = args.views, args.points
    sp = mvr.synth_params(V, 3)
    scans = [mvr.synth_view(sp, v, N) for v in range(V)]
    piv, ax = mvr.synth_prior(sp)
    poses0 = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    ts = torch.cuda.Stream(device=local)
    n_parts = world if use_dist else args.parts
    parts = [seq.HipPart(scans, device=local, tstream=ts) for _ in range(1 if use_dist else n_parts)]
    rmin = rsum = None
    if use_dist:
        def rmin(k):
            with torch.cuda.stream(ts):
                dist.all_reduce(k, op=dist.ReduceOp.MIN)
        def rsum(r):
            with torch.cuda.stream(ts):
                dist.all_reduce(r, op=dist.ReduceOp.SUM)
    drv = seq.ShardedSequentialICP(parts, V, N, n_parts, part0=rank if use_dist else 0, all_reduce_min=rmin, all_reduce_sum=rsum,
                                   origin=np.array(sp.pivot))
    params = mvr.icp_params(max_dist=args.max_dist)
    drv.run(poses0, params)                        # warm-up (index builds, allocations)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t0 = time.perf_counter()
    poses, log = drv.run(poses0, params, repeat=args.repeat)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        aligns = len(log)
        print(json.dumps(dict(config="%d views x %d pts, sequential vs growing target, target sharded %d ways (%s)" %
                              (V, N, n_parts, "one shard per rank, RCCL" if use_dist else "one process, shards walked serially"),
                              aligns=aligns, ms_per_align=1e3 * dt / aligns, queries_per_s=aligns * N / dt,
                              n_corr=[e["n_corr"] for e in log], iterations=[e["iterations"] for e in log],
                              allreduce_bytes_per_iteration={"keys_min": 8 * N, "moments_sum": 256})))
    for p in parts:
        p.close()
    if dist: dist.destroy_process_group()


if __name__ == "__main__":
    main()
