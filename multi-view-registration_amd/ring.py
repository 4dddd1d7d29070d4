"""ring.py -- the global multi-view step over all scan pairs of the view graph,
sharded over one process per GPU.

One step = one outer pass of Registrator::registrationLUM
(mvr/src/registrator.cpp:625-664):
    posed clouds  <- getTransformedPoints(pose_v)              (K1, GPU)
    per ring edge (i -> i+1): reciprocal correspondences        (K2/K3, GPU)
                  + raw second moments of the accepted pairs    (K8, GPU)
    [world > 1]   one all-reduce (SUM) of the V x 32 f64 edge table (RCCL over xGMI)
    host          per-pair Umeyama + residual, LUM::compute from the moments
    pose_v        <- LUM_v * pose_v                              (:656-662)

Sharding: the V*Ns source queries of all edges are split into
`world` contiguous, equal ranges; every rank holds all scans, so no point ever
crosses the fabric -- only the 3 KB table does.  Each rank then solves the same
small system redundantly (no broadcast needed).

The compute backend is injected: `HipBackend` (the product; fails loudly without
a GPU) or any object with the same three methods -- the world_size-2 gloo tests
plug the CPU oracle in from tests/, the product never does.
"""
from __future__ import annotations

import os

import numpy as np

from . import Context, ring_host_step

ROW = 32   # doubles per edge row: {n, origin[3], sp[3], sq[3], spp[6], sqq[6], spq[9], 0}


def ring_edges(n_views):
    """registrator.cpp:640-643: source i, target (i+1) mod V."""
    return [(i, (i + 1) % n_views) for i in range(n_views)]


def split_queries(sizes, world, rank):
    """Even, contiguous split of the concatenated source queries of all edges.
    sizes[e] = number of source points of edge e.  -> [(edge, q_begin, q_count)]"""
    total = int(sum(sizes))
    lo, hi = total * rank // world, total * (rank + 1) // world
    segs, base = [], 0
    for e, n in enumerate(sizes):
        a, b = max(lo, base), min(hi, base + n)
        if b > a:
            segs.append((e, a - base, b - a))
        base += n
    return segs


class HipBackend:
    """Raw scans in slots V..2V-1, posed clouds in slots 0..V-1 of one mvr_ctx;
    the edge table is a torch CUDA tensor so that torch.distributed (RCCL) can
    all-reduce it in place."""

    def __init__(self, scans, device=0, stream=None, fma=False):
        import torch
        self.torch = torch
        self.V = len(scans)
        # ONE explicit HIP stream orders the library's kernels, torch's fills /
        # copies and the collective.  `stream` = raw hipStream_t of a torch
        # stream that is current in the caller; otherwise a private torch stream.
        self._tstream = None
        if not stream:
            self._tstream = torch.cuda.Stream(device=device)
            stream = self._tstream.cuda_stream
        self.ctx = Context(device, stream=stream)
        for v, s in enumerate(scans):
            self.ctx.upload(self.V + v, s)
        self.sizes = [len(s) for s in scans]
        with self._on_stream():
            self.table = torch.zeros((self.V, ROW), dtype=torch.float64, device=torch.device("cuda", device))
        self.fma = fma

    def _on_stream(self):
        import contextlib
        return self.torch.cuda.stream(self._tstream) if self._tstream is not None else contextlib.nullcontext()

    def pose_clouds(self, poses, views=None):
        vs = list(range(self.V) if views is None else views)
        self.ctx.transform_batch(vs, [self.V + v for v in vs], [poses[v] for v in vs])     # one launch for all views

    def edge_rows(self, segments, edges, max_dist, origin):
        """Fill this rank's rows (partial sums for split edges); returns the table."""
        with self._on_stream():
            self.table.zero_()
        if segments:
            # a rank's segments cover consecutive edges: one batched call, rows e0.. of the table;
            # the library runs every stage of all these pairs as one launch (culled mode) or the pairs on worker streams
            es = [e for e, _, _ in segments]
            assert es == list(range(es[0], es[0] + len(es)))
            self.ctx.pair_moments2_batch([edges[e] for e in es], max_dist, origin, dev_ptr=self.table[es[0]].data_ptr(),
                                         reciprocal=True, fma=self.fma, ranges=[(qb, qn) for _, qb, qn in segments])
        return self.table

    def to_host(self, table):
        with self._on_stream():
            return table.cpu().numpy()

    def fused_step(self, edges, poses, max_dist, origin, lum_iterations, steps=None):
        """the whole step in one native call (mvr_ring_step; steps=K: K of them, mvr_ring_run): single process only,
        every edge whole"""
        return self.ctx.ring_step(list(range(self.V)), [self.V + v for v in range(self.V)], edges, poses, max_dist, origin,
                                  lum_iterations=lum_iterations, reciprocal=True, fma=self.fma, steps=steps)

    def sharded_run(self, edges, poses, max_dist, origin, lum_iterations, steps):
        """this rank's share of `steps` passes with the all-reduce done natively (RCCL on the context's stream,
        mvr_ring_run_sharded): needs ctx.comm_init() on every rank first"""
        return self.ctx.ring_run_sharded(list(range(self.V)), [self.V + v for v in range(self.V)], edges, poses, max_dist, origin,
                                         steps=steps, lum_iterations=lum_iterations, reciprocal=True, fma=self.fma)

    def close(self):
        self.ctx.close()


class RingLUM:
    def __init__(self, backend, n_views, sizes, max_dist, origin, rank=0, world=1, all_reduce=None,
                 lum_iterations=16, fused=True, native_comm=False):
        self.b, self.V = backend, n_views
        self.edges = ring_edges(n_views)
        self.max_dist, self.origin = float(max_dist), np.asarray(origin, np.float64)
        self.rank, self.world, self.all_reduce = rank, world, all_reduce
        # native_comm: the backend's context carries an RCCL communicator (Context.comm_init) and the whole sharded loop,
        # all-reduce included, is native (mvr_ring_run_sharded); otherwise the all-reduce is a callable (torch.distributed)
        self.native_comm = bool(native_comm)
        if world > 1 and all_reduce is None and not self.native_comm:
            raise ValueError("world > 1 needs an all_reduce callable or native_comm=True")
        self.lum_iterations = lum_iterations
        self.fused = fused            # world == 1: use the backend's one-call step when it has one
        # edge e's queries are the points of its SOURCE view
        self.segments = split_queries([sizes[s] for s, _ in self.edges], world, rank)
        self.views_needed = sorted({v for e, _, _ in self.segments for v in self.edges[e]})
        self.last = {}

    def _native(self):
        b = self.b
        return (self.world == 1 and self.all_reduce is None and self.fused and hasattr(b, "fused_step")
                and os.environ.get("MVR_RING_FUSED", "1") != "0")

    def run(self, poses, steps):
        """`steps` outer passes (the loop of registrator.cpp:625-664).  Single process: one native call for all of them
        (mvr_ring_run); otherwise step by step.  self.last describes the last pass, its ms_* fields are per-pass means."""
        if steps <= 0:
            return poses
        if self.native_comm:
            new, info = self.b.sharded_run(self.edges, poses, self.max_dist, self.origin, self.lum_iterations, steps)
            self._set_last(info, steps)
            return new
        if not self._native():
            acc = [0.0, 0.0, 0.0]
            for _ in range(steps):
                poses = self.step(poses)
                for j, k in enumerate(("ms_enqueue", "ms_drain", "ms_host_solve")):
                    acc[j] += self.last[k]
            self.last.update(ms_enqueue=acc[0] / steps, ms_drain=acc[1] / steps, ms_host_solve=acc[2] / steps)
            return poses
        new, info = self.b.fused_step(self.edges, poses, self.max_dist, self.origin, self.lum_iterations, steps=steps)
        self._set_last(info, steps)
        return new

    def _set_last(self, info, steps=1):
        n = float(sum(info["pair_n"]))
        tm = info["timing_ms"]
        self.last = dict(info, n_corr=n,
                         mse=(sum(a * c for a, c in zip(info["pair_n"], info["pair_mse"])) / n) if n else 0.0,
                         ms_enqueue=tm[0] / steps, ms_drain=tm[1] / steps, ms_host_solve=tm[2] / steps)

    def step(self, poses):
        """poses: list of V (4,4) float64 column-vector poses -> new list."""
        import time
        b = self.b
        if self.native_comm:
            return self.run(poses, 1)
        if self._native():
            # single process: posing, searches, reductions, copy of the table and the host solve in ONE native call
            new, info = b.fused_step(self.edges, poses, self.max_dist, self.origin, self.lum_iterations)
            self._set_last(info)
            return new
        t0 = time.perf_counter()
        b.pose_clouds(poses, self.views_needed)     # only the scans this rank's segments touch
        table = b.edge_rows(self.segments, self.edges, self.max_dist, self.origin)
        t1 = time.perf_counter()              # everything enqueued (asynchronous)
        if self.all_reduce is not None:
            import contextlib
            # the collective must be ordered after the library's kernels: issue it on the backend's stream
            with getattr(b, "_on_stream", contextlib.nullcontext)():
                self.all_reduce(table)        # per-pair sums/residuals of all ranks -> every rank
        rows = b.to_host(table)
        t2 = time.perf_counter()              # GPU drained, table on the host
        # host: per-pair rigid solve (3x3 SVD on the moments) + residuals, LUM graph
        # solve, pose_v <- LUM_v * pose_v -- one native call, identical on every rank
        rc, new, info = ring_host_step(self.V, self.edges, rows, self.origin, poses, self.lum_iterations)
        if rc != 0:
            raise RuntimeError("LUM solve failed with status %d" % rc)
        n = float(sum(info["pair_n"]))
        t3 = time.perf_counter()
        self.last = dict(info, n_corr=n,
                         mse=(sum(a * b for a, b in zip(info["pair_n"], info["pair_mse"])) / n) if n else 0.0,
                         ms_enqueue=1e3 * (t1 - t0), ms_drain=1e3 * (t2 - t1), ms_host_solve=1e3 * (t3 - t2))
        return new
