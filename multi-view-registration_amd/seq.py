"""seq.py -- the reference's SEQUENTIAL registration (Registrator::registrationICP,
mvr/src/registrator.cpp:526-588) with the growing target sharded by points over ranks
("Sequential ICP vs growing target").

The loop is loop-carried (target_k depends on pose_{k-1}), so it does not shard by scan
pair.  Instead every rank holds the full source scan and one SLICE of every scan that has
been merged into the target; one ICP iteration is

    forward NN of all source points in the rank's target shard        (device)
    all-reduce MIN of the Ns packed keys (d2 bits << 32 | GLOBAL idx)  (RCCL int64/min)
    reciprocal check + raw second moments of the matches it OWNS       (device)
    all-reduce SUM of 32 doubles                                       (RCCL)
    3x3 SVD (Umeyama), convergence criteria, transform of the source   (host + device)

and after the align every rank appends ITS slice of the transformed source to its shard.
Ties go to the lowest GLOBAL target index, so the correspondences are exactly those of the
single-GPU / single-process run; the sums differ only in the order of f64 additions.

THE PRODUCT'S HOST FOR THIS MODE IS NATIVE: mvr_seq_run_sharded / mvr_seq_align_sharded (csrc/mvr_ctx.hip, the
collectives on the library's own RCCL communicator in csrc/mvr_world.cpp), wrapped below as
NativeShardedSequentialICP -- one rank = one context + its communicator, torch (or any launcher) only carries
rank 0's 128-byte unique id.

ShardedSequentialICP is the same loop, statement by statement, over PLUGGABLE parts and reductions: it exists so
that the partitioning, the tie rule and the failure protocol can be exercised where no second GPU is --
  * several parts in one process (a 1-GPU "fake world" that walks the shards serially, tests/test_gpu_seq.py),
  * one part per process over gloo with the CPU oracle as the part (tests/test_seq_dist.py).
A part is a backend object: HipPart (this package, the very C entry points the native loop calls) on the GPU; the
tests plug the CPU oracle in as a second backend.

Failure protocol (the native loop's, rehearsed here): a rank whose LOCAL work fails does not leave the loop -- it
joins the iteration's two reductions with neutral keys, a zero row and a raised failure count, so that every rank
ends that iteration with RankFailure instead of one rank leaving its peers inside a collective; a reduction that
itself fails (a peer died: gloo / RCCL error or timeout) surfaces as RankFailure too."""
import numpy as np

from . import (Context, IcpParams, PairMoments2, mat4d_mul, mat4f_mul, moments_from_moments2,
               umeyama_from_moments)

INT64_MAX = np.iinfo(np.int64).max
ROW = 32
DBL_MAX = np.finfo(np.float64).max


def view_order(n_views=12):
    """registrator.cpp:530-541: 1, 11, 2, 10, ..., then the centre view."""
    half = n_views // 2
    order = []
    for i in range(1, half):
        order += [i, n_views - i]
    order.append(half)
    return order


def slice_bounds(n, parts):
    """equal contiguous slices of a scan of n points: part g owns [b[g], b[g+1])"""
    return [n * g // parts for g in range(parts + 1)]


class RankFailure(RuntimeError):
    """this rank's local work, a peer's, or a collective failed: the registration cannot go on (never a hang)"""


class HipPart:
    """One target shard on one GPU context.  Slots: 0 target shard, 1 posed source, 2 current source
    (input_transformed), 3 aligned output, 8.. raw scans."""
    TARGET, SOURCE, CUR, OUT, RAW0 = 0, 1, 2, 3, 8

    def __init__(self, scans, device=0, tstream=None):
        """tstream: a torch.cuda.Stream shared by every part of this process (library kernels, torch's
        reductions between parts and the collectives are ordered by that ONE stream); default: a new one."""
        import torch
        self.torch = torch
        self._tstream = tstream if tstream is not None else torch.cuda.Stream(device=device)
        self.ctx = Context(device, stream=self._tstream.cuda_stream)
        self.device = torch.device("cuda", device)
        for v, s in enumerate(scans):
            self.ctx.upload(self.RAW0 + v, s)
        self.n = [len(s) for s in scans]
        with self._on_stream():
            self.keys = torch.empty(max(self.n), dtype=torch.int64, device=self.device)
            self.row = torch.zeros(ROW, dtype=torch.float64, device=self.device)

    def _on_stream(self):
        return self.torch.cuda.stream(self._tstream)

    # -- target shard
    def start_target(self, view, pose, lo, hi, global_begin):
        c = self.ctx
        c.transform(self.OUT, self.RAW0 + view, pose)
        c.clear(self.TARGET)
        c.append_range(self.TARGET, self.OUT, lo, hi - lo, global_begin)

    def append_out(self, lo, hi, global_begin):
        self.ctx.append_range(self.TARGET, self.OUT, lo, hi - lo, global_begin)

    # -- source
    def pose_source(self, view, pose):
        self.ctx.transform(self.SOURCE, self.RAW0 + view, pose)
        self.ctx.copy(self.CUR, self.SOURCE)
        self.ns = self.n[view]

    def forward_keys(self, max_dist, fma):
        """-> torch int64 [Ns] on the device (a view into the part's key buffer)"""
        k = self.keys[: self.ns]
        self.ctx.nn_forward_keys(self.CUR, self.TARGET, max_dist, k.data_ptr(), fma=fma)
        return k

    def moments_from_keys(self, keys, max_dist, origin, reciprocal, fma):
        self.ctx.pair_moments2_from_keys(self.CUR, self.TARGET, keys.data_ptr(), max_dist, origin, self.row.data_ptr(),
                                         reciprocal=reciprocal, fma=fma)
        return self.row

    def transform_current(self, T):
        self.ctx.transform_f32(self.CUR, self.CUR, T)

    def finish(self, final):
        """out = final * (*input), from the ORIGINAL input (App. A.1)"""
        self.ctx.transform_f32(self.OUT, self.SOURCE, final)

    def download_out(self):
        return self.ctx.download(self.OUT)

    # -- reductions between parts of ONE process (same device)
    def min_into(self, a, b):
        with self._on_stream():
            self.torch.minimum(a, b, out=a)

    def add_into(self, a, b):
        with self._on_stream():
            a.add_(b)

    def keys_to_host(self, keys):
        with self._on_stream():
            return keys.cpu().numpy()

    # -- failure protocol: neutral contributions, the failure count rides in the row's origin slot (a constant, re-set
    #    after the reduction)
    def neutral_keys(self):
        with self._on_stream():
            k = self.keys[: self.ns]
            k.fill_(INT64_MAX)
        return k

    def neutral_row(self):
        with self._on_stream():
            self.row.zero_()
        return self.row

    def set_status(self, row, failed):
        with self._on_stream():
            row[1:4] = 0.0
            row[1] = 1.0 if failed else 0.0

    def row_to_host(self, row):
        with self._on_stream():
            return row.cpu().numpy()

    def close(self):
        self.ctx.close()


def converged(state, T, mse, params, iterations):
    """pcl::registration::DefaultConvergenceCriteria::hasConverged.  state: dict(prev_mse)."""
    if iterations >= params.max_iterations:
        return True, "ITERATIONS"
    cos_angle = 0.5 * (float(T[0, 0]) + float(T[1, 1]) + float(T[2, 2]) - 1.0)
    trans2 = float(T[0, 3]) ** 2 + float(T[1, 3]) ** 2 + float(T[2, 3]) ** 2
    if cos_angle >= 1.0 - params.transformation_epsilon and trans2 <= params.transformation_epsilon:
        return True, "TRANSFORM"
    prev = state["prev_mse"]
    if abs(mse - prev) < 1e-12:
        return True, "ABS_MSE"
    if abs(mse - prev) / prev < params.euclidean_fitness_eps:
        return True, "REL_MSE"
    state["prev_mse"] = mse
    return False, "NOT"


class ShardedSequentialICP:
    """parts: the shards this process holds (global part ids part0 .. part0 + len(parts) - 1 of n_parts);
    all_reduce_min / all_reduce_sum: in-place reductions over PROCESSES of a keys / row object of the
    backend (None when this process holds every part)."""

    def __init__(self, parts, n_views, n_points, n_parts, part0=0, all_reduce_min=None, all_reduce_sum=None,
                 origin=(0.0, 0.0, 0.0)):
        self.parts, self.V, self.N = parts, n_views, n_points
        self.G, self.g0 = n_parts, part0
        if (len(parts) != n_parts) and (all_reduce_min is None or all_reduce_sum is None):
            raise ValueError("parts missing from this process need all_reduce_min / all_reduce_sum")
        self.rmin, self.rsum = all_reduce_min, all_reduce_sum
        self.bounds = slice_bounds(n_points, n_parts)
        self.origin = np.asarray(origin, np.float64)

    def _bounds(self, k):
        g = self.g0 + k
        return self.bounds[g], self.bounds[g + 1]

    @staticmethod
    def _reduce(fn, buf):
        """a reduction over processes; its own failure (a peer died, a timeout) is a RankFailure, not a hang"""
        if fn is None:
            return
        try:
            fn(buf)
        except Exception as e:
            raise RankFailure("a collective failed: a peer is gone (%s)" % (str(e).splitlines()[0] if str(e) else type(e).__name__)) from e

    def align(self, params: IcpParams):
        """one IterativeClosestPoint::align (App. A.1) of the posed source against the sharded target.
        Returns (final 4x4 float32, stats)."""
        p0 = self.parts[0]
        final = np.eye(4, dtype=np.float32)
        state = {"prev_mse": DBL_MAX}
        iters, conv, why, n, mse = 0, False, "NOT", 0, 0.0
        fma, rec = bool(params.fma_dist), bool(params.use_reciprocal)
        while True:
            local_error = None
            keys = None
            try:
                for part in self.parts:                  # forward search of every local shard
                    k = part.forward_keys(params.max_corr_dist, fma)
                    if keys is None:
                        keys = k
                    else:
                        p0.min_into(keys, k)
            except Exception as e:                       # local work failed: stay in the loop, contribute nothing
                local_error, keys = e, p0.neutral_keys()
            self._reduce(self.rmin, keys)                # MIN over ranks: ties -> lowest global index
            row = None
            if local_error is None:
                try:
                    for part in self.parts:              # every match is reduced by the part that owns its target
                        r = part.moments_from_keys(keys, params.max_corr_dist, self.origin, rec, fma)
                        if row is None:
                            row = r
                        else:
                            p0.add_into(row, r)
                except Exception as e:
                    local_error = e
            if local_error is not None:
                row = p0.neutral_row()
            p0.set_status(row, local_error is not None)
            self._reduce(self.rsum, row)
            h = p0.row_to_host(row)
            if h[1] > 0:                                 # some rank's local work failed in this iteration: all ranks stop here
                raise RankFailure("local work failed on this rank: %r" % (local_error,) if local_error is not None
                                  else "a peer's local work failed in this iteration") from local_error
            n = int(round(h[0]))
            if n < 3:                                    # "Not enough correspondences found" (App. A.1)
                conv, why = False, "NO_CORRESPONDENCES"
                break
            m2 = PairMoments2()
            np.ctypeslib.as_array(m2.origin)[:] = self.origin          # a constant, not a sum over ranks
            m2.n = h[0]
            np.ctypeslib.as_array(m2.sp)[:] = h[4:7]; np.ctypeslib.as_array(m2.sq)[:] = h[7:10]
            np.ctypeslib.as_array(m2.spp)[:] = h[10:16]; np.ctypeslib.as_array(m2.sqq)[:] = h[16:22]
            np.ctypeslib.as_array(m2.spq)[:] = h[22:31]
            mse = h[31] / h[0]                           # mean of the correspondences' (f32) d2, as PCL does
            T, _ = umeyama_from_moments(moments_from_moments2(m2))
            for part in self.parts:
                part.transform_current(T)
            final = mat4f_mul(T, final)
            iters += 1
            conv, why = converged(state, T, mse, params, iters)
            if conv:
                break
        for part in self.parts:
            part.finish(final)
        return final, dict(iterations=iters, converged=conv, state=why, n_corr=n, mse=mse)

    def run(self, poses, params: IcpParams, repeat=1):
        """registrationICP: returns (poses, log).  poses: list of V (4,4) float64 column-vector poses."""
        poses = [np.array(p, np.float64) for p in poses]
        log = []
        for _ in range(repeat):
            for k, part in enumerate(self.parts):
                lo, hi = self._bounds(k)
                part.start_target(0, poses[0], lo, hi, lo)                     # scan 0 = global points [0, N)
            for a, v in enumerate(view_order(self.V)):
                for part in self.parts:
                    part.pose_source(v, poses[v])
                T, st = self.align(params)
                log.append(dict(st, view=v, T=T.copy()))
                poses[v] = mat4d_mul(T.astype(np.float64), poses[v])
                for k, part in enumerate(self.parts):
                    lo, hi = self._bounds(k)
                    part.append_out(lo, hi, (a + 1) * self.N + lo)             # merged scan a+1 = global [(a+1) N, (a+2) N)
        return poses, log


class NativeShardedSequentialICP:
    """The product path: one rank = one library context with (optionally) its own RCCL communicator; the whole
    registrationICP loop -- searches, ncclAllReduce(min) of the keys, sums, ncclAllReduce(sum), host solve, append of the
    rank's slice -- is ONE native call per rank (mvr_seq_run_sharded).  Slots as HipPart."""
    TARGET, SOURCE, OUT, RAW0 = 0, 1, 3, 8

    def __init__(self, scans, device=0, stream=None, origin=(0.0, 0.0, 0.0)):
        self.ctx = Context(device, stream=stream)
        for v, s in enumerate(scans):
            self.ctx.upload(self.RAW0 + v, s)
        self.V = len(scans)
        self.origin = np.asarray(origin, np.float64)

    def comm_init(self, unique_id, rank, world):
        """collective: every rank, with rank 0's mvr.comm_unique_id()"""
        self.ctx.comm_init(unique_id, rank, world)

    def run(self, poses, params: IcpParams, repeat=1):
        new, log = self.ctx.seq_run_sharded([self.RAW0 + v for v in range(self.V)], self.TARGET, self.SOURCE, self.OUT, params,
                                            self.origin, poses, repeat=repeat)
        return [new[v] for v in range(self.V)], log

    def shard_size(self):
        return self.ctx.size(self.TARGET)

    def close(self):
        self.ctx.close()
