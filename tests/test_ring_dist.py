"""The N>1 path on CPU: world_size-2 (and 3) gloo runs of the sharded ring step
(multi-view-registration_amd/ring.py) with the CPU oracle plugged in as the
compute backend.  Checks: the query split covers every source point exactly
once, the all-reduced edge table equals the unsharded one, every rank ends with
the same poses, and those poses equal the oracle's registrationLUM pass."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

import ref_driver
from conftest import PKG, ROOT, load_golden

from test_host import _moments_numpy


def test_split_queries_covers_everything(mvr):
    ring = importlib.import_module(PKG + ".ring")
    for sizes in ([200000] * 12, [5, 0, 7, 3], [1000] * 36, [10]):
        for world in (1, 2, 3, 4, 8, 16):
            seen = [np.zeros(n, int) for n in sizes]
            counts = []
            for r in range(world):
                segs = ring.split_queries(sizes, world, r)
                counts.append(sum(n for _, _, n in segs))
                for e, b, n in segs:
                    assert n > 0 and b + n <= sizes[e]
                    seen[e][b:b + n] += 1
                assert len({e for e, _, _ in segs}) == len(segs)      # one segment per (rank, edge)
            assert all((s == 1).all() for s in seen)
            assert max(counts) - min(counts) <= 1                      # balanced
    assert ring.ring_edges(12)[-1] == (11, 0)


class OracleBackend:
    """Test-only backend: same interface as ring.HipBackend, CPU oracle inside."""

    def __init__(self, orc, scans):
        import torch
        self.orc, self.torch, self.scans, self.V = orc, torch, scans, len(scans)
        self.clouds = None

    def pose_clouds(self, poses, views=None):
        self.clouds = [self.orc.transform_f64(poses[v], self.scans[v]) for v in range(self.V)]

    def edge_rows(self, segments, edges, max_dist, origin):
        table = self.torch.zeros((self.V, 32), dtype=self.torch.float64)
        for e, qb, qn in segments:
            s, t = edges[e]
            c = self.orc.correspondences(self.clouds[s], self.clouds[t], max_dist, reciprocal=True)
            c = c[(c["query"] >= qb) & (c["query"] < qb + qn)]
            table[e] = self.torch.from_numpy(_moments_numpy(self.clouds[s], self.clouds[t], c["query"], c["match"], origin))
        return table

    def to_host(self, table):
        return table.numpy()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mvr = importlib.import_module(PKG)
    ring = importlib.import_module(PKG + ".ring")
    g = load_golden("ring_12x2048.npz")
    scans, poses = list(g["scans"]), list(g["poses0"])
    sp = mvr.synth_params(12, 3)
    be = OracleBackend(orc, scans)
    r = ring.RingLUM(be, 12, [len(s) for s in scans], 8.0, np.array(sp.pivot), rank=rank, world=world,
                     all_reduce=dist.all_reduce)
    tables = []
    orig = be.to_host
    be.to_host = lambda t: (tables.append(t.numpy().copy()), orig(t))[1]
    for _ in range(2):
        poses = r.step(poses)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), poses=np.stack(poses), table=tables[0],
             ncorr=np.array(r.last["pair_n"]), segs=np.array(r.segments))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ring_lum_sharded_gloo(mvr, orc, tmp_path, world):
    import torch.multiprocessing as mp
    ring = importlib.import_module(PKG + ".ring")
    g = load_golden("ring_12x2048.npz")
    scans, poses0 = list(g["scans"]), list(g["poses0"])
    sp = mvr.synth_params(12, 3)
    # unsharded reference run in this process
    be = OracleBackend(orc, scans)
    r1 = ring.RingLUM(be, 12, [len(s) for s in scans], 8.0, np.array(sp.pivot))
    tables = []
    orig = be.to_host
    be.to_host = lambda t: (tables.append(t.numpy().copy()), orig(t))[1]
    p = [q.copy() for q in poses0]
    first = r1.step(p)
    second = r1.step(first)
    # the first pass is exactly the golden registrationLUM pass of the oracle driver
    assert [int(n) for n in np.array(tables[0])[:, 0]] == list(g["lum_ncorr"])
    for v in range(12):
        assert np.abs(first[v] - g["lum_poses"][v]).max() < 1e-6
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(tmp_path, "rank%d.npz" % k)) for k in range(world)]
    for o in outs:
        assert np.allclose(o["table"][:, 0], tables[0][:, 0])                       # counts add up exactly
        assert np.allclose(o["table"][:, 4:], tables[0][:, 4:], rtol=1e-12, atol=1e-6)
        assert np.abs(o["poses"] - np.stack(second)).max() < 1e-9
        assert np.array_equal(o["poses"], outs[0]["poses"])                         # every rank agrees bit for bit
    assert sum(int(o["segs"][:, 2].sum()) for o in outs) == 12 * 2048
