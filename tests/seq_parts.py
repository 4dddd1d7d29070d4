"""Test-only backend of multi-view-registration_amd/seq.py: one target shard held in numpy, every
operation done by the CPU oracle.  Same interface as seq.HipPart."""
import numpy as np

INT64_MAX = np.iinfo(np.int64).max


class OraclePart:
    def __init__(self, orc, scans, kdtree=True):
        self.orc, self.scans, self.kd = orc, scans, kdtree
        self.tgt = np.zeros((0, 4), np.float32)
        self.gidx = np.zeros(0, np.int64)

    def start_target(self, view, pose, lo, hi, global_begin):
        self.out = self.orc.transform_f64(pose, self.scans[view])
        self.tgt = self.out[lo:hi].copy()
        self.gidx = global_begin + np.arange(hi - lo, dtype=np.int64)

    def append_out(self, lo, hi, global_begin):
        self.tgt = np.concatenate([self.tgt, self.out[lo:hi]])
        self.gidx = np.concatenate([self.gidx, global_begin + np.arange(hi - lo, dtype=np.int64)])

    def pose_source(self, view, pose):
        self.src = self.orc.transform_f64(pose, self.scans[view])
        self.cur = self.src.copy()

    def forward_keys(self, max_dist, fma):
        keys = np.full(len(self.cur), INT64_MAX, np.int64)
        if len(self.tgt):
            idx, d2 = self.orc.nn(self.cur, self.tgt, fma=fma, kdtree=self.kd)
            keys = (np.ascontiguousarray(d2, np.float32).view(np.uint32).astype(np.int64) << 32) | self.gidx[idx]
        return keys

    def moments_from_keys(self, keys, max_dist, origin, reciprocal, fma):
        row = np.zeros(32)
        row[1:4] = origin
        ok = keys != INT64_MAX
        d2 = (keys >> 32).astype(np.uint32).view(np.float32)
        jg = keys & 0xFFFFFFFF
        pos = np.searchsorted(self.gidx, jg)
        pos = np.minimum(pos, max(len(self.gidx) - 1, 0))
        owned = ok & (len(self.gidx) > 0) & (self.gidx[pos] == jg if len(self.gidx) else False)
        cand = np.nonzero(owned & (d2.astype(np.float64) <= max_dist * max_dist))[0]
        j = pos[cand]
        if reciprocal and len(cand):
            uj, inv = np.unique(j, return_inverse=True)
            bi, bd2 = self.orc.nn(self.tgt[uj], self.cur, fma=fma, kdtree=self.kd)
            keep = (bi[inv] == cand) & (bd2[inv].astype(np.float64) <= max_dist * max_dist)
            cand, j = cand[keep], j[keep]
        if len(cand):
            p = self.cur[cand, :3].astype(np.float64) - origin
            t = self.tgt[j, :3].astype(np.float64) - origin
            iu = np.triu_indices(3)
            row[0] = len(cand)
            row[4:7], row[7:10] = p.sum(0), t.sum(0)
            row[10:16], row[16:22], row[22:31] = (p.T @ p)[iu], (t.T @ t)[iu], (p.T @ t).ravel()
            row[31] = d2[cand].astype(np.float64).sum()
        self.last_pairs = (cand, self.gidx[j] if len(cand) else np.zeros(0, np.int64))
        return row

    def transform_current(self, T):
        self.cur = self.orc.transform_f32(T, self.cur)

    def finish(self, final):
        self.out = self.orc.transform_f32(final, self.src)

    def download_out(self):
        return self.out

    def min_into(self, a, b):
        np.minimum(a, b, out=a)

    def add_into(self, a, b):
        a += b

    def keys_to_host(self, keys):
        return keys

    def neutral_keys(self):
        return np.full(len(self.cur), INT64_MAX, np.int64)

    def neutral_row(self):
        return np.zeros(32)

    def set_status(self, row, failed):
        row[1:4] = 0.0
        row[1] = 1.0 if failed else 0.0

    def row_to_host(self, row):
        return row

    def close(self):
        pass
