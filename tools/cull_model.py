#!/usr/bin/env python3
"""tools/cull_model.py -- CPU model of the culled search's work: how many distance evaluations per query
are NECESSARY for a given target box granularity, under the ideal bound (each query's final NN distance,
capped at max_dist)?  Used to size the tile/cell structure of csrc/mvr_cull.hip.
    python tools/cull_model.py [n_points] [max_dist]"""
import importlib, os, sys
import numpy as np
from scipy.spatial import cKDTree
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)

def hilbert_order(p):                         # Skilling's transpose algorithm, 10 bits per axis (as morton_kernel)
    lo, hi = p.min(0), p.max(0)
    X = np.clip((p - lo) / np.where(hi > lo, hi - lo, 1) * 1023.0, 0, 1023).astype(np.uint32).T.copy()
    Qb = 1 << 9
    while Qb > 1:
        P = Qb - 1
        for k in range(3):
            m = (X[k] & Qb) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = np.where(~m, (X[0] ^ X[k]) & P, 0)
            X[0] ^= t; X[k] ^= t
        Qb >>= 1
    X[1] ^= X[0]; X[2] ^= X[1]
    t = np.zeros_like(X[0]); Qb = 1 << 9
    while Qb > 1:
        t = np.where((X[2] & Qb) != 0, t ^ (Qb - 1), t); Qb >>= 1
    X ^= t
    code = np.zeros(len(p), np.uint64)
    for b in range(10):
        for k in range(3):
            code |= ((X[k].astype(np.uint64) >> b) & 1) << (3 * b + (2 - k))
    return np.argsort(code, kind="stable")

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    md = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    mvr = importlib.import_module("multi-view-registration_amd")
    sp = mvr.synth_params(12, 3)
    a, b = mvr.synth_view(sp, 0, n)[:, :3].astype(np.float64), mvr.synth_view(sp, 1, n)[:, :3].astype(np.float64)
    piv, ax = mvr.synth_prior(sp)
    R = np.asarray(mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 12)), np.float64).reshape(4, 4)
    q = b @ R[:3, :3].T + R[:3, 3]                     # queries: view 1 posed; targets: view 0
    t = a
    q = q[hilbert_order(q)]; t = t[hilbert_order(t)]
    d, _ = cKDTree(t).query(q)
    bound = np.minimum(d * d, md * md) * 1.00002
    print("n=%d max_dist=%g  queries with NN within max_dist: %.1f%%" % (n, md, 100 * (d <= md).mean()))
    for qs in (64, 32):
        for cell in (256, 128, 64, 32):
            nc = (len(t) + cell - 1) // cell
            lo = np.array([t[c * cell:(c + 1) * cell].min(0) for c in range(nc)]); hi = np.array([t[c * cell:(c + 1) * cell].max(0) for c in range(nc)])
            tot = 0; ncand = 0
            for s in range(0, len(q), qs):
                Q = q[s:s + qs]; B = bound[s:s + qs]
                slo, shi = Q.min(0), Q.max(0)
                gap = np.maximum(0, np.maximum(lo - shi, slo - hi)); cand = np.nonzero((gap * gap).sum(1) <= B.max())[0]     # box-box prefilter
                ncand += len(cand)
                g = np.maximum(0, np.maximum(lo[cand][None] - Q[:, None], Q[:, None] - hi[cand][None]))       # [q, c, 3]
                need = ((g * g).sum(2) <= B[:, None]).any(0)
                tot += need.sum() * cell * len(Q)
            print("  set=%3d box=%3d targets: %7.1f evals/query (%.2f boxes tested per set)" % (qs, cell, tot / len(q), ncand / (len(q) / qs)))

if __name__ == "__main__":
    main()
