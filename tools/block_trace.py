#!/usr/bin/env python3
"""tools/block_trace.py -- per-block timeline of ONE forward culled-NN launch (diagnostic build, tools/build_variant.sh
stamp -DMVR_STAMP): when does each workgroup start and end, how many cells did it evaluate, on which XCC?
    python3 tools/block_trace.py [n_points] [cull_w]      (on the GPU box, with the stamp library in place)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mvr = importlib.import_module("multi-view-registration_amd")
sp = mvr.synth_params(12, 3)
a, b = mvr.synth_view(sp, 0, n), mvr.synth_view(sp, 1, n)
piv, ax = mvr.synth_prior(sp)
path = os.path.join(ROOT, "gpurun_out", "block_trace_w%d.txt" % w)
with mvr.Context(0) as ctx:
    ctx.tune(cull_w=w)
    ctx.upload(0, a); ctx.upload(1, b)
    ctx.transform(1, 1, mvr.axis_rotation(piv, ax, mvr.turntable_angle(1, 12)))
    ctx.correspondences(1, 0, 4.0, reciprocal=False)          # warm-up (index build)
    ctx.correspondences(1, 0, 4.0, reciprocal=False)          # forward search only
    os.environ["MVR_STAMP_TRACE"] = path
    ctx.debug_counters()
t = np.loadtxt(path, dtype=np.uint64)
blk, st, en, cells, hw = t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 4]
marks, nquads = t[:, 5:13].astype(np.float64), t[:, 13]
ntests, n_open, n_exp = t[:, 14], t[:, 15] >> np.uint64(32), t[:, 15] & np.uint64(0xFFFFFFFF)
t_open, t_exp, t_next = (t[:, 16] >> np.uint64(40)).astype(float) / 100, ((t[:, 16] >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(float) / 100, (t[:, 16] & np.uint64(0xFFFFF)).astype(float) / 100
t0 = st.min()
st = (st - t0).astype(np.float64) / 100.0; en = (en - t0).astype(np.float64) / 100.0          # us
dur = en - st
print("blocks %d  kernel span %.1f us  block duration mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us" %
      (len(blk), en.max(), dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("start times: p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us" % tuple(np.percentile(st, [50, 90, 99, 100])))
edges = np.arange(0, en.max() + 10, 10.0)
print("t(us)   started  ended  running")
for lo in edges:
    print("%6.0f  %7d %6d %8d" % (lo, ((st >= lo) & (st < lo + 10)).sum(), ((en >= lo) & (en < lo + 10)).sum(), ((st < lo + 5) & (en > lo + 5)).sum()))
print("cells -> mean duration (us), count")
for c in sorted(set(cells.tolist())):
    m = cells == c
    print("  %3d  %7.1f  %5d" % (c, dur[m].mean(), m.sum()))
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
print("per-XCC blocks:", np.bincount(xcc)[:8], " last end per XCC (us):", [round(float(en[xcc == k].max()), 1) for k in range(8) if (xcc == k).any()])
order = np.argsort(-dur)[:12]
print("longest blocks (id, start, dur, cells):", [(int(blk[k]), round(float(st[k]), 1), round(float(dur[k]), 1), int(cells[k])) for k in order])

# wave 0's phase marks relative to the block start, for the blocks resident from t = 0 and for the late ones
names = ["prologue done", "1st quad chosen", "1st stage done", "2nd quad chosen+fetched", "1st process done", "loop end", "merge barrier", "keys written"]
rel = (marks - (t[:, 1].astype(np.float64))[:, None]) / 100.0
for label, sel in (("resident at t=0", st < 5), ("started later", st >= 5)):
    for nq in (1, 2, 3):
        m = sel & (nquads == nq) & (marks.min(1) > 0)
        if m.sum() < 5: continue
        print("%s, wave 0 ran %d quad(s): %d blocks, block duration %.1f us" % (label, nq, m.sum(), dur[m].mean()))
        print("   " + "  ".join("%s %.1f" % (nm, v) for nm, v in zip(names, rel[m].mean(0))))
        print("   wave 0: %.1f box tests, %.1f ballot blocks opened (%.2f us in all), %.1f tiles expanded (%.2f us), next_quad total %.2f us" %
              (ntests[m].mean(), n_open[m].mean(), t_open[m].mean(), n_exp[m].mean(), t_exp[m].mean(), t_next[m].mean()))
