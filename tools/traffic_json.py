#!/usr/bin/env python3
"""tools/traffic_json.py <dir with FETCH_SIZE/ and WRITE_SIZE/ rocprofv3 --pmc outputs of tools/step_probe.py>
-> the JSON kept as profiles/nn_cull_traffic.json: per-launch HBM-side bytes of the fused culled search (forward and
reverse launches of the timed steps, warm-up dispatches dropped), with the gfx950 correction of the guide (FETCH_SIZE
counts 128-byte requests as 64: doubled; WRITE_SIZE as is), the algorithmic bytes next to it and the hash of the
kernel source measured."""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
GRID = len(sys.argv) > 2 and sys.argv[2] == "grid"        # the grid-search launches (nn_grid_kernel) instead of the culled ones
_probe = json.loads([l for l in open(os.path.join(d, "FETCH_SIZE.log")) if l.startswith("{")][-1])      # the probe says what it ran
V, N, STEPS, WARM = _probe["views"], _probe["n"], _probe["steps"], _probe["warm"]
NAME = "nn_grid_kernel" if GRID else "nn_cull_kernel"


def per_dispatch(counter):
    rows = {}
    for f in glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if NAME in r["Kernel_Name"] and r["Counter_Name"] == counter:
                rows[int(r["Dispatch_Id"])] = rows.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    vals = [rows[k] for k in sorted(rows)]
    if V > 12:
        # more pairs than one fused launch holds: several launches per direction and step -- only per-STEP sums are meaningful
        per_step = len(vals) // (STEPS + max(WARM, 1) - (1 if GRID else 0))
        tail = vals[-per_step * STEPS:]
        return [sum(tail[i * per_step:(i + 1) * per_step]) / 2.0 for i in range(STEPS) for _ in (0, 1)]      # (spread evenly over a "forward" and a "reverse" slot)
    if GRID:
        return vals[-2 * STEPS:]              # the first pass of all takes the culled kernel; the timed steps are the last ones
    return vals[2 * max(WARM, 1):]            # forward + reverse per step; the warm-up steps come first


def other_kernels():
    """the step's other kernels from the same two passes: HBM-side bytes per launch (FETCH_SIZE doubled + WRITE_SIZE), mean over
    the last STEPS launches of each (one launch of each per step with MVR_PAIR_GROUPS=1 and up to 12 pairs)"""
    names = {"refresh_sorted_kernel": "mvr_index.hip", "accept_moments2_batch_kernel": "mvr_reduce.hip", "nn_grid_tail_kernel": "mvr_grid.hip",
             "nn_grid_wide_kernel": "mvr_grid.hip", "count_flags": "mvr_reduce.hip", "compact_flags": "mvr_reduce.hip", "moments2_final": "mvr_reduce.hip"}
    acc = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter: continue
                for nm in names:
                    if nm in r["Kernel_Name"]:
                        acc.setdefault(nm, {}).setdefault(counter, {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                        acc[nm][counter][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {}
    for nm, c in acc.items():
        f = [c.get("FETCH_SIZE", {})[k] for k in sorted(c.get("FETCH_SIZE", {}))][-STEPS:]
        w = [c.get("WRITE_SIZE", {})[k] for k in sorted(c.get("WRITE_SIZE", {}))][-STEPS:]
        if not f or not w: continue
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        src_ = os.path.join(ROOT, "multi-view-registration_amd", "csrc", names[nm])
        out[nm] = {"hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024, "fetch_size_kb": fk, "write_size_kb": wk, "launches_averaged": len(f),
                   "source_file": names[nm], "source_sha256": hashlib.sha256(open(src_, "rb").read()).hexdigest()}
    return out


fetch, write = per_dispatch("FETCH_SIZE"), per_dispatch("WRITE_SIZE")
assert len(fetch) == 2 * STEPS and len(write) == 2 * STEPS, (len(fetch), len(write))
kb = lambda v: sum(v) / len(v)
ff, fr, wf, wr = kb(fetch[0::2]), kb(fetch[1::2]), kb(write[0::2]), kb(write[1::2])
fwd, rev = 2 * ff * 1024 + wf * 1024, 2 * fr * 1024 + wr * 1024
probe = json.loads([l for l in open(os.path.join(d, "FETCH_SIZE.log")) if l.startswith("{")][-1])
m = probe["n_corr"] / V                      # accepted pairs per scan pair ~ distinct matched targets
# distinct bytes a launch has to touch, 16-byte points: forward = queries + target (points, 1/64 cell boxes of 32 B, ...) +
# keys written + start bounds preset; reverse = matched targets (gathered 16 B) + source cloud + bounds + keys
alg_fwd = V * (16 * N + 16 * N * (1 + 2 / 64.0) + 8 * N + 4 * N)
alg_rev = V * (16 * m + 4 * m + 4 * m + 16 * N * (1 + 2 / 64.0) + 8 * m)
src = os.path.join(ROOT, "multi-view-registration_amd", "csrc", "mvr_grid.hip" if GRID else "mvr_cull.hip")
if GRID:
    # the grid walk: query (16 B) + previous key or start bound (8 B) + key written (8 B) per query, the posed target once
    # (16 B per point, grid order) -- the candidates a query evaluates are that same target array, re-read through the caches
    alg_fwd = V * (32 * N + 16 * N)
    alg_rev = V * (32 * m + 4 * m + 16 * N)
# SURVEY 8(d)'s compulsory bytes (12-byte points, every array once per kernel): K2 = 12 Ns + 12 Nt + 8 Ns, K3 = 12 Nt' + 12 Ns + 4 Nt' + 4 Ns
s8d_fwd, s8d_rev = V * (12 * N + 12 * N + 8 * N), V * (12 * m + 12 * N + 4 * m + 4 * N)
out = {
    "survey_8d_bytes_forward": s8d_fwd, "survey_8d_bytes_reverse": s8d_rev, "survey_8d_bytes_per_launch": 0.5 * (s8d_fwd + s8d_rev),
    "ratio_to_survey_8d": 0.5 * (fwd + rev) / (0.5 * (s8d_fwd + s8d_rev)),
    "fetch_uncorrected_bytes_per_launch": 0.5 * ((ff + fr) * 1024 + (wf + wr) * 1024),
    "calibration_note": "the x2 of the guide holds for wide coalesced reads; tools/exp_fetch.hip (profiles/r03_*_exp_fetch.txt) measures what this counter "
                        "reports for one-line-per-lane gathers, the walk's dominant access: the truth lies between the uncorrected and the doubled figure",
    "per_step_note": ("more than 12 pairs: the launches of a step are summed and halved -- forward / reverse figures are the per-step mean of both" if V > 12 else None),
    "hbm_bytes_per_step": fwd + rev,
    "kernel": ("nn_grid_kernel<false,1> (exact grid walk of the bounded queries, fused launch over the %d scan pairs of a ring step)" if GRID else
               "nn_cull_kernel<false,1,1> (exact culled NN, fused launch over the %d scan pairs of a ring step)") % V,
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/step_probe.py %d %d %d %d, MVR_PAIR_GROUPS=1 "
              "(tools/measure_traffic.sh); mean over the %d timed steps" % (V, N, STEPS, WARM, STEPS),
    "fetch_size_kb_forward": ff, "fetch_size_kb_reverse": fr, "write_size_kb_forward": wf, "write_size_kb_reverse": wr,
    "correction": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; WRITE_SIZE as is",
    "hbm_bytes_forward": fwd, "hbm_bytes_reverse": rev, "hbm_bytes_per_launch": 0.5 * (fwd + rev),
    "algorithmic_bytes_forward": alg_fwd, "algorithmic_bytes_reverse": alg_rev, "algorithmic_bytes_per_launch": 0.5 * (alg_fwd + alg_rev),
    "ratio_to_algorithmic": 0.5 * (fwd + rev) / (0.5 * (alg_fwd + alg_rev)),
    "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
    "memory_side_note": ("at 12 x 200k the working set (~100 MB of posed points and index lines) sits in the 256 MB Infinity Cache: these counters tally "
                         "L2 <-> fabric requests, MALL hits included -- they are an upper bound of the HBM bytes there; at 36 x 1M (1.7 GB) they are HBM traffic"),
    "other_kernels": (other_kernels() if GRID else None),
    "probe": probe,
}
print(json.dumps(out, indent=1))
