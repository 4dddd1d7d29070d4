#!/bin/bash
# the default bench line into gpurun_out/bench/<tag>.json, with a one-screen digest:  tools/run_bench.sh <tag> [bench flags]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bench; mkdir -p $O; cd $R; tag=$1; shift
( time timeout -k 10 900 python bench.py "$@" ) > $O/$tag.json 2> $O/$tag.err; echo "rc=$?"
grep -v amdgpu.ids $O/$tag.err | tail -6
python3 - $O/$tag.json <<'P'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rf = r.get("roofline") or {}
print("value %.4g %s  ms_per_step %.4f  settled %.4f  frac %.4f  avg_launch_ms %.4f  traffic %s" % (r["value"], r["unit"], r["ms_per_step"], (r.get("settled_window") or {}).get("ms_per_step", 0), rf.get("frac", 0), rf.get("avg_launch_ms", 0), rf.get("traffic")))
print("breakdown", r.get("step_breakdown_ms"))
print("projected", (r.get("projected_scaling") or {}).get("ms_per_step"))
st = r.get("secondary_stress_36x1M") or {}
print("stress", {k: st.get(k) for k in ("ms_per_step", "value")}, (st.get("roofline") or {}).get("frac"))
cr = r.get("cold_registration") or {}
print("cold", cr.get("ms_per_pass", [])[:6], cr.get("total_ms"))
sq = r.get("secondary_sequential") or {}
print("seq", sq.get("ms_per_align"), sq.get("ms_per_align_by_sweep"), sq.get("pose_delta_vs_oracle"))
print("cpu", (r.get("cpu_baseline") or {}).get("value"))
P
