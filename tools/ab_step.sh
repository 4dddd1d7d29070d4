#!/bin/bash
# A/B of the fused ring step over launch knobs (results never depend on them): tools/step_probe.py per combination.
#   gpurun -- tools/ab_step.sh <tag> "<knobs A>" "<knobs B>" ...
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
export MVR_PROBE_PROF=2
for cfg in "$@"; do
  for rep in 1 2; do
    echo "== $cfg (run $rep)" >> $O/ab.log
    timeout -k 10 120 python3 $R/tools/step_probe.py 12 200000 40 5 $cfg >> $O/ab.log 2>> $O/ab.err || exit 1
  done
done
python3 - <<PY
import json
for l in open("$O/ab.log"):
    if l.startswith("=="): print(l.strip(), end="  ")
    elif l.startswith("{"):
        d = json.loads(l); print("ms/step %.4f  ms/launch %.4f  evals/launch %.3e  timing %s" % (d["ms_per_step"], d["ms_per_launch"], d["evals_per_launch"], [round(t, 3) for t in d["timing_ms"]]))
PY
