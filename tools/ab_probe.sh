#!/bin/bash
# same-box A/B of the grid search's probe for wide balls (grid_probe = 0 / 1): tests, a window pass by pass, the 40-step mean
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_probe; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_ring.py tests/test_gpu_exact.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for k in "grid_probe=1" "grid_probe=0" "grid_probe=1" "grid_probe=0"; do
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 $k >> $O/ab.log 2>> $O/ab.err || exit 1
done
python3 - <<'P'
import json
for l in open("gpurun_out/ab_probe/ab.log"):
    d=json.loads(l); p=d["pass_ms"]; print(d["knobs"], "40-step mean %.4f" % d["ms_per_step"], "first 6:", p[:6], "last 10 mean %.4f" % (sum(p[-10:])/10), "10-step window %.4f" % (sum(p[:10])/10))
P
for k in "grid_probe=1" "grid_probe=0"; do timeout -k 10 200 python3 tools/cold_probe.py 12 200000 12 one_call=1 reps=2 $k 2>/dev/null | tail -1 | cut -c1-330; done
