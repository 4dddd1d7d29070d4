#!/bin/bash
# same-box sweep of the rows of cells a thread walks itself (wider balls leave for a wave / the listed sets) x the probe's threshold, on the final walk
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "12 12" "24 12" "24 24" "32 12" "32 32" "48 12" "64 16"; do set -- $v
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 grid_light_rows=$1 grid_probe_rows=$2 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['pass_ms']
print('light_rows=%s probe_rows=%s  ms/step %.4f  first4 %s sum %.3f  last20 %.4f  n_corr %d' % (j['knobs']['grid_light_rows'], j['knobs']['grid_probe_rows'], j['ms_per_step'], [round(x,3) for x in p[:4]], sum(p[:4]), sum(p[-20:])/20, j['n_corr']))"
done; done
