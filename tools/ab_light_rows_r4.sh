#!/bin/bash
# round 4 (staged forward walk): same-box sweep of the rows of cells a thread walks itself x the probe's threshold; the first four
# passes of a window restarted from the prior next to the settled ones
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in ${SWEEP:-24:12 12:12 16:12 8:8 16:8 24:8 24:6 32:12}; do set -- ${v/:/ }
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 grid_light_rows=$1 grid_probe_rows=$2 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['pass_ms']
print('light_rows=%s probe_rows=%s  ms/step %.4f  first4 %s sum %.3f  first10 %.4f  last20 %.4f  n_corr %d' % (j['knobs']['grid_light_rows'], j['knobs']['grid_probe_rows'], j['ms_per_step'], [round(x,3) for x in p[:4]], sum(p[:4]), sum(p[:10])/10, sum(p[-20:])/20, j['n_corr']))"
done; done
