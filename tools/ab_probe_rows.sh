#!/bin/bash
# same-box A/B of the probe's threshold (rows of cells): a 20-pass window restarted from the prior, pass by pass
cd $GRAFT_REPO_ROOT
for r in 1 2; do for k in 12 8 6 4 3; do
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 20 25 grid_probe_rows=$k | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['pass_ms']
print('probe_rows=%s  ms/step %.4f  first4 %s  last10 %.4f  n_corr %d' % (j['knobs'].get('grid_probe_rows'), j['ms_per_step'], p[:4], sum(p[-10:])/10, j['n_corr']))"
done; done
