#!/bin/bash
# same-box A/B of the partial rows (= blocks) of the sums launches: 40 passes from the prior, pass log
cd $GRAFT_REPO_ROOT
for r in 1 2; do for k in 0 64 85 128 170 512; do
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 reduce_rows=$k | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['pass_ms']
print('reduce_rows=%s  ms/step %.4f  last20 %.4f  n_corr %d' % (j['knobs'].get('reduce_rows'), j['ms_per_step'], sum(p[-20:])/20, j['n_corr']))"
done; done
