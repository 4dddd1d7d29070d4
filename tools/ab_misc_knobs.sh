#!/bin/bash
# same-box sweep of the remaining routing knobs on the final walk: 40 passes from the prior, pass log
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "" "grid_wide_waves=4" "grid_wide_waves=64" "cull_slices=1" "cull_slices=2" "cull_slices=4" "cull_slices=8" "grid_sets=2" "cull_list_w=4" "grid_light_rows=20" "grid_light_rows=28" "grid_probe_rows=10" "grid_probe_rows=14"; do
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 $v | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['pass_ms']
print('%-22s ms/step %.4f  first4 sum %.3f  last20 %.4f' % (j['knobs'], j['ms_per_step'], sum(p[:4]), sum(p[-20:])/20))"
done; done
