#!/bin/bash
# same-box sweep on the round-4 walk (forward staged): cell size, rows walked in-thread, probe threshold: 40 passes from the prior, pass log
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "grid_cell_points=4" "grid_cell_points=3" "grid_cell_points=5" "grid_cell_points=6" "grid_cell_points=8" "grid_cell_points=5 grid_light_rows=16" "grid_light_rows=16" "grid_light_rows=32" "grid_probe_rows=8" "grid_stage=2" "grid_stage=0"; do
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 $v 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=j['pass_ms']
print('%-44s ms/step %.4f  first4 %s  last20 %.4f' % (' '.join('%s=%s' % kv for kv in j['knobs'].items()), j['ms_per_step'], [round(x,3) for x in p[:4]], sum(p[-20:])/20))"
done; done
