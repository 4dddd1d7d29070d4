#!/bin/bash
# same-box sweep of the grid's cell size (points per occupied cell aimed at) x rows a thread walks itself, on the walk that reads
# four consecutive points per round: 40 passes from the prior, pass log
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "4 12" "3 12" "5 12" "6 12" "8 12" "6 9" "8 9" "5 16"; do set -- $v
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 grid_cell_points=$1 grid_light_rows=$2 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['pass_ms']
print('cell_points=%s light_rows=%s  ms/step %.4f  first4 %s  last20 %.4f' % (j['knobs']['grid_cell_points'], j['knobs']['grid_light_rows'], j['ms_per_step'], [round(x,3) for x in p[:4]], sum(p[-20:])/20))"
done; done
