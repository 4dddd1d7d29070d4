#!/bin/bash
# same-box A/B of the accept/sums launch: queries a lane walks at once (build variants un2 / un4 / un6; default 3) x partial rows
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "default 0" "un2 0" "un4 0" "un6 0" "default 48" "default 96" "un4 48"; do set -- $v
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 $([ $1 = default ] || echo lib=$1) reduce_rows=$2 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['pass_ms']
print('%-8s rows=%-3s  ms/step %.4f  last20 %.4f' % (j['lib'], j['knobs'].get('reduce_rows'), j['ms_per_step'], sum(p[-20:])/20))"
done; done
