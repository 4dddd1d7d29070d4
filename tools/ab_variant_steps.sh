#!/bin/bash
# same-box A/B of library variants (tools/build_variant.sh; '' = the tree's) by the 40-pass window from the prior: whole, first four,
# first ten (the bench's window), last twenty (settled)
#   tools/ab_variant_steps.sh "" bp160 bp224
R=$GRAFT_REPO_ROOT; cd $R
for r in 1 2; do for v in "$@"; do echo -n "[${v:-tree}] "; MVR_LIB_VARIANT=$v MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 pipeline=1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['pass_ms']; print('ms/step %.4f first4 %s first10 %.4f last20 %.4f n_corr %d' % (j['ms_per_step'], [round(x,3) for x in p[:4]], sum(p[:10])/10, sum(p[-20:])/20, j['n_corr']))"; done; done
