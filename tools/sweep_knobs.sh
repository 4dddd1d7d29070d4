#!/bin/bash
# the 12 x 200k step under a list of knob settings, each twice (same box):  tools/sweep_knobs.sh "grid_cell_points=3" "grid_cell_points=6" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sweep; mkdir -p $O; cd $R
for rep in 1 2; do for k in "pipeline=1" "$@"; do
  MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 $k 2>> $O/err.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('%-40s %.4f ms/step  n_corr %d' % (json.dumps(d['knobs']), d['ms_per_step'], d['n_corr']))" >> $O/sweep.txt || exit 1
done; done
cat $O/sweep.txt
