#!/bin/bash
# same-box A/B of the grid search's wave-staged walk (grid_stage = 0 / N) on the 12 x 200k ring step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_stage; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ring.py -x -q -p no:cacheprovider > $O/pytest_ring.log 2>&1 || { tail -30 $O/pytest_ring.log; exit 1; }
tail -1 $O/pytest_ring.log
for k in "grid_stage=0" "grid_stage=512" "grid_stage=256" "grid_stage=1024" "grid_stage=0" "grid_stage=512"; do
  MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 $k >> $O/ab.log 2>> $O/ab.err || exit 1
done
for k in "grid_stage=0" "grid_stage=512"; do
  timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 $k >> $O/ab_prof.log 2>> $O/ab.err || exit 1
done
cat $O/ab.log $O/ab_prof.log | cut -c1-400
