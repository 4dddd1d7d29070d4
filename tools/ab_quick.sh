#!/bin/bash
# quick same-box check of a change: the ring / exactness tests, then the 12 x 200k step (pipelined and not), then the kernel timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_quick; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_ring.py tests/test_gpu_exact.py tests/test_gpu_world.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for k in "pipeline=1" "pipeline=0" "pipeline=1" "pipeline=0"; do
  MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 $k >> $O/ab.log 2>> $O/ab.err || exit 1
done
cut -c1-330 $O/ab.log
timeout -k 10 200 tools/timeline.sh quick pipeline=1 > $O/timeline.txt 2>&1; tail -32 $O/timeline.txt | cut -c1-200
