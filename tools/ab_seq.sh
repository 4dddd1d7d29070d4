#!/bin/bash
# the sequential mode (BASELINE config 3: 12 x 200k, repeat_times = 5): exactness tests, then ms per align by sweep with and without
# the seeds one align of a scan leaves for the next
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_seq; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_exact.py tests/test_gpu_seq.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for m in "1 1" "0 1" "1 2" "0 2" "1 1" "0 1"; do
  set -- $m
  MVR_SEQ_SEED=$1 MVR_SEQ_SEARCH=$2 timeout -k 10 200 python3 tools/seq_bench.py --no-cpu --no-brute --repeat 5 2>> $O/err.txt | sed "s/^/seq_seed=$1 seq_search=$2 /" >> $O/ab_seq_seed.log || exit 1
done
cut -c1-420 $O/ab_seq_seed.log
