#!/bin/bash
# the align's round trip: the sums launch stores the iteration's row to the host and the host spins (align_spin 1) vs a copy and
# hipStreamSynchronize (0); exactness tests first, then the sequential mode as registrationICP runs it, ms per align by sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_align_spin; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_exact.py tests/test_gpu_seq.py tests/test_gpu_parity.py tests/test_gpu_shim.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for m in 1 0 1 0; do
  MVR_ALIGN_SPIN=$m timeout -k 10 200 python3 tools/seq_bench.py --no-cpu --no-brute --repeat 5 2>> $O/err.txt | sed "s/^/align_spin=$m /" >> $O/ab_align_spin.log || exit 1
done
cut -c1-520 $O/ab_align_spin.log
