#!/bin/bash
# same-box A/B of the tree's library against a variant built before the change (tools/build_variant.sh <name> on the old tree):
#   tools/ab_variant.sh <name> [<name> ...]     -> the ring / exactness tests on the tree's library, then the 12 x 200k step of each, twice round
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_variant; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_ring.py tests/test_gpu_exact.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for v in "" "$@" "" "$@"; do
  echo "variant='$v'" >> $O/ab.log
  MVR_LIB_VARIANT=$v MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 pipeline=1 >> $O/ab.log 2>> $O/ab.err || exit 1
  MVR_LIB_VARIANT=$v MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 10 0 pipeline=1 >> $O/ab.log 2>> $O/ab.err || exit 1
done
cut -c1-260 $O/ab.log
