#!/bin/bash
# per-kernel microseconds of a settled pass for library variants (tools/build_variant.sh), same box; '' = the tree's library.
#   tools/ab_variant_kernels.sh "" tw8 ...      (each variant first passes the ring and exactness tests)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/abvk
for v in "$@"; do
  if [ -n "$v" ]; then
    MVR_LIB_VARIANT=$v timeout -k 10 500 python -m pytest tests/test_gpu_ring.py tests/test_gpu_exact.py -x -q -m gpu -p no:cacheprovider > gpurun_out/abvk/pytest_$v.log 2>&1 || { tail -n 20 gpurun_out/abvk/pytest_$v.log; exit 1; }
    echo "variant $v: $(tail -n 1 gpurun_out/abvk/pytest_$v.log)"
  fi
done
for round in 1 2; do for v in "$@"; do
  echo -n "[${v:-tree}] "; MVR_LIB_VARIANT=$v tools/kernels_by_knob.sh "pipeline=1" | tr ',' '\n' | grep -E "kernels|nn_grid|tail|wide" | tr '\n' ' '; echo
done; done
