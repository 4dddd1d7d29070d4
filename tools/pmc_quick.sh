#!/bin/bash
# one PMC pass (VALU / SALU / LDS instruction counts + wave cycles) of a probe for the library variants given
#   tools/pmc_quick.sh "<probe command relative to the repo>" variant...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
probe=$1; shift
for v in "$@"; do
  cp $R/build/libmvr_hip_$v.so $R/multi-view-registration_amd/libmvr_hip.so || exit 1
  O=$R/gpurun_out/pmc_quick_$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
    --output-format csv -d $O -- python3 $R/$probe > $O/log.txt 2>&1 || exit 1
done
