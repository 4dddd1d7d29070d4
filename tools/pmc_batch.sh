#!/bin/bash
# PMC passes for the fused batch launches of the culled kernel (12 x 200k ring): gpurun_out/pmc_batch/pass{1,2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_batch
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $O/pass1 -- python3 $R/tools/batch_probe.py 12 200000 3 > $O/pass1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $O/pass2 -- python3 $R/tools/batch_probe.py 12 200000 3 > $O/pass2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pass3 -- python3 $R/tools/batch_probe.py 12 200000 3 > $O/pass3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pass4 -- python3 $R/tools/batch_probe.py 12 200000 3 > $O/pass4.log 2>&1 || exit 1
find $O -name "*counter_collection.csv"
