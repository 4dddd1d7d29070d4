#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the culled NN kernel on the 200k pair.
# Output: gpurun_out/pmc_cull/pass{1..4}/... csv ; summarised by tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_cull
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $O/pass1 -- python3 $R/tools/nn_probe.py 200000 3 > $O/pass1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS \
  --output-format csv -d $O/pass2 -- python3 $R/tools/nn_probe.py 200000 3 > $O/pass2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pass3 -- python3 $R/tools/nn_probe.py 200000 3 > $O/pass3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pass4 -- python3 $R/tools/nn_probe.py 200000 3 > $O/pass4.log 2>&1 || exit 1
find $O -name "*counter_collection.csv"
