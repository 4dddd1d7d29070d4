#!/bin/bash
# second set of PMC passes for the culled kernel: instruction fetch / scalar cache / memory-instruction latency levels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_cull2
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY \
  --output-format csv -d $O/pass1 -- python3 $R/tools/nn_probe.py 200000 3 > $O/pass1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL \
  --output-format csv -d $O/pass2 -- python3 $R/tools/nn_probe.py 200000 3 > $O/pass2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM \
  --output-format csv -d $O/pass3 -- python3 $R/tools/nn_probe.py 200000 3 > $O/pass3.log 2>&1 || exit 1
find $O -name "*counter_collection.csv"
