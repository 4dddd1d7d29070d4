#!/bin/bash
# counter passes over the grid-walk launches of the 12 x 200k ring step, staged walk on / off (separate --pmc passes, kernel trace only)
#   tools/pmc_stage.sh <tag> [knob=value ...]      -> gpurun_out/pmc_stage_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/pmc_stage_$tag; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for st in 2 0; do
    mkdir -p $O/p${i}_s$st
    echo "pass $i stage=$st: $set" >> $O/progress.txt
    MVR_PROBE_PROF=0 timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p${i}_s$st -- python3 $R/tools/step_probe.py 12 200000 6 25 pair_groups=1 grid_stage=$st "$@" > $O/p${i}_s$st/log.txt 2>&1 || echo "pass $i stage=$st failed" >> $O/summary.txt
  done
done
python3 - $O >> $O/summary.txt <<'P'
import csv, sys, glob, collections, os
for f in sorted(glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True)):
    rows = collections.defaultdict(dict); order = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'nn_grid_kernel' not in k: continue
        rows[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(rows)
    tagp = f.split('/')[-4] if 'pmc_stage' not in f.split('/')[-3] else f.split('/')[-3]
    # a step = forward launch then reverse launch: even positions forward
    for label, sel in (("forward", ids[0::2]), ("reverse", ids[1::2])):
        if not sel: continue
        acc = collections.defaultdict(float)
        for i in sel:
            for k, v in rows[i].items(): acc[k] += v
        print("%s %-8s launches %d  " % ([p for p in f.split('/') if p.startswith('p') and '_s' in p][0], label, len(sel)) + "  ".join("%s=%.4g" % (k, acc[k] / len(sel)) for k in sorted(acc)))
P
find $O -name '*.csv' -size +512k -delete; cat $O/summary.txt
