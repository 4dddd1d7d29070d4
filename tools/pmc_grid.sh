#!/bin/bash
# counter passes over the ring step of tools/step_probe.py for the grid-search launches: tools/pmc_grid.sh <tag> [knob=value ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/pmc_grid_$tag; rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); mkdir -p $O/p$i
  echo "pass $i: $set" >> $O/progress.txt; MVR_PROBE_PROF=0 timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/step_probe.py 12 200000 6 25 pair_groups=1 "$@" > $O/p$i/log.txt 2>&1 || echo "pass $i failed" >> $O/summary.txt
done
python3 - $O >> $O/summary.txt <<'P'
import csv, sys, glob, collections, os
for f in sorted(glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('mvr::(anonymous namespace)::','').replace('void ','').split('(')[0][:40]
        if os.environ.get('PMC_PATTERN', 'nn_') not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    for k in acc:
        n = len(cnt[k])
        print(k, 'launches', n, ' '.join('%s=%.4g' % (c, v / n) for c, v in sorted(acc[k].items())))
P
find $O -name '*.csv' -size +512k -delete; cat $O/summary.txt
