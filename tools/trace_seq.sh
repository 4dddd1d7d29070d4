#!/bin/bash
# kernel timeline of one sequential-mode align (rocprofv3 --kernel-trace of tools/seq_bench.py): the third-from-last align of the run
#   tools/trace_seq.sh <tag> [sweeps]     (two sweeps or more: an align whose searches start from the previous sweep's matches)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_seq${1:+_$1}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/seq_bench.py --no-cpu --no-brute --repeat ${2:-1} > $O/seq.json 2> $O/seq.err || exit 1
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
# an align's forward search is its one launch of the culled kernel; print the third-from-last align with the gaps, from the posing
# of its source (a few launches ahead of that search) to the same point of the next align
idx=[i for i,r in enumerate(rows) if 'nn_cull_kernel' in r['Kernel_Name']]
back=lambda i: max(j for j in range(i) if 'append_order_kernel' in rows[j]['Kernel_Name']) + 1
i0,i1=back(idx[-3]),back(idx[-2]); t0=int(rows[i0]['Start_Timestamp']); prev=None
for r in rows[i0:i1]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print("%8.1f %8.1f  dur %7.1f  gap %6.1f  %s" % (s/1e3, e/1e3, (e-s)/1e3, 0.0 if prev is None else (s-prev)/1e3, r['Kernel_Name'].replace('mvr::(anonymous namespace)::', '').replace('void ', '').split('(')[0][:50]))
    prev=e
P
