#!/bin/bash
# kernel timeline of one sequential-mode align (rocprofv3 --kernel-trace of tools/seq_bench.py): the last align of the culled run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_seq; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/seq_bench.py --no-cpu --no-brute > $O/seq.json 2> $O/seq.err || exit 1
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'nn_cull' in r['Kernel_Name']]
# an align = fwd + reverse cull launches; take a window around the 10th-from-last cull launch
i=idx[-12]; t0=int(rows[i-8]['Start_Timestamp'])
for r in rows[i-8:i+40]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print("%8.1f %8.1f  dur %7.1f  %s" % (s/1e3, e/1e3, (e-s)/1e3, r['Kernel_Name'].split('(')[0].split('::')[-1][:50]))
P
