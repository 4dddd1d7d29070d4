#!/bin/bash
# kernel timeline of one bench step (rocprofv3 --kernel-trace): tools/trace_step.sh <views> <points> [step index]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_step; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --views $1 --points $2 --steps 12 --warmup 3 --no-cpu-baseline --no-bruteforce-pass > $O/bench.json 2> $O/bench.err || exit 1
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" "${3:-8}" <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'refresh_sorted' in r['Kernel_Name']]
w=int(sys.argv[2]); i0,i1=idx[w],idx[w+1]; t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1+1]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    name=r['Kernel_Name'].replace('mvr::(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
    print("%8.1f %8.1f  dur %7.1f  q%s %s" % (s/1e3, e/1e3, (e-s)/1e3, r.get('Queue_Id','?'), name))
P
