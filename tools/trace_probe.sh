#!/bin/bash
# kernel timeline of the last ring step of tools/step_probe.py (rocprofv3 --kernel-trace): tools/trace_probe.sh <tag> [knob=value ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/trace_$tag; rm -rf $O; mkdir -p $O
MVR_PROBE_PROF=0 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/step_probe.py 12 200000 12 25 "$@" > $O/probe.json 2> $O/probe.err || exit 1
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" > $O/timeline.txt <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'refresh_sorted' in r['Kernel_Name']]      # the first kernel of a step
i0,i1=idx[-2],idx[-1]-1; t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1+1]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    name=r['Kernel_Name'].replace('mvr::(anonymous namespace)::','').replace('mvr::','').replace('void ','').split('(')[0][:48]
    print("%8.1f %8.1f  dur %7.1f  q%s %s" % (s/1e3, e/1e3, (e-s)/1e3, r.get('Queue_Id','?'), name))
P
find $O -name '*.csv' -size +1M -delete; cat $O/timeline.txt
