#!/bin/bash
# every GPU operation of ONE settled pipelined pass in order (rocprofv3 --kernel-trace of tools/step_probe.py), the runtime's own
# kernels (fills, copies, stream write / wait operations) included:   tools/trace_pass.sh [knob=value ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_pass; rm -rf $O; mkdir -p $O
MVR_PROBE_PROF=0 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/step_probe.py 12 200000 40 20 pipeline=1 "$@" > $O/probe.json 2> $O/probe.err || { tail -n 5 $O/probe.err; exit 1; }
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'refresh_sorted' in r['Kernel_Name']]
i0,i1=idx[-6],idx[-5]; t0=int(rows[i0]['Start_Timestamp']); prev=None
for r in rows[i0:i1+1]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    name=r['Kernel_Name'].replace('mvr::(anonymous namespace)::','').replace('mvr::','').replace('void ','').split('(')[0][:60]
    print("%8.1f %8.1f  dur %7.1f  gap %6.1f  q%s grid %s  %s" % (s/1e3, e/1e3, (e-s)/1e3, 0.0 if prev is None else (s-prev)/1e3, r.get('Queue_Id','?'), r.get('Grid_Size','?'), name))
    prev=e
P
