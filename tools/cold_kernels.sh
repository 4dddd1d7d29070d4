#!/bin/bash
# the GPU side of a registration's FIRST pass in a warm process: every kernel of the second registration's first pass in order, with
# stream (queue) and gaps (rocprofv3 --kernel-trace of tools/cold_probe.py ... reps=2)
#   tools/cold_kernels.sh [views] [points]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cold_kernels; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/cold_probe.py ${1:-12} ${2:-200000} 3 one_call=1 reps=2 > $O/probe.json 2> $O/probe.err || { tail -n 5 $O/probe.err; exit 1; }
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('mvr::(anonymous namespace)::','').replace('mvr::','').replace('void ','').split('(')[0][:60]
# the second registration: from the last bbox/hilbert batch launch (its orderings) to the second accept_moments2 after it
starts=[i for i,r in enumerate(rows) if 'hilbert_many_kernel' in r['Kernel_Name']]
i0=starts[-1]-1
ends=[i for i,r in enumerate(rows) if i>i0 and 'moments2_final' in r['Kernel_Name']]
i1=ends[0]
t0=int(rows[i0]['Start_Timestamp']); busy=0.0; last_end=t0
for r in rows[i0:i1+1]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print("%8.1f %8.1f  dur %7.1f  q%-3s %s" % (s/1e3, e/1e3, (e-s)/1e3, r.get('Queue_Id','?'), nm(r)))
print("first pass on the GPU: %.1f us from its first to its last kernel" % ((int(rows[i1]['End_Timestamp'])-t0)/1e3))
P
