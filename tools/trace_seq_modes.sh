#!/bin/bash
# per-kernel totals of the sequential sweep (tools/seq_bench.py) under rocprofv3 --kernel-trace, for each seq_search mode given
#   tools/trace_seq_modes.sh "<modes, e.g. 1 3>" [sweeps] [extra env, e.g. MVR_TUNE=...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in $1; do
  O=$R/gpurun_out/trace_seq_mode$m; rm -rf $O; mkdir -p $O
  export MVR_SEQ_SEARCH=$m
  rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/seq_bench.py --no-cpu --no-brute --repeat ${2:-2} > $O/seq.json 2> $O/seq.err || { tail -n 20 $O/seq.err; exit 1; }
  echo "== seq_search=$m"; python3 -c "
import json,sys
d=json.loads([l for l in open('$O/seq.json') if l.startswith('{')][-1]); g=d['gpu_culled']; print(g['ms_per_align'], g['native_ms_per_align_by_sweep'], g['evals'])"
  python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" <<'P'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.OrderedDict()
for r in rows:
    k=r['Kernel_Name'].replace('mvr::(anonymous namespace)::','').replace('void ','').split('(')[0][:70]
    a=agg.setdefault(k,[0,0.0,0.0]); d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; a[0]+=1; a[1]+=d; a[2]=max(a[2],d)
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:22]:
    print("%7d x %8.1f us mean  %8.1f max  %10.1f total  %s" % (a[0], a[1]/a[0], a[2], a[1], k))
P
done
