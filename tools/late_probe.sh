#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/late; rm -rf $O; mkdir -p $O
MVR_PROBE_PROF=0 timeout -k 5 250 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/step_probe.py 12 200000 1500 25 pair_groups=1 > $O/probe.json 2> $O/probe.err || exit 1
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" <<'P'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'refresh_sorted' in r['Kernel_Name']]
def summarize(a,b,label):
    acc=collections.defaultdict(list)
    for s in range(a,b):
        for r in rows[idx[s]:idx[s+1]]:
            name=r['Kernel_Name'].replace('mvr::(anonymous namespace)::','').replace('mvr::','').replace('void ','').split('(')[0][:36]
            acc[name].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    span=(int(rows[idx[b]]['Start_Timestamp'])-int(rows[idx[a]]['Start_Timestamp']))/1e3/(b-a)
    print(label, 'period %.1f us' % span, ' '.join('%s=%.1f' % (k, sum(v)/ (b-a)) for k,v in acc.items()))
n=len(idx)
summarize(60,160,'early')
summarize(n-200,n-100,'late ')
P
find $O -name '*.csv' -size +1M -delete
