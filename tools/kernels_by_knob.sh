#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt; mkdir -p $O
: > $O/kern.txt
for v in "$@"; do
  rm -rf $O/t; mkdir -p $O/t
  MVR_PROBE_PROF=0 timeout -k 10 ${PROBE_TIMEOUT:-200} rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/step_probe.py ${PROBE_VIEWS:-12} ${PROBE_POINTS:-200000} ${PROBE_STEPS:-40} ${PROBE_WARM:-20} pipeline=1 $v > $O/t/probe.json 2> $O/t/probe.err || exit 1
  python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" "$v" >> $O/kern.txt <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('mvr::(anonymous namespace)::', '').replace('mvr::', '').replace('void ', '').split('(')[0][:40]
idx = [i for i, r in enumerate(rows) if 'refresh_sorted' in r['Kernel_Name']]
acc = collections.OrderedDict(); n = 0
for a, b in zip(idx[-22:-2], idx[-21:-1]):
    seen = collections.Counter(); n += 1
    for r in rows[a:b]:
        k = nm(r)
        if k.startswith('__amd'): k = 'runtime' if 'streamOpsWait' not in k else None
        if k is None: continue
        seen[k] += 1; key = k if k.startswith('runtime') else "%s #%d" % (k, seen[k])
        acc[key] = acc.get(key, 0.0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(acc.values()) / n
print("%-22s kernels %.1f us/pass: " % (sys.argv[2], tot) + ", ".join("%s %.1f" % (k.replace('_batch_kernel', '').replace('_kernel', ''), v / n) for k, v in acc.items()))
P
done
cat $O/kern.txt
