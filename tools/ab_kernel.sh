#!/bin/bash
# per-kernel A/B of library variants on one box: mean durations over the steady passes of the 12 x 200k step under rocprofv3 --kernel-trace
#   tools/ab_kernel.sh <variant|-> [<variant> ...]      ("-" = the tree's library; variants: tools/build_variant.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_kernel; mkdir -p $O
for v in "$@" "$@"; do
  [ "$v" = "-" ] && export MVR_LIB_VARIANT= || export MVR_LIB_VARIANT=$v
  rm -rf $O/t; mkdir -p $O/t
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/step_probe.py 12 200000 40 20 ${MVR_AB_KNOBS:-pipeline=1} > $O/t/probe.json 2> $O/t/probe.err || exit 1
  python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" "$v" >> $O/ab.txt <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('mvr::(anonymous namespace)::', '').replace('mvr::', '').replace('void ', '').split('(')[0][:40]
idx = [i for i, r in enumerate(rows) if 'refresh_sorted' in r['Kernel_Name']]
acc = collections.OrderedDict(); n = 0
for a, b in zip(idx[-22:-2], idx[-21:-1]):
    seen = collections.Counter(); n += 1
    for r in rows[a:b]:
        k = nm(r)
        if k.startswith('__amd'): k = 'runtime fills / copies / waits (not the gate)' if 'streamOpsWait' not in k else None
        if k is None: continue
        seen[k] += 1; key = k if k.startswith('runtime') else "%s #%d" % (k, seen[k])
        acc[key] = acc.get(key, 0.0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(acc.values()) / n
print("variant %-8s kernels %.1f us/pass: " % (sys.argv[2], tot) + ", ".join("%s %.1f" % (k.replace('_batch_kernel', '').replace('_kernel', ''), v / n) for k, v in acc.items()))
P
done
cat $O/ab.txt
