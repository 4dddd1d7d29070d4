#!/bin/bash
# same-box A/B of the cell-start tables (grid_index = 0 dense / 1 compact): the 12 x 200k step and its kernels, a cold registration
# at 12 x 200k, and (unless MVR_AB_BIG=0) the 36 x 1M step and cold registration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_index; mkdir -p $O; cd $R
: > $O/ab.log
for v in 0 1 2 0 1 2; do
  MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 pipeline=1 grid_index=$v >> $O/ab.log 2>> $O/ab.err || exit 1
done
for v in 0 1 2; do
  timeout -k 10 200 python3 tools/cold_probe.py 12 200000 8 one_call=1 reps=2 grid_index=$v 2>> $O/ab.err | tail -1 >> $O/ab.log || exit 1
done
if [ "$MVR_AB_BIG" != "0" ]; then
  for v in 0 1 2; do
    MVR_PROBE_PROF=0 timeout -k 10 300 python3 tools/step_probe.py 36 1000000 6 4 pipeline=1 grid_index=$v >> $O/ab.log 2>> $O/ab.err || exit 1
    timeout -k 10 300 python3 tools/cold_probe.py 36 1000000 4 one_call=1 reps=2 grid_index=$v 2>> $O/ab.err | tail -1 >> $O/ab.log || exit 1
  done
fi
python3 - $O/ab.log <<'P'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    if "ms_per_pass" in r: print("cold  %dx%d %-16s passes %s total %.2f" % (r["views"], r["n"], r["knobs"], r["ms_per_pass"][:5], r["total_ms"]))
    else: print("step  %dx%d %-40s ms_per_step %.4f n_corr %d" % (r["views"], r["n"], {k: v for k, v in r["knobs"].items() if k != "pipeline"}, r["ms_per_step"], r["n_corr"]))
P
cd /tmp
: > $O/kern.txt
for v in 0 1 2; do
  rm -rf $O/t; mkdir -p $O/t
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/step_probe.py 12 200000 40 20 pipeline=1 grid_index=$v > $O/t/probe.json 2> $O/t/probe.err || exit 1
  python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" "index=$v" >> $O/kern.txt <<'P'
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
cat $O/kern.txt
