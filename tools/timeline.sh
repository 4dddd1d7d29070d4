#!/bin/bash
# kernel timeline of the ring step in steady state, WITH the gaps between kernels (rocprofv3 --kernel-trace of tools/step_probe.py):
#   tools/timeline.sh <tag> [knob=value ...]     -> gpurun_out/timeline_<tag>/{timeline.txt, probe.json}
# timeline.txt: one pass in the middle of the run (start, end, duration, gap to the previous kernel's end), then the
# means over the steady passes: sum of kernel durations, sum of gaps, pass period (first kernel to first kernel).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
O=$R/gpurun_out/timeline_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/step_probe.py 12 200000 40 20 "$@" > $O/probe.json 2> $O/probe.err || exit 1
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" > $O/timeline.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('mvr::(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
idx = [i for i, r in enumerate(rows) if 'refresh_sorted' in r['Kernel_Name']]
w = len(idx) - 12
i0, i1 = idx[w], idx[w + 1]; t0 = int(rows[i0]['Start_Timestamp']); prev = None
for r in rows[i0:i1 + 1]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print("%8.1f %8.1f  dur %7.1f  gap %6.1f  q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, r.get('Queue_Id', '?'), nm(r)))
    prev = e
per, dur, gaps = [], [], []
for a, b in zip(idx[-22:-2], idx[-21:-1]):
    ks = rows[a:b]
    per.append((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
    dur.append(sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ks) / 1e3)
    gaps.append(sum(int(ks[i + 1]['Start_Timestamp']) - int(ks[i]['End_Timestamp']) for i in range(len(ks) - 1)) / 1e3)
n = len(per)
print("steady passes %d: period %.1f us, kernels %.1f us, gaps inside a pass %.1f us, between passes (last kernel -> next pass's first) %.1f us" %
      (n, sum(per) / n, sum(dur) / n, sum(gaps) / n, sum(per) / n - sum(dur) / n - sum(gaps) / n))
P
cat $O/timeline.txt; cat $O/probe.json
