#!/bin/bash
# kernel timeline of the passes right after a restart from the prior (every seed a millimetre or two off): the first passes of
# the bench's timed window, kernel by kernel, next to a settled pass.
#   tools/restart_trace.sh <tag> [knob=value ...]     -> gpurun_out/restart_<tag>/{timeline.txt, probe.json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
O=$R/gpurun_out/restart_$tag; rm -rf $O; mkdir -p $O
MVR_PROBE_PROF=0 MVR_PROBE_PASSLOG=1 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/step_probe.py 12 200000 10 25 "$@" > $O/probe.json 2> $O/probe.err || exit 1
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" > $O/timeline.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('mvr::(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
idx = [i for i, r in enumerate(rows) if 'refresh_sorted' in r['Kernel_Name']]
idx.append(len(rows))
for w in (-11, -10, -9, -8, -2):
    i0, i1 = idx[w], idx[w + 1]; t0 = int(rows[i0]['Start_Timestamp']); prev = None
    print("pass %d of the window" % (w + 12))
    tot = 0.0
    for r in rows[i0:i1]:
        s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0; tot += (e - s) / 1e3
        print("%8.1f %8.1f  dur %7.1f  gap %6.1f  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, nm(r)))
        prev = e
    print("   kernels %.1f us" % tot)
P
cat $O/timeline.txt; cat $O/probe.json
