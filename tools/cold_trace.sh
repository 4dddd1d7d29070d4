#!/bin/bash
# where the first passes' time goes: HIP API + kernel statistics of tools/cold_probe.py (3 passes)
#   tools/cold_trace.sh <tag> [views] [points]    -> gpurun_out/cold_<tag>/{hip_api_stats.csv, kernel_stats.csv, probe.json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out/cold_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/cold_probe.py ${2:-12} ${3:-200000} 3 > $O/probe.json 2> $O/probe.err || exit 1
for k in hip_api_stats kernel_stats; do f=$(find $O -name "*${k}.csv" | head -1); [ -n "$f" ] && cp $f $O/$k.csv; done
python3 - $O <<'P'
import csv, sys
for k in ("hip_api_stats", "kernel_stats"):
    try: rows = list(csv.DictReader(open(sys.argv[1] + "/" + k + ".csv")))
    except Exception as e: print(k, "missing", e); continue
    print("==", k)
    for r in rows[:16]:
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ("%.1f" % (float(r['TotalDurationNs']) / 1e3)).rjust(11), "us total", ("%.1f" % (float(r['AverageNs']) / 1e3)).rjust(9), "us avg")
P
cat $O/probe.json
