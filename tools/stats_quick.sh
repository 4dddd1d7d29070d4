#!/bin/bash
# rocprofv3 kernel stats of a short bench run -> gpurun_out/stats_quick/ (top kernels printed)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stats_quick; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-bruteforce-pass > $O/bench.json 2> $O/bench.err || exit 1
f=$(find $O -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
python3 - "$O/kernel_stats.csv" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), ("%.1f" % float(r['AverageNs'])).rjust(10), r['Percentage'])
P
