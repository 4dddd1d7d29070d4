#!/bin/bash
# Read requests of the L2's memory side by SIZE (TCC_EA0_RDREQ, _32B, _64B, _128B): the exact HBM-side read bytes of a kernel,
# without FETCH_SIZE's one-size-fits-all 64 bytes per request.  First on tools/exp_fetch.hip (known byte counts: a stream and
# one-line-per-lane gathers), then on the grid-search launches of the ring step.
#   gpurun --timeout 600 -- tools/measure_rdreq.sh [views] [points]    -> gpurun_out/rdreq/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rdreq; rm -rf $O; mkdir -p $O
V=${1:-12}; N=${2:-200000}
export MVR_PAIR_GROUPS=1 MVR_PROBE_PROF=0
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/fetch$i -- $R/build/exp_fetch > $O/fetch$i.log 2>&1 || { tail -3 $O/fetch$i.log; exit 1; }
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/probe$i -- python3 $R/tools/step_probe.py $V $N 6 25 > $O/probe$i.log 2>&1 || { tail -3 $O/probe$i.log; exit 1; }
done
python3 - $O > $O/summary.txt <<'P'
import csv, glob, sys, collections
root = sys.argv[1]
def collect(prefix, pick):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/" + prefix + "*/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("mvr::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
            if not pick(k): continue
            per[(k, int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, d), cs in sorted(per.items()):
            for c, v in cs.items(): acc[k][c].append(v)
    return acc
print("== tools/exp_fetch.hip: requests by size against known bytes (stream16: 268 435 456 bytes asked in 2 097 152 lines; gathers: 4 194 304 distinct lines)")
for k, cs in collect("fetch", lambda k: k in ("stream16", "gather16", "gather64")).items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    n32, n64, n128, tot = m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0), m.get("TCC_EA0_RDREQ_sum", 0)
    print("%-10s RDREQ %.4g  32B %.4g  64B %.4g  128B %.4g  -> bytes by size %.4g (32 n32 + 64 n64 + 128 n128), FETCH_SIZE-style 64 x RDREQ = %.4g" % (k, tot, n32, n64, n128, 32 * n32 + 64 * n64 + 128 * n128, 64 * tot))
print("== ring step, grid-search launches (forward = even, reverse = odd dispatches of the last 6 steps)")
for k, cs in collect("probe", lambda k: k.startswith("nn_grid_kernel")).items():
    for label, sl in (("forward", slice(-12, None, 2)), ("reverse", slice(-11, None, 2))):
        m = {c: (sum(v[sl]) / max(1, len(v[sl]))) for c, v in cs.items()}
        n32, n64, n128, tot = m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0), m.get("TCC_EA0_RDREQ_sum", 0)
        print("%-28s %-8s RDREQ %.4g  32B %.4g  64B %.4g  128B %.4g  -> read bytes by size %.4g, 64 x RDREQ %.4g, 2 x 64 x RDREQ %.4g" %
              (k, label, tot, n32, n64, n128, 32 * n32 + 64 * n64 + 128 * n128, 64 * tot, 128 * tot))
P
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
