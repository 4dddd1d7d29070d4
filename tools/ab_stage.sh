#!/bin/bash
# same-box A/B of the staged walk (grid_stage = 1 / 0): exactness tests first, then the 12 x 200k step both ways (twice round),
# the per-kernel means under rocprofv3 --kernel-trace, and how many waves were staged / fell back and why
#   tools/ab_stage.sh [skiptests] [variant ...]      (variants: tools/build_variant.sh <name> -D...; each is run with grid_stage = 1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_stage; mkdir -p $O; cd $R
SKIP=0; if [ "$1" = "skiptests" ]; then SKIP=1; shift; fi
if [ $SKIP = 0 ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_exact.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
  tail -1 $O/pytest.log
fi
: > $O/ab.log
for v in 1 0 1 0; do
  MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 pipeline=1 grid_stage=$v >> $O/ab.log 2>> $O/ab.err || exit 1
done
for lv in "$@" "$@"; do
  MVR_LIB_VARIANT=$lv MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 pipeline=1 grid_stage=1 >> $O/ab.log 2>> $O/ab.err || exit 1
done
MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 20 25 pipeline=1 grid_stage=1 grid_stage_stat=1 >> $O/ab.log 2>> $O/ab.err || exit 1
MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 10 0 pipeline=1 grid_stage=1 grid_stage_stat=1 >> $O/ab.log 2>> $O/ab.err || exit 1
python3 - $O/ab.log <<'P'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print("lib=%-8s %-40s ms_per_step %.4f  n_corr %d  %s %s" % (r["lib"], " ".join("%s=%s" % kv for kv in r["knobs"].items() if kv[0] != "pipeline"), r["ms_per_step"], r["n_corr"],
          r.get("stage_waves_fwd_rev", ""), ""))
P
cd /tmp
: > $O/kern.txt
for v in 1 0 "$@"; do
  rm -rf $O/t; mkdir -p $O/t
  if [ "$v" = "1" -o "$v" = "0" ]; then export MVR_LIB_VARIANT=; st=$v; else export MVR_LIB_VARIANT=$v; st=1; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/step_probe.py 12 200000 40 20 pipeline=1 grid_stage=$st > $O/t/probe.json 2> $O/t/probe.err || exit 1
  python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" "stage=$v" >> $O/kern.txt <<'P'
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
