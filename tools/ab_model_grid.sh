#!/bin/bash
# the sequential sweep (tools/seq_bench.py, three sweeps) with the forward search through ONE grid over the model (seq_search 3) under
# the knob sets given -- "ref" = the default (seq_search 1: culled kernel) --: ms per align by sweep WITHOUT the profiler (these
# include seq_search 3's from-scratch grid build per align: the prototype has no incremental update, so only the KERNEL columns are
# what a free grid would cost), then the search launches' mean durations from a kernel trace: over all aligns / over the last sweep
#   tools/ab_model_grid.sh ref "seq_cell_points=8 grid_light_rows=32 grid_light_rows_lone=32" ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() {   # $1 = knobs
  local mode=3 knobs="$1"
  if [ "$1" = ref ]; then mode=1; knobs=""; fi
  O=$R/gpurun_out/sweep_tmp; rm -rf $O; mkdir -p $O
  MVR_SEQ_SEARCH=$mode MVR_SEQ_KNOBS="$knobs" python3 $R/tools/seq_bench.py --no-cpu --no-brute --repeat 3 > $O/plain.json 2> $O/plain.err || { tail -n 5 $O/plain.err; return; }
  MVR_SEQ_SEARCH=$mode MVR_SEQ_KNOBS="$knobs" rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/seq_bench.py --no-cpu --no-brute --repeat 3 > $O/seq.json 2> $O/seq.err || { tail -n 5 $O/seq.err; return; }
  python3 - "$O/plain.json" "$(find $O -name '*kernel_trace.csv' | head -1)" "$1" <<'P'
import csv, sys, json, collections
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])['gpu_culled']
rows=list(csv.DictReader(open(sys.argv[2])))
agg=collections.OrderedDict()
for r in rows:
    k=r['Kernel_Name'].replace('mvr::(anonymous namespace)::','').replace('mvr::','').replace('void ','').split('(')[0]
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last={}
for r in rows:
    k=r['Kernel_Name'].replace('mvr::(anonymous namespace)::','').replace('mvr::','').replace('void ','').split('(')[0]
    last.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
pick=lambda s: ' '.join('%s=%.0f/%.0f' % (k.split('<')[0][:14]+('S' if 'true>' in k else ''), a[1]/a[0], sum(last[k][-11:])/11) for k,a in agg.items() if any(x in k for x in s))
print("(mean over all aligns / over the last sweep's 11)")
print("%-60s wall %.3f by sweep %s | %s" % (sys.argv[3], d['ms_per_align'], d['native_ms_per_align_by_sweep'], pick(['nn_grid_kernel','nn_cull_list','nn_grid_wide','nn_grid_tail','nn_cull_kernel'])))
P
}
for k in "$@"; do run "$k"; done
