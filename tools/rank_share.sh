#!/bin/bash
# the projection of one rank's share (tools/rank_share_bench.py) at 12 x 200k and 36 x 1M -> gpurun_out/rank_share/*.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rank_share; mkdir -p $O; cd $R
timeout -k 10 500 python3 -X faulthandler tools/rank_share_bench.py 12 200000 20 1,2,4,8 "$@" > $O/12x200k.json 2> $O/err_12.txt; echo "12x200k rc=$?"
if [ "$MVR_RANK_SHARE_BIG" != "0" ]; then
  timeout -k 10 600 python3 -X faulthandler tools/rank_share_bench.py 36 1000000 6 1,2,4,8 "$@" > $O/36x1M.json 2> $O/err_36.txt; echo "36x1M rc=$?"
fi
python3 - $O <<'P'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*x*.json")):
    try: r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e: print(f, "unreadable:", e); continue
    print(f.split("/")[-1], "rccl", r["rccl"])
    for w, v in r["projected"].items():
        print("  world %s: %.4f ms/step unpipelined, %.4f pipelined; ranks (unpipelined) %s; rank 0 {enqueue, wait, solve} %s" % (
            w, v["ms_per_step_pipeline0"], v["ms_per_step_pipeline1"], [round(x["ms_per_step_pipeline0"], 3) for x in v["ranks"]],
            [round(t, 4) for t in v["ranks"][0]["timing_ms_pipeline0"]]))
P
tail -5 $O/err_12.txt | grep -v amdgpu.ids
