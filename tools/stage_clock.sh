#!/bin/bash
# where a wave of the staged walk spends its cycles (diagnostics build: tools/build_variant.sh clock -DMVR_STAGE_CLOCK)
#   tools/stage_clock.sh [knob=value ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stage_clock; mkdir -p $O; cd $R
MVR_LIB_VARIANT=clock MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 20 25 pipeline=1 grid_stage_stat=1 "$@" > $O/clock.json 2> $O/clock.err || { tail $O/clock.err; exit 1; }
python3 - $O/clock.json <<'P'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", r["ms_per_step"], "waves", r["stage_waves"])
names = ["prologue", "box+rows+table", "point staging", "walk", "epilogue", "-", "-", "waves"]
for o, v in r["stage_clock"].items():
    w = max(v[7], 1.0)
    print("%-9s waves %9d  cycles per wave: " % (o, v[7]) + ", ".join("%s %.0f" % (names[k], v[k] / w) for k in range(5)) + "  | total %.0f" % (sum(v[:5]) / w))
P
