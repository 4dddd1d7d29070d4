#!/bin/bash
# tools/cull_sweep.sh -- A/B the culled kernel's queries-per-lane on one box
N=${1:-200000}
for round in 1 2; do
for cfg in "cull_q=1" "cull_q=2" "cull_q=4"; do
  python3 tools/nn_probe.py $N 6 0 $cfg
done
done
