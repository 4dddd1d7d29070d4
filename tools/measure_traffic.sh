#!/bin/bash
# HBM-side traffic of the fused search launch, per launch, from the PMC counters -- collected as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel trace only)
# and written to profiles/nn_cull_traffic.json together with the hash of the kernel source it was measured on
# (bench.py reports `traffic` only while that hash matches the tree).
#   gpurun --timeout 900 -- tools/measure_traffic.sh [tag]
set -o pipefail
R=$GRAFT_REPO_ROOT; TAG=${1:-traffic}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
[ -n "$2" ] && export MVR_LIB_VARIANT=$2       # optional: a build variant (tools/build_variant.sh)
cd /tmp && export TMPDIR=/tmp
export MVR_PAIR_GROUPS=1          # one launch = the searches of all 12 scan pairs of a step
VV=${MVR_TRAFFIC_VIEWS:-12}; NN=${MVR_TRAFFIC_POINTS:-200000}
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/$pass -- python3 $R/tools/step_probe.py $VV $NN 10 2 ring_search=0 > $O/$pass.log 2>&1 || { tail -5 $O/$pass.log; exit 1; }
done
python3 $R/tools/traffic_json.py $O > $O/nn_cull_traffic.json || exit 1
cat $O/nn_cull_traffic.json
# the grid search's launches (the dominant kernel once the queries have bounds): a longer probe, every step seeded
for pass in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $O/grid
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/grid/$pass -- python3 $R/tools/step_probe.py $VV $NN 10 25 > $O/grid/$pass.log 2>&1 || { tail -5 $O/grid/$pass.log; exit 1; }
done
python3 $R/tools/traffic_json.py $O/grid grid > $O/nn_grid_traffic.json || exit 1
cat $O/nn_grid_traffic.json
