#!/bin/bash
# run tools/nn_probe.py against each build/libmvr_hip_<name>.so given (GPU box only: overwrites the package library)
# usage: tools/variant_probe.sh "<probe args>" name...
args=$1; shift
for v in "$@"; do
  cp build/libmvr_hip_$v.so multi-view-registration_amd/libmvr_hip.so || exit 1
  echo "== $v $args"
  MVR_STAMP_DUMP=1 timeout -k 10 120 python3 tools/nn_probe.py $args 2>&1 | grep -v amdgpu.ids || exit 1
done
