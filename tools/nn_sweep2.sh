#!/bin/bash
N=${1:-200000}
for round in 1 2; do
for cfg in "nn_q=8 nn_sub=32 nn_blocks_per_cu=2" "nn_q=8 nn_sub=32 nn_blocks_per_cu=1" "nn_q=6 nn_sub=32 nn_blocks_per_cu=2" "nn_q=4 nn_sub=32 nn_blocks_per_cu=1" "nn_q=4 nn_sub=32 nn_blocks_per_cu=2" "nn_q=8 nn_sub=16 nn_blocks_per_cu=2"; do
  python3 tools/nn_probe.py $N 4 0 $cfg
done
done
for cfg in "nn_q=8 nn_sub=32 nn_blocks_per_cu=2" "nn_q=4 nn_sub=32 nn_blocks_per_cu=2"; do python3 tools/nn_probe.py $N 4 1 $cfg; done
