#!/bin/bash
# tools/nn_sweep.sh -- A/B the NN launch knobs on one box (interleaved rounds, one process each)
N=${1:-200000}
for round in 1 2; do
for cfg in "nn_q=4 nn_sub=32 nn_blocks_per_cu=4" "nn_q=4 nn_sub=32 nn_blocks_per_cu=2" "nn_q=4 nn_sub=32 nn_blocks_per_cu=3" "nn_q=4 nn_sub=32 nn_blocks_per_cu=5" \
           "nn_q=4 nn_sub=64 nn_blocks_per_cu=4" "nn_q=4 nn_sub=16 nn_blocks_per_cu=4" "nn_q=2 nn_sub=32 nn_blocks_per_cu=5" \
           "nn_q=6 nn_sub=32 nn_blocks_per_cu=4" "nn_q=6 nn_sub=32 nn_blocks_per_cu=3" "nn_q=8 nn_sub=32 nn_blocks_per_cu=4" "nn_q=8 nn_sub=32 nn_blocks_per_cu=2" "nn_q=8 nn_sub=16 nn_blocks_per_cu=3"; do
  python3 tools/nn_probe.py $N 4 0 $cfg
done
done
