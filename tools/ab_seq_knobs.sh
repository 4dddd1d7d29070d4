#!/bin/bash
# same-box A/B of a knob on the sequential mode (five sweeps): tools/ab_seq_knobs.sh "lazy_super=1" "lazy_super=0" ...
R=$GRAFT_REPO_ROOT; cd $R
for r in 1 2; do for k in "$@"; do
  MVR_SEQ_KNOBS="$k" timeout -k 10 200 python3 tools/seq_bench.py --no-cpu --no-brute --repeat 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())['gpu_culled']; print('%-24s wall %.4f ms/align  native by sweep %s' % (sys.argv[1], j['ms_per_align'], j['native_ms_per_align_by_sweep']))" "$k"
done; done
