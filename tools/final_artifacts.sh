#!/bin/bash
# One GPU-box call that regenerates the evidence kept under profiles/ (see profiles/README.md); output in gpurun_out/final/.
#   gpurun --timeout 1100 -- tools/final_artifacts.sh
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
echo bench done
timeout -k 10 600 python bench.py --views 36 --points 1000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n1_stress_36x1M.json 2> $O/stress.err || { tail -5 $O/stress.err; exit 1; }
echo stress done
cd /tmp && export TMPDIR=/tmp
# kernel stats of the bench: default (two groups of pairs) and with one group (one launch = all 12 pairs, the roofline's unit)
for g in 2 1; do
  export MVR_PAIR_GROUPS=$g
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_g$g -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-bruteforce-pass --repeats 0 > $O/bench_under_rocprof_groups$g.json 2> $O/rocprof_g$g.err || { tail -5 $O/rocprof_g$g.err; exit 1; }
  cp $(find $O/stats_g$g -name "*kernel_stats.csv" | head -1) $O/bench_n1_kernel_stats_groups$g.csv
done
unset MVR_PAIR_GROUPS
echo stats done
# HBM-side traffic of the fused search launch (FETCH_SIZE / WRITE_SIZE in separate passes) and the SQ counters of the same launches
$R/tools/measure_traffic.sh final_traffic || exit 1
export MVR_PAIR_GROUPS=1
P=$O/pmc; mkdir -p $P
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $P/pass1 -- python3 $R/tools/step_probe.py 12 200000 6 2 > $P/pass1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $P/pass2 -- python3 $R/tools/step_probe.py 12 200000 6 2 > $P/pass2.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $P nn_cull > $P/summary.txt
echo pmc done
