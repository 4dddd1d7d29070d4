#!/bin/bash
# One GPU-box call that regenerates the evidence kept under profiles/ (see profiles/README.md); output in gpurun_out/final/.
#   gpurun --timeout 1100 -- tools/final_artifacts.sh part1     (tests, bench lines, rocprof kernel stats)
#   gpurun --timeout 1100 -- tools/final_artifacts.sh part2     (PMC traffic and counters, same-box A/B, step timelines)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; [ "$1" != "part2" ] && rm -rf $O; mkdir -p $O
cd $R
if [ "$1" != "part2" ]; then
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
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_g$g -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-bruteforce-pass --repeats 0 > $O/bench_under_rocprof_groups$g.json 2> $O/rocprof_g$g.err || { tail -5 $O/rocprof_g$g.err; exit 1; }
  cp $(find $O/stats_g$g -name "*kernel_stats.csv" | head -1) $O/bench_n1_kernel_stats_groups$g.csv
done
unset MVR_PAIR_GROUPS
echo stats done
[ "$1" = "part1" ] && exit 0
fi
if [ "$1" != "part1" ]; then
O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# HBM-side traffic of the search launches (FETCH_SIZE / WRITE_SIZE in separate passes): culled kernel (ring_search=0) and grid search
$R/tools/measure_traffic.sh final_traffic || exit 1
export MVR_PAIR_GROUPS=1
P=$O/pmc; mkdir -p $P
# counters of the grid-search launches (every step seeded: 25 warm-up steps), one small set per pass
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/pass$i -- python3 $R/tools/step_probe.py 12 200000 6 25 > $P/pass$i.log 2>&1 || { echo "pmc pass $i failed"; exit 1; }
done
python3 $R/tools/pmc_summary.py $P nn_grid_kernel > $P/summary.txt
python3 $R/tools/pmc_summary.py $P nn_cull_list >> $P/summary.txt
find $P -name "*.csv" -size +2M -delete
echo pmc done
unset MVR_PAIR_GROUPS
# same-box A/B of the search used for bounded queries, and the kernel timeline of one step
for k in "ring_search=0" "ring_search=1" "ring_search=0" "ring_search=1"; do
  MVR_PROBE_PROF=0 timeout -k 10 120 python3 $R/tools/step_probe.py 12 200000 40 25 $k >> $O/ab_ring_search.log 2>&1 || exit 1
done
timeout -k 10 200 $R/tools/trace_probe.sh final_g1 pair_groups=1 > $O/step_timeline_groups1.txt || exit 1
timeout -k 10 200 $R/tools/trace_probe.sh final_g2 > $O/step_timeline_groups2.txt || exit 1
echo ab done
fi
