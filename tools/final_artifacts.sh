#!/bin/bash
# GPU-box calls that regenerate the evidence kept under profiles/ (see profiles/README.md); output in gpurun_out/final/.
#   gpurun --timeout 1100 -- tools/final_artifacts.sh part1     (tests, bench lines, rocprof kernel stats 12x200k)
#   gpurun --timeout 1100 -- tools/final_artifacts.sh part2     (PMC traffic + counters 12x200k, FETCH calibration, A/Bs, timelines, cold start)
#   gpurun --timeout 1100 -- tools/final_artifacts.sh part3     (config 5's shape, 36 x 1M: bench line, kernel stats, PMC traffic)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
case "$1" in
part1)
  timeout -k 10 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
  tail -1 $O/pytest_gpu.log
  timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
  echo bench done
  MVR_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 \
      --no-cpu-baseline --no-bruteforce-pass > $O/bench_torchrun_1rank_rccl.json 2> $O/torchrun.err || { tail -5 $O/torchrun.err; exit 1; }
  echo torchrun done
  ;&
stats)
  cd /tmp && export TMPDIR=/tmp
  # kernel stats of the bench with the pairs in one group (one launch = all 12 pairs, the roofline's unit) and with the default.
  # MVR_BENCH_PREWARM=0: without the 20 set-up passes every launch of the run belongs to a window that starts from the
  # prior, like the windows the bench's own HIP events cover, and --no-secondary keeps the one-pair launches of the sequential
  # mode (the same kernel on a 200k-query launch) out of the statistics -- the two averages are then over the same kind of launch
  export MVR_BENCH_PREWARM=0
  for g in 1 2; do
    export MVR_PAIR_GROUPS=$g
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_g$g -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-bruteforce-pass --no-secondary --repeats 0 > $O/bench_under_rocprof_groups$g.json 2> $O/rocprof_g$g.err || { tail -5 $O/rocprof_g$g.err; exit 1; }
    cp $(find $O/stats_g$g -name "*kernel_stats.csv" | head -1) $O/bench_n1_kernel_stats_groups$g.csv
  done
  unset MVR_PAIR_GROUPS
  echo stats done
  ;;
part2)
  cd /tmp && export TMPDIR=/tmp
  # HBM-side traffic of the search launches (FETCH_SIZE / WRITE_SIZE in separate passes): culled kernel (ring_search=0) and grid search
  $R/tools/measure_traffic.sh final_traffic || exit 1
  # what FETCH_SIZE counts for gathers (the walk's dominant access), against a known byte count
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/exp_fetch -- $R/build/exp_fetch > $O/exp_fetch.txt 2>&1 || { tail -5 $O/exp_fetch.txt; exit 1; }
  python3 - $O/exp_fetch >> $O/exp_fetch.txt <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": print("FETCH_SIZE  %-10s %.0f KiB" % (r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
P
  cat $O/exp_fetch.txt
  export MVR_PAIR_GROUPS=1
  P=$O/pmc; mkdir -p $P
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
             "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
             "TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum" \
             "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/pass$i -- python3 $R/tools/step_probe.py 12 200000 6 25 > $P/pass$i.log 2>&1 || { echo "pmc pass $i failed"; exit 1; }
  done
  python3 $R/tools/pmc_summary.py $P nn_grid_kernel > $P/summary.txt
  find $P -name "*.csv" -size +2M -delete
  echo pmc done
  unset MVR_PAIR_GROUPS
  cd $R
  # same-box A/Bs: the pipelined pass loop, the search used for bounded queries, the sequential mode's reverse search
  for k in "pipeline=1" "pipeline=0" "pipeline=1" "pipeline=0" "pipeline=1 ring_search=0" "pipeline=0 ring_search=0"; do
    MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 40 25 $k >> $O/ab_pipeline_ring_search.log 2>> $O/ab.err || exit 1
  done
  for m in 1 0 2 1 0 2; do MVR_SEQ_SEED=0 MVR_SEQ_SEARCH=$m timeout -k 10 200 python3 tools/seq_bench.py --no-cpu --no-brute --repeat 2 2>> $O/ab.err | sed "s/^/seq_search=$m /" >> $O/ab_seq_search.log; done
  # the sequential mode as the reference runs it (repeat_times = 5): ms per align by sweep, with and without the seeds an align leaves
  for m in 1 0 1 0; do MVR_SEQ_SEED=$m timeout -k 10 200 python3 tools/seq_bench.py --no-cpu --no-brute --repeat 5 2>> $O/ab.err | sed "s/^/seq_seed=$m /" >> $O/ab_seq_seed.log; done
  # one bench-like window pass by pass: 20 passes restarted from the prior in a warm context
  MVR_PROBE_PASSLOG=1 MVR_PROBE_PROF=0 timeout -k 10 120 python3 tools/step_probe.py 12 200000 20 25 > $O/window_pass_log.json 2>> $O/ab.err
  # timeline of the steady passes WITHOUT the per-launch instrumentation of the probe (no fill / copy operations between the kernels)
  MVR_PROBE_PROF=0 timeout -k 10 200 tools/timeline.sh final_noprof pipeline=1 > $O/step_timeline_pipelined_plain.txt || exit 1
  # a registration from a standing start, in a warm process (second context): per-pass wall times, with the host stopwatch
  MVR_TRACE_HOST=1 timeout -k 10 200 python3 tools/cold_probe.py 12 200000 12 one_call=1 reps=2 > $O/cold_registration.jsonl 2> $O/cold_host_trace.txt || exit 1
  timeout -k 10 200 python3 tools/cold_probe.py 12 200000 12 one_call=1 reps=2 pipeline=0 >> $O/cold_registration.jsonl 2>> $O/ab.err
  # kernel timelines of one pass with the gaps: pipelined and not
  timeout -k 10 200 tools/timeline.sh final_p1 pipeline=1 > $O/step_timeline_pipelined.txt || exit 1
  timeout -k 10 200 tools/timeline.sh final_p0 pipeline=0 > $O/step_timeline_unpipelined.txt || exit 1
  build/exp_gate 12 30 75 300 > $O/exp_gate.txt 2>&1
  echo ab done
  ;;
part3)
  # config 5's shape on one GPU: the bench line with >= 10 timed steps, the kernel statistics and the counters at ITS size
  timeout -k 10 900 python bench.py --views 36 --points 1000000 --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline > $O/bench_n1_stress_36x1M.json 2> $O/stress.err || { tail -5 $O/stress.err; exit 1; }
  echo stress done
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_36x1M -- python3 $R/tools/step_probe.py 36 1000000 12 6 > $O/step_probe_36x1M_under_rocprof.json 2> $O/rocprof_36.err || { tail -5 $O/rocprof_36.err; exit 1; }
  cp $(find $O/stats_36x1M -name "*kernel_stats.csv" | head -1) $O/step_probe_36x1M_kernel_stats.csv
  find $O/stats_36x1M -name "*.csv" -size +2M -delete
  echo stats done
  MVR_TRAFFIC_VIEWS=36 MVR_TRAFFIC_POINTS=1000000 $R/tools/measure_traffic.sh final_traffic_36x1M || exit 1
  ;;
*) echo "usage: $0 part1|stats|part2|part3"; exit 2 ;;
esac
