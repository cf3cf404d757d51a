#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel trace of a short bench.
# Usage (from the repo root on the GPU box): bash scripts/gpu_round.sh <tag> [what...]
#   what: tests smoke bench drv prof tl trace pmc calib next cfg1 dlrm dlrmsps1 dlrmdense dlrmprof dlrmpmc shard shardprof ranks2
#   (default: tests bench prof)
set -u
TAG=${1:-r01}; shift || true
WHAT=${*:-tests bench prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in $WHAT; do
case $w in
tests)
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log ;;
smoke)
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log ;;
bench)
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  cat $OUT/bench.json; tail -5 $OUT/bench.err ;;
prof)
  rm -rf /tmp/prof && timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o trace -- \
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-windows > $OUT/prof_bench.json 2> $OUT/prof.err
  echo "prof rc=$?"; find /tmp/prof -type f | head -20
  for f in $(find /tmp/prof -name '*kernel_stats*.csv'); do cp $f $OUT/; done
  db=$(find /tmp/prof -name '*.db' | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats.md --by-grid --timeline 40 | head -100; fi
  cat $OUT/prof_bench.json ;;
tl)
  # timeline of the overlapped step (graph replay is the last thing the process runs)
  rm -rf /tmp/tl && timeout 600 rocprofv3 --kernel-trace -d /tmp/tl -o trace -- \
    python bench.py --steps 20 --warmup 4 --no-cpu-baseline --resident-rows 16777216 --no-stage-timing > $OUT/tl_bench.json 2> $OUT/tl.err
  db=$(find /tmp/tl -name '*.db' | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/timeline.md --by-grid --timeline 24 | tail -30; fi ;;
trace)
  # per-wavefront timeline of three pipelined steps (mhte_trace_begin) + the bench line of the run
  timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --trace-out $OUT/trace.npz \
    > $OUT/trace_bench.json 2> $OUT/trace.err
  echo "trace rc=$?"; cat $OUT/trace_bench.json; tail -5 $OUT/trace.err
  python scripts/trace_report.py $OUT/trace.npz > $OUT/trace_report.md 2>> $OUT/trace.err; cat $OUT/trace_report.md ;;
pmc)
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pmc_$c && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- \
      python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra-windows --launch eager > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
    echo "pmc $c rc=$?"
    find /tmp/pmc_$c -type f | head
    for f in $(find /tmp/pmc_$c -name '*counter_collection*.csv'); do python scripts/pmc_csv.py $f > $OUT/pmc_${c}.md; head -30 $OUT/pmc_${c}.md; done
  done ;;
calib)
  # FETCH_SIZE / WRITE_SIZE per known byte count on the engine's own access patterns
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/cal_$c && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -o pmc -- \
      python scripts/pmc_calibrate.py > $OUT/calib.json 2> $OUT/calib_$c.err
    echo "calib $c rc=$?"
    for f in $(find /tmp/cal_$c -name '*counter_collection*.csv'); do python scripts/pmc_csv.py $f > $OUT/calib_${c}.md; head -8 $OUT/calib_${c}.md; done
  done ;;
cfg1)
  # BASELINE.json configs[1]: 100 M ids, dim 32, SGD, one table on one GPU
  timeout 600 python bench.py --universe 100000000 --dim 32 --opt sgd > $OUT/bench_configs1.json 2> $OUT/bench_configs1.err
  echo "cfg1 rc=$?"; cut -c1-600 $OUT/bench_configs1.json ;;
dlrm)
  timeout 900 python bench.py --config dlrm26 > $OUT/bench_dlrm26.json 2> $OUT/bench_dlrm26.err
  echo "dlrm rc=$?"; cut -c1-1500 $OUT/bench_dlrm26.json ;;
dlrmsps1)
  # a new update_time second on EVERY step: every touched id's timestamp store is due
  timeout 900 python bench.py --config dlrm26 --steps-per-second 1 --no-cpu-baseline > $OUT/bench_dlrm26_sps1.json 2> $OUT/bench_dlrm26_sps1.err
  echo "dlrmsps1 rc=$?"; cut -c1-300 $OUT/bench_dlrm26_sps1.json ;;
dlrmdense)
  timeout 900 python bench.py --config dlrm26 --dense --no-cpu-baseline > $OUT/bench_dlrm26_dense.json 2> $OUT/bench_dlrm26_dense.err
  echo "dlrmdense rc=$?"; cut -c1-600 $OUT/bench_dlrm26_dense.json; python -c "import json,sys; print(json.load(open('$OUT/bench_dlrm26_dense.json'))['dense'])" ;;
dlrmprof)
  rm -rf /tmp/dprof && timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/dprof -o trace -- \
    python bench.py --config dlrm26 --steps 50 --warmup 10 --no-cpu-baseline --no-parity-check > $OUT/prof_dlrm26_bench.json 2> $OUT/prof_dlrm26.err
  echo "dlrmprof rc=$?"
  db=$(find /tmp/dprof -name '*.db' | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_dlrm26.md | head -14; fi ;;
dlrmpmc)
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/dpmc_$c && timeout -k 5 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/dpmc_$c -o pmc -- \
      python bench.py --config dlrm26 --steps 30 --warmup 10 --no-cpu-baseline --no-parity-check > $OUT/pmc_dlrm_$c.json 2> $OUT/pmc_dlrm_$c.err
    echo "dlrmpmc $c rc=$?"
    for f in $(find /tmp/dpmc_$c -name '*counter_collection*.csv'); do python scripts/pmc_csv.py $f > $OUT/pmc_dlrm_${c}.md; grep mstep $OUT/pmc_dlrm_${c}.md; done
  done ;;
shard)
  timeout -k 5 600 python bench.py --force-sharded --no-cpu-baseline > $OUT/bench_sharded_n1.json 2> $OUT/bench_sharded_n1.err
  echo "shard rc=$?"; cut -c1-400 $OUT/bench_sharded_n1.json
  timeout -k 5 600 python bench.py --config dlrm26 --force-sharded --no-cpu-baseline > $OUT/bench_sharded_n1_dlrm26.json 2> $OUT/bench_sharded_n1_dlrm26.err
  echo "shard dlrm rc=$?"; cut -c1-400 $OUT/bench_sharded_n1_dlrm26.json ;;
shardprof)
  rm -rf /tmp/sprof && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/sprof -o trace -- \
    python bench.py --force-sharded --steps 100 --warmup 10 --no-cpu-baseline --no-parity-check > $OUT/prof_sharded_n1_bench.json 2> $OUT/prof_sharded_n1.err
  echo "shardprof rc=$?"
  db=$(find /tmp/sprof -name '*.db' | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_sharded_n1.md | head -14; fi ;;
drv)
  # the line as the driver asks for it
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
  echo "drv rc=$?"; cut -c1-300 $OUT/bench_driver_args.json ;;
ranks2)
  # N = 2 as the driver launches it; on a 1-GPU box the two processes share the device (peer-store transport)
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 200 --warmup 20 > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err
  echo "ranks2 rc=$?"; cut -c1-300 $OUT/bench_2ranks.json; tail -3 $OUT/bench_2ranks.err ;;
next)
  # the rows next to the hot path: checkpoint, eviction, filter, gather, reductions, optimizers
  NEXT_ROWS_MD=$OUT/next_rows.md timeout 900 python scripts/next_rows_bench.py > $OUT/next_rows.jsonl 2> $OUT/next_rows.err
  echo "next rc=$?"; cat $OUT/next_rows.md; tail -5 $OUT/next_rows.err ;;
esac
done
