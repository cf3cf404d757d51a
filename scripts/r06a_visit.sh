#!/bin/bash
# round 6, visit a: the GPU suite on the round's first library (own grouping instead of rocPRIM, the exact-order
# pre-sum, save_as_tensor's side slot, FilterBudget's 16-byte fetch), the pooling rows, the driver's line with
# the host profile of the eager C loop, and the multi-table step with ONE table (what a unique-lookup forward
# would cost on the single-table shape).
set -u
OUT=gpurun_out/r06a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -4 $OUT/pytest_gpu.log
# pooling rows: own sort | dedup form | atomics
timeout 300 python scripts/next_rows_bench.py gather reduce > $OUT/pool_sorted.jsonl 2> $OUT/pool.err; echo "pool rc=$?"
cut -c1-260 $OUT/pool_sorted.jsonl
MHTE_GROUP_DD=1 timeout 300 python scripts/next_rows_bench.py gather reduce > $OUT/pool_dd.jsonl 2>> $OUT/pool.err
cut -c1-260 $OUT/pool_dd.jsonl
rm -rf /tmp/pprof && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/pprof -o trace -- python scripts/next_rows_bench.py gather reduce > $OUT/prof_run.jsonl 2> $OUT/prof.err
db=$(find /tmp/pprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_pooling.md | head -24 | cut -c1-150; fi
# the driver's line (exact_order and eager_cpp windows included), with the host profile of the C loop
MHTE_HOST_PROF=1 MHTE_HOST_PROF_OUT=$OUT/host_prof.md timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
echo "drv rc=$?"; python - <<EOF
import json
d = json.load(open("$OUT/bench_driver_args.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "timing_ms_per_step", "roofline", "parity_check")})
EOF
cat $OUT/host_prof.md
# the multi-table step with one dim-64 table: mstep_fwd_dedup = unique lookup + scatter | dedup
rm -rf /tmp/m1 && timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/m1 -o trace -- \
  python bench.py --config dlrm26 --tables 1 --dims 64 --steps 100 --warmup 10 --no-cpu-baseline --no-parity-check > $OUT/mstep_t1_bench.json 2> $OUT/mstep_t1.err
echo "mstep t1 rc=$?"; cut -c1-300 $OUT/mstep_t1_bench.json
db=$(find /tmp/m1 -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_mstep_t1.md | head -12 | cut -c1-150; fi
# hip-trace of the eager C loop: what the HIP runtime itself takes per launch call
rm -rf /tmp/ht && timeout -k 5 600 rocprofv3 --hip-trace --stats -d /tmp/ht -o trace -- \
  python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity-check --launch eager > $OUT/hiptrace_bench.json 2> $OUT/hiptrace.err
echo "hiptrace rc=$?"
for f in $(find /tmp/ht -name '*hip_api_stats*.csv' | head -1); do head -14 $f | cut -c1-160; cp $f $OUT/hip_api_stats.csv; done
