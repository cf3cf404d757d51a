#!/bin/bash
# round 6, visit c: persistent dedup workgroups fetch their next block's ids a block ahead (multi-table and
# sharded steps); the gather gradient stores into the freshly zeroed buffer without reading it.  The suite, the
# 26-table line, the sharded lines with one rank, the pooling rows.
set -u
OUT=gpurun_out/r06c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python scripts/next_rows_bench.py gather reduce > $OUT/pool_sorted.jsonl 2> $OUT/pool.err; echo "pool rc=$?"
cut -c1-200 $OUT/pool_sorted.jsonl | grep -i "gradient\|unsorted"
timeout 900 python bench.py --config dlrm26 --no-cpu-baseline > $OUT/bench_dlrm26.json 2> $OUT/bench_dlrm26.err
echo "dlrm rc=$?"; python - <<EOF
import json
d = json.load(open("$OUT/bench_dlrm26.json"))
print("dlrm26", d.get("ms_per_step"), d.get("value"), d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("avg_launch_us"), d.get("stages"))
EOF
rm -rf /tmp/dprof && timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/dprof -o trace -- \
  python bench.py --config dlrm26 --steps 50 --warmup 10 --no-cpu-baseline --no-parity-check > $OUT/prof_dlrm26_bench.json 2> $OUT/prof_dlrm26.err
db=$(find /tmp/dprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_dlrm26.md | head -9 | cut -c1-150; fi
timeout -k 5 600 python bench.py --force-sharded --no-cpu-baseline > $OUT/bench_sharded_n1.json 2> $OUT/bench_sharded_n1.err
echo "shard rc=$?"; python -c "import json; d=json.load(open('$OUT/bench_sharded_n1.json')); print('sharded n1', d['ms_per_step'], d['config'].get('shard_step'))"
timeout -k 5 600 python bench.py --config dlrm26 --force-sharded --no-cpu-baseline > $OUT/bench_sharded_n1_dlrm26.json 2> $OUT/bench_sharded_n1_dlrm26.err
echo "shard dlrm rc=$?"; python -c "import json; d=json.load(open('$OUT/bench_sharded_n1_dlrm26.json')); print('sharded n1 dlrm26', d['ms_per_step'])"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
python -c "import json; d=json.load(open('$OUT/bench_driver_args.json')); print('drv', d['ms_per_step'], d['timing_ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
