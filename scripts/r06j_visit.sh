#!/bin/bash
# round 6, visit j: the library rebuilt from the committed tree after the hot list left it — the GPU suite (with the
# new test of the grouping's many-key and four-pass shapes), smoke, the driver's line, and a rocprofv3 summary of the
# bit-exact mode (rd_exact_sum_kernel in front of step_bwd).
set -u
OUT=gpurun_out/r06j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -2; grep -n "Error\|assert" $OUT/pytest_gpu.log | head -10
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
python -c "import json; d=json.load(open('$OUT/bench_driver_args.json')); print('drv', d['ms_per_step'], d['value'], d['timing_ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['cpu_baseline'].get('value'))"
rm -rf /tmp/xprof && timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/xprof -o trace -- \
  python bench.py --exact-order --steps 60 --warmup 10 --no-cpu-baseline --no-extra-windows --launch eager > $OUT/bench_exact_order_under_rocprof.json 2> $OUT/xprof.err
echo "xprof rc=$?"
db=$(find /tmp/xprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_exact_order.md --by-grid | grep "rd_exact\|step_bwd\|step_fwd" | cut -c1-140; fi
python -c "import json; d=json.load(open('$OUT/bench_exact_order_under_rocprof.json')); print('exact', d['ms_per_step'], d['parity_check'])" | cut -c1-400
