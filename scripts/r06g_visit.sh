#!/bin/bash
# round 6, visit g: the whole GPU suite three times on the final library (flakiness check), smoke, the driver's line
set -u
OUT=gpurun_out/r06g
mkdir -p $OUT
export TMPDIR=/tmp
fails=0
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_$i.log 2>&1 || { fails=$((fails+1)); tail -15 $OUT/pytest_$i.log; }
  grep -n "passed\|failed" $OUT/pytest_$i.log | tail -1
done
echo "pytest -m gpu: $fails failures of 3 runs" | tee $OUT/summary.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
python -c "import json; d=json.load(open('$OUT/bench_driver_args.json')); print('drv', d['ms_per_step'], d['timing_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'], d['cpu_baseline']['value'], d['cpu_baseline'].get('single_threaded_ms'))"
