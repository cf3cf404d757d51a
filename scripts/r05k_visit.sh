#!/bin/bash
# round 5, visit k: the admission filter inside the step (lane-group consultation, the window's two
# launches only when the head split can be full) and the step_bwd instances without filter code
set -u
OUT=gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "filter or Filter" > $OUT/pytest_filter.log 2>&1; echo "pytest filter rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_filter.log
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $OUT/pytest_parity.log 2>&1; echo "pytest parity rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_parity.log
# the filtered step: this build; the same with the window's launches behind every step (round 4's form)
timeout 300 python scripts/next_rows_bench.py filter > $OUT/filter_step.md 2> $OUT/filter_step.err; echo "filter bench rc=$?"
cat $OUT/filter_step.md
MHTE_FILTER_MAINTAIN_ALWAYS=1 timeout 300 python scripts/next_rows_bench.py filter > $OUT/filter_step_maintain_always.md 2>> $OUT/filter_step.err
cat $OUT/filter_step_maintain_always.md
# the headline with the driver's arguments, twice (box noise)
for i in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_args_$i.json 2> $OUT/bench_$i.err; echo "bench rc=$?"
  python - $OUT/bench_driver_args_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["value"], d["roofline"], d.get("timing_ms_per_step"))
PY
done
timeout 300 python scripts/next_rows_bench.py optimizers > $OUT/step_optimizers.md 2> $OUT/opt.err
cat $OUT/step_optimizers.md
