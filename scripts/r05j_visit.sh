#!/bin/bash
# after the fixes (owed pass before credits; the pass rides only with one rank): the process test repeated,
# then the shard suites once
set -u
OUT=gpurun_out/r05j
mkdir -p $OUT
export TMPDIR=/tmp
fails=0
for i in 1 2 3 4 5 6 7 8; do
  timeout 200 python -m pytest "tests/test_shard_ipc_gpu.py::test_processes_against_oracle" -m gpu -x -q > $OUT/procs_$i.log 2>&1 || { fails=$((fails+1)); grep -n "Mismatched\|step [0-9]" $OUT/procs_$i.log | head -3; }
done
echo "process test: $fails failures of 8" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_shard_ipc_gpu.py tests/test_shard_step_gpu.py -m gpu -x -q > $OUT/pytest_shard.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_shard.log
timeout 600 python bench.py --no-cpu-baseline --gpus 2 --steps 100 --warmup 10 > $OUT/ranks2.json 2> $OUT/ranks2.err; echo "ranks2 rc=$?"
python - $OUT/ranks2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["value"], d["config"]["shard_step"]["launches_per_step"], d.get("parity_check"))
PY
