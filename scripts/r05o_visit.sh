#!/bin/bash
# round 5, visit o: mstep_bwd instances without the admission filter's code (38 -> 4 spilled VGPRs)
set -u
OUT=gpurun_out/r05o
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multi_step_gpu.py -m gpu -x -q > $OUT/pytest_mstep.log 2>&1; echo "pytest mstep rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_mstep.log
for i in 1 2; do
  timeout 600 python bench.py --config dlrm26 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_dlrm26_$i.json 2> $OUT/dlrm26_$i.err; echo "dlrm26 rc=$?"
  python - $OUT/bench_dlrm26_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d.get("stages"))
PY
done
timeout 600 python bench.py --config dlrm26 --force-sharded --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_dlrm26_sharded.json 2> $OUT/dlrm26_s.err; echo "dlrm26 sharded rc=$?"
python - $OUT/bench_dlrm26_sharded.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["value"], d.get("stages"))
PY
