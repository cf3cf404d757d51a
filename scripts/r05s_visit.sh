#!/bin/bash
# round 5, visit s: three and four PROCESSES sharing the GPU (launch count at N = 3, 4; time-sliced: not a scaling number)
set -u
OUT=gpurun_out/r05s
mkdir -p $OUT
export TMPDIR=/tmp
for n in 3 4; do
  timeout 600 python bench.py --no-cpu-baseline --gpus $n --steps 100 --warmup 10 > $OUT/ranks$n.json 2> $OUT/ranks$n.err; echo "ranks$n rc=$?"
  python - $OUT/ranks$n.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["n_gpus"], d["ms_per_step"], d["value"], d["config"]["shard_step"]["launches_per_step"], d["config"]["shard_step"]["transport"], d.get("parity_check"))
PY
done
