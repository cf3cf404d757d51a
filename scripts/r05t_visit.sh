#!/bin/bash
# round 5, visit t: EIGHT processes sharing the GPU (the N = 8 code path: windows, sync points, one owner launch
# over eight senders' blocks) on a small resident set; time-sliced: not a scaling number
set -u
OUT=gpurun_out/r05t
mkdir -p $OUT
export TMPDIR=/tmp
for n in 8; do
  timeout 400 python bench.py --no-cpu-baseline --gpus $n --steps 50 --warmup 10 --resident-rows 4194304 > $OUT/ranks$n.json 2> $OUT/ranks$n.err; echo "ranks$n rc=$?"
  python - $OUT/ranks$n.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["n_gpus"], d["ms_per_step"], d["value"], d["config"]["shard_step"], d.get("parity_check"))
PY
  tail -5 $OUT/ranks$n.err
done
