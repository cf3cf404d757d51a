#!/bin/bash
# A/B on one box
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
run() { # name env... -- bench args
  local name=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 $EXTRA > $OUT/$name.$i.json 2> $OUT/$name.$i.err
    python - <<PY
import json
try:
  d=json.loads(open("$OUT/$name.$i.json").read().strip().splitlines()[-1])
  st=d.get("stages",{})
  print("$name", $i, "us/step %.2f"%(d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), d.get("graph_error"), {k:v["avg_us"] for k,v in st.items() if k.startswith("step_")})
except Exception as e: print("$name", $i, "FAILED", e)
PY
    tail -2 $OUT/$name.$i.err
  done
}
run base X=1
run side5 MHTE_SIDE_DEDUP=1
run side4 MHTE_SIDE_DEDUP=1 MHTE_BWD_BLOCKS_PER_CU=4
run side3 MHTE_SIDE_DEDUP=1 MHTE_BWD_BLOCKS_PER_CU=3
EXTRA="--launch eager"
run base_eager X=1
run side4_eager MHTE_SIDE_DEDUP=1 MHTE_BWD_BLOCKS_PER_CU=4
EXTRA=""
MHTE_SIDE_DEDUP=1 MHTE_BWD_BLOCKS_PER_CU=4 timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pipelin or step" 2>&1 | tail -3
