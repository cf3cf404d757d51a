#!/bin/bash
# Same-box A/B: the previous build of the library (monolith_amd/libmhte_prev.so, loaded through
# MHTE_LIBRARY) against the current one, interleaved, two bench runs each way.
# Usage: bash scripts/ab_visit.sh <tag>
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
run() { # name env...
  local name=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $OUT/$name.$i.json 2> $OUT/$name.$i.err
    python - <<PY
import json
try:
  d=json.loads(open("$OUT/$name.$i.json").read().strip().splitlines()[-1])
  st=d.get("stages",{})
  print("$name", $i, "us/step %.2f"%(d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), {k:v["avg_us"] for k,v in st.items() if k.startswith("step_")})
except Exception as e: print("$name", $i, "FAILED", e)
PY
  done
}
run prev MHTE_LIBRARY=monolith_amd/libmhte_prev.so
run new X=1
run prev_b MHTE_LIBRARY=monolith_amd/libmhte_prev.so
run new_b X=1
