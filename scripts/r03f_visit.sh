#!/bin/bash
# GPU visit r03f: write-through (sc1) output-row stores in the forward kernels, same-box A/B
export TMPDIR=/tmp
OUT=gpurun_out/r03f; mkdir -p $OUT
ab() { # name lib args...
  local name=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export MHTE_LIBRARY=$lib; else unset MHTE_LIBRARY; fi
  timeout 300 python bench.py --no-cpu-baseline --no-parity-check "$@" >> $OUT/ab_$name.jsonl 2>> $OUT/ab_$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/ab_$name.jsonl").read().strip().splitlines()[-1])
print("$name", "us/step %.2f" % (d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), {k: v_["avg_us"] for k, v_ in d.get("stages", {}).items() if "step" in k})
PY
  unset MHTE_LIBRARY
}
for i in 1 2; do
  ab default_new ""
  ab default_wt monolith_amd/libmhte_wt.so
done
for i in 1 2; do
  ab dlrm26_new "" --config dlrm26
  ab dlrm26_wt monolith_amd/libmhte_wt.so --config dlrm26
done
