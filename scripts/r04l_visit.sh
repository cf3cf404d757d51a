#!/bin/bash
# GPU visit: two batches of look-ahead (run dedup in the backward launch) against one, same library
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04l}; mkdir -p $OUT
export MHTE_LIBRARY=${DEV:-monolith_amd/libmhte_dev.so}
show() { python - "$1" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1], "us/step %.2f" % (d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), {k: v_.get("avg_us") for k, v_ in d.get("stages", {}).items() if "step" in k}, (d.get("parity_check") or {}).get("rows_bit_exact"), (d.get("parity_check") or {}).get("n"), (d.get("parity_check") or {}).get("max_abs"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2; do
  for la in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --lookahead $la > $OUT/la$la.$i.json 2> $OUT/la$la.$i.err; show $OUT/la$la.$i.json; tail -2 $OUT/la$la.$i.err | grep -v amdgpu.ids
  done
done
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-check --trace-out $OUT/trace.npz > $OUT/trace_bench.json 2> $OUT/trace.err
python scripts/trace_report.py $OUT/trace.npz > $OUT/trace_report.md 2>> $OUT/trace.err; grep -A9 "| 2 |" $OUT/trace_report.md | cut -c1-330
