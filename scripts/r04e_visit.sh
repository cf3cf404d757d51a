#!/bin/bash
# GPU visit: full test suite of the new library; warm and cold-table A/B against the base library
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04e}; mkdir -p $OUT
BASE=monolith_amd/libmhte_base.so
show() { python - "$1" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1], "us/step %.2f" % (d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), {k: v_.get("avg_us") for k, v_ in d.get("stages", {}).items() if "step" in k}, (d.get("parity_check") or {}).get("rows_bit_exact"), (d.get("parity_check") or {}).get("n"), (d.get("parity_check") or {}).get("max_abs"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
for i in 1 2; do
  MHTE_LIBRARY=$BASE timeout 300 python bench.py --no-cpu-baseline > $OUT/base.$i.json 2> $OUT/base.$i.err; show $OUT/base.$i.json
  timeout 300 python bench.py --no-cpu-baseline > $OUT/new.$i.json 2> $OUT/new.$i.err; show $OUT/new.$i.json
done
# cold table: nothing resident, every id of the first steps is new
MHTE_LIBRARY=$BASE timeout 300 python bench.py --no-cpu-baseline --resident-rows 0 > $OUT/base_cold.json 2> $OUT/base_cold.err; show $OUT/base_cold.json
timeout 300 python bench.py --no-cpu-baseline --resident-rows 0 > $OUT/new_cold.json 2> $OUT/new_cold.err; show $OUT/new_cold.json
