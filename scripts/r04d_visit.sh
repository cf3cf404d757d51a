#!/bin/bash
# GPU visit: development build against the committed library on one box (bench lines), optional trace
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04d}; mkdir -p $OUT
DEV=${DEV:-monolith_amd/libmhte_dev.so}
show() { python - "$1" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1], "us/step %.2f" % (d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), {k: v_.get("avg_us") for k, v_ in d.get("stages", {}).items() if "step" in k}, (d.get("parity_check") or {}).get("rows_bit_exact"), (d.get("parity_check") or {}).get("n"), (d.get("parity_check") or {}).get("max_abs"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline > $OUT/base.$i.json 2> $OUT/base.$i.err; show $OUT/base.$i.json
  MHTE_LIBRARY=$DEV timeout 300 python bench.py --no-cpu-baseline > $OUT/dev.$i.json 2> $OUT/dev.$i.err; show $OUT/dev.$i.json; tail -2 $OUT/dev.$i.err
done
if [ "${TRACE:-1}" = "1" ]; then
MHTE_LIBRARY=$DEV timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-check --trace-out $OUT/trace.npz > $OUT/trace_bench.json 2> $OUT/trace.err
python scripts/trace_report.py $OUT/trace.npz > $OUT/trace_report.md 2>> $OUT/trace.err; grep -A4 "| 3 |" $OUT/trace_report.md | cut -c1-420
fi
