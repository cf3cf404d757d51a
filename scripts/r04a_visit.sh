#!/bin/bash
# GPU visit r04a: same-box reference numbers at the start of round 4 — the single-table step against
# the multi-table step's kernels driven with 1 / 2 / 4 dim-64 tables (unique lookup + scatter + hints)
export TMPDIR=/tmp
OUT=gpurun_out/r04a; mkdir -p $OUT
show() { python - "$1" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1], "us/step %.2f" % (d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), {k: v_.get("avg_us") for k, v_ in d.get("stages", {}).items() if "step" in k})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
timeout 300 python bench.py --no-cpu-baseline --no-parity-check > $OUT/default.json 2> $OUT/default.err; show $OUT/default.json
for T in 1 2 4; do
  timeout 300 python bench.py --config dlrm26 --tables $T --dims 64 --resident-rows $((134217728)) --universe $((1000000000*T)) --no-cpu-baseline --no-parity-check --evict-every 0 > $OUT/mstep_t$T.json 2> $OUT/mstep_t$T.err; show $OUT/mstep_t$T.json
done
timeout 300 python bench.py --no-cpu-baseline --no-parity-check --steps 20 --warmup 5 > $OUT/driver_args.json 2> $OUT/driver_args.err; show $OUT/driver_args.json
