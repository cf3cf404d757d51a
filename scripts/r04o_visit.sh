#!/bin/bash
# GPU visit: the hot-path register spill of the id-major trip removed (dev build, dims 64 only) against
# the library on disk: default bench and the multi-table step on dim-64 tables
export TMPDIR=/tmp MHTE_NO_REBUILD=1
OUT=gpurun_out/${1:-r04o}; mkdir -p $OUT
DEV=monolith_amd/libmhte_dev.so
show() { python - "$1" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1], "us/step %.2f" % (d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), {k: v_.get("avg_us") for k, v_ in d.get("stages", {}).items() if "step" in k})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-parity-check > $OUT/base.$i.json 2> $OUT/base.$i.err; show $OUT/base.$i.json
  MHTE_LIBRARY=$DEV timeout 300 python bench.py --no-cpu-baseline --no-parity-check > $OUT/dev.$i.json 2> $OUT/dev.$i.err; show $OUT/dev.$i.json
done
for i in 1 2; do
  timeout 300 python bench.py --config dlrm26 --dims 64 --tables 8 --no-cpu-baseline --no-parity-check > $OUT/base_m.$i.json 2> $OUT/base_m.$i.err; show $OUT/base_m.$i.json
  MHTE_LIBRARY=$DEV timeout 300 python bench.py --config dlrm26 --dims 64 --tables 8 --no-cpu-baseline --no-parity-check > $OUT/dev_m.$i.json 2> $OUT/dev_m.$i.err; show $OUT/dev_m.$i.json
done
