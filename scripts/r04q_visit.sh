#!/bin/bash
# GPU visit: item workgroups of step_bwd — base = library on disk; a = workgroups without a next item
# leave without fetching another header; b = a + the id's row prefetched behind the groups' sums
export TMPDIR=/tmp MHTE_NO_REBUILD=1
OUT=gpurun_out/${1:-r04q}; mkdir -p $OUT
show() { python - "$1" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1], "us/step %.2f" % (d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), {k: v_.get("avg_us") for k, v_ in d.get("stages", {}).items() if isinstance(v_, dict) and not k.startswith("unzipped")})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-parity-check > $OUT/base.$i.json 2> $OUT/base.$i.err; show $OUT/base.$i.json
  for v in a b; do
    MHTE_LIBRARY=monolith_amd/libmhte_dev_$v.so timeout 300 python bench.py --no-cpu-baseline --no-parity-check > $OUT/$v.$i.json 2> $OUT/$v.$i.err; show $OUT/$v.$i.json
  done
done
MHTE_LIBRARY=monolith_amd/libmhte_dev_b.so timeout 300 python -m pytest tests/test_parity_gpu.py -q -x -k "pipelined or lookahead or full_batch or heavy or duplicate" 2>&1 | tail -3
