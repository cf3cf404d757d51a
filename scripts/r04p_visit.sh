#!/bin/bash
# GPU visit: the owner's upsert of the sharded step (13.75 us in round 3, 17.5 now): with / without the
# opt-in AVX form of Adagrad compiled in, and at 4 / 5 workgroups per CU (dev builds, dim 64 only)
export TMPDIR=/tmp MHTE_NO_REBUILD=1
OUT=gpurun_out/${1:-r04p}; mkdir -p $OUT
show() { python - "$1" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1], "us/step %.2f" % (d["ms_per_step"]*1e3), {k: v_.get("avg_us") for k, v_ in d.get("stages", {}).items() if isinstance(v_, dict)})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2; do
  for v in a b c d; do
    MHTE_LIBRARY=monolith_amd/libmhte_dev_$v.so timeout 300 python bench.py --force-sharded --no-cpu-baseline --no-parity-check > $OUT/sh_$v.$i.json 2> $OUT/sh_$v.$i.err; show $OUT/sh_$v.$i.json
  done
done
for v in a b; do
  MHTE_LIBRARY=monolith_amd/libmhte_dev_$v.so timeout 300 python bench.py --no-cpu-baseline --no-parity-check > $OUT/def_$v.json 2> $OUT/def_$v.err; show $OUT/def_$v.json
done
