#!/bin/bash
# round 5, visit m: (1) where the filtered step's remaining 10 us are (development builds that leave out the
# element count / the slot CAS / the older splits' windows — timing only, their results are wrong);
# (2) the layout copies as a workgroup per 16 batch rows (layout_rows_kernel) against the lane group per
# (slice, row) form
set -u
OUT=gpurun_out/r05m
mkdir -p $OUT
export TMPDIR=/tmp
for n in base nocount nocas back0; do
  echo "== $n" | tee -a $OUT/filter_variants.md
  MHTE_LIBRARY=monolith_amd/libmhte_fx_$n.so timeout 300 python scripts/next_rows_bench.py filter 2>> $OUT/fx.err | tee -a $OUT/filter_variants.md
done
timeout 600 python -m pytest tests -m gpu -x -q -k "layout" > $OUT/pytest_layout.log 2>&1; echo "pytest layout rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_layout.log
timeout 600 python bench.py --config dlrm26 --dense --no-cpu-baseline --steps 30 --warmup 5 > $OUT/dlrm26_dense_rows.json 2> $OUT/dense1.err; echo "dense rc=$?"
MHTE_LAYOUT_ROWS=0 timeout 600 python bench.py --config dlrm26 --dense --no-cpu-baseline --steps 30 --warmup 5 > $OUT/dlrm26_dense_groups.json 2> $OUT/dense2.err; echo "dense rc=$?"
python - $OUT/dlrm26_dense_rows.json $OUT/dlrm26_dense_groups.json <<'PY'
import json,sys
for f in sys.argv[1:]:
  d=json.load(open(f))
  print(f, d["ms_per_step"], {k:v for k,v in d.get("config",{}).items() if "layout" in k or "mlp" in k}, {k:v for k,v in d.items() if "layout" in k or "dense" in k})
PY
