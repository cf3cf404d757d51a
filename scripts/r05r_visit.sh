#!/bin/bash
# round 5, visit r (the copy loop column-per-thread): the layout copies writing whole output rows (no zero fill in front), one zero launch for
# the gradient matrices; the filtered step with the older splits fetched only when the window has moved
set -u
OUT=gpurun_out/r05r
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "layout or filter or Filter" > $OUT/pytest_layout_filter.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_layout_filter.log
timeout 300 python scripts/next_rows_bench.py filter > $OUT/filter_step.md 2> $OUT/filter_step.err; echo "filter bench rc=$?"
cat $OUT/filter_step.md
timeout 600 python bench.py --config dlrm26 --dense --no-cpu-baseline --steps 30 --warmup 5 > $OUT/dlrm26_dense_rows.json 2> $OUT/dense1.err; echo "dense rc=$?"
MHTE_LAYOUT_ROWS=0 timeout 600 python bench.py --config dlrm26 --dense --no-cpu-baseline --steps 30 --warmup 5 > $OUT/dlrm26_dense_groups.json 2> $OUT/dense2.err; echo "dense rc=$?"
python - $OUT/dlrm26_dense_rows.json $OUT/dlrm26_dense_groups.json <<'PY'
import json,sys
for f in sys.argv[1:]:
  d=json.load(open(f))
  dn=d["dense"]
  print(f, "ms/step", d["ms_per_step"], "dense_leg", dn["dense_leg_us"], "mlp", dn["mlp_us"], "layout", dn["layout_fwd_bwd_us"])
PY
rm -rf /tmp/dprof && timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/dprof -o trace -- \
  python bench.py --config dlrm26 --dense --no-cpu-baseline --no-parity-check --steps 10 --warmup 3 > $OUT/prof_run.json 2> $OUT/prof.err
echo "rocprof rc=$?"
db=$(find /tmp/dprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_dlrm26_dense.md | head -40; fi
