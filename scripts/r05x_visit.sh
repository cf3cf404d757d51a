#!/bin/bash
# round 5, visit x: gradient sums with a sliding window of 8 rows in flight (lists of 9-32 occurrences, item
# windows) against batches of 8 — development builds (G = 16 only), same box
set -u
OUT=gpurun_out/r05x
mkdir -p $OUT
export TMPDIR=/tmp
for r in 1 2; do
for n in pipe nopipe; do
  MHTE_LIBRARY=monolith_amd/libmhte_dev_$n.so timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-extra-windows > $OUT/bench_${n}_$r.json 2> $OUT/err_${n}_$r.txt; echo "$n rc=$?"
  python - $OUT/bench_${n}_$r.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["ms_per_step"], {k:v["avg_us"] for k,v in d["stages"].items() if k.startswith("step_")}, d["parity_check"]["rows_bit_exact"], d["parity_check"]["n"], d["parity_check"]["max_abs"])
PY
done
done
