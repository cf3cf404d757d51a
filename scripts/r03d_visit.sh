#!/bin/bash
# GPU visit r03d: the parallel displacement pass of the multi-table step (tests, same-box A/B against
# the previous library), the PMC passes of the spill-free step_bwd build (VERDICT r2 #3 i), the
# multi-table step's PMC passes, the dense leg's bench lines and a kernel summary of its overlap form.
export TMPDIR=/tmp
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 600 python -m pytest tests/test_multi_step_gpu.py -x -q -m gpu > $OUT/pytest_multi_step.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_multi_step.log
for v in prev new prev new; do
  if [ $v = prev ]; then export MHTE_LIBRARY=monolith_amd/libmhte_prev.so; else unset MHTE_LIBRARY; fi
  timeout 300 python bench.py --config dlrm26 --no-cpu-baseline --no-parity-check >> $OUT/ab_dlrm26_$v.jsonl 2>> $OUT/ab_dlrm26_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/ab_dlrm26_$v.jsonl").read().strip().splitlines()[-1])
print("$v", "us/step %.2f" % (d["ms_per_step"]*1e3), {k: v_["avg_us"] for k, v_ in d.get("stages", {}).items()})
PY
done
unset MHTE_LIBRARY
bash scripts/gpu_round.sh r03d dlrm dlrmprof dlrmpmc
# spill-free step_bwd (4 workgroups per CU, 120 VGPRs): where do the bytes go?
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/occ4_$c && MHTE_LIBRARY=monolith_amd/libmhte_occ4.so timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/occ4_$c -o pmc -- \
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity-check --launch eager > $OUT/pmc_occ4_$c.json 2> $OUT/pmc_occ4_$c.err
  echo "occ4 pmc $c rc=$?"
  for f in $(find /tmp/occ4_$c -name '*counter_collection*.csv'); do python scripts/pmc_csv.py $f > $OUT/pmc_occ4_$c.md; grep "step_" $OUT/pmc_occ4_$c.md; done
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/occ5_$c && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/occ5_$c -o pmc -- \
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity-check --launch eager > $OUT/pmc_occ5_$c.json 2> $OUT/pmc_occ5_$c.err
  echo "occ5 pmc $c rc=$?"
  for f in $(find /tmp/occ5_$c -name '*counter_collection*.csv'); do python scripts/pmc_csv.py $f > $OUT/pmc_occ5_$c.md; grep "step_" $OUT/pmc_occ5_$c.md; done
done
# the dense leg
timeout 600 python bench.py --config dlrm26 --dense --no-cpu-baseline --no-parity-check > $OUT/bench_dlrm26_dense.json 2> $OUT/bench_dlrm26_dense.err; echo "dense rc=$?"
timeout 600 python bench.py --config dlrm26 --dense --force-sharded --no-cpu-baseline --no-parity-check > $OUT/bench_dlrm26_dense_sharded.json 2> $OUT/bench_dlrm26_dense_sharded.err; echo "dense sharded rc=$?"
timeout 600 python bench.py --config dlrm26 --dense --force-sharded --overlap --no-cpu-baseline --no-parity-check > $OUT/bench_dlrm26_dense_sharded_overlap.json 2> $OUT/bench_dlrm26_dense_sharded_overlap.err; echo "dense overlap rc=$?"
for f in dense dense_sharded dense_sharded_overlap; do python - <<PY
import json
try:
  d=json.loads(open("$OUT/bench_dlrm26_$f.json").read().strip().splitlines()[-1]); print("$f", d["ms_per_step"], d.get("dense"))
except Exception as e: print("$f FAILED", e)
PY
done
rm -rf /tmp/oprof && timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/oprof -o trace -- \
  python bench.py --config dlrm26 --dense --force-sharded --overlap --steps 30 --warmup 10 --no-cpu-baseline --no-parity-check > $OUT/prof_dense_overlap.json 2> $OUT/prof_dense_overlap.err
db=$(find /tmp/oprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_dlrm26_dense_sharded_overlap.md | head -24; fi
