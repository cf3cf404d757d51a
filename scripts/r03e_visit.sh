#!/bin/bash
# GPU visit r03e: the step kernels without their loop-entry spills (anti-hoist barriers, 32-bit trip
# indices) — parity suite, same-box A/B against the previous library on both bench shapes, the PMC
# write pass of the headline step, and the write-through row store variant (-DMHTE_ROW_STORE_WT).
export TMPDIR=/tmp
OUT=gpurun_out/r03e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_multi_step_gpu.py tests/test_shard_step_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
ab() { # name lib args...
  local name=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export MHTE_LIBRARY=$lib; else unset MHTE_LIBRARY; fi
  timeout 300 python bench.py --no-cpu-baseline --no-parity-check "$@" >> $OUT/ab_$name.jsonl 2>> $OUT/ab_$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/ab_$name.jsonl").read().strip().splitlines()[-1])
print("$name", "us/step %.2f" % (d["ms_per_step"]*1e3), d.get("timing_ms_per_step"), {k: v_["avg_us"] for k, v_ in d.get("stages", {}).items() if "step" in k})
PY
  unset MHTE_LIBRARY
}
for i in 1 2; do
  ab default_prev monolith_amd/libmhte_prev.so
  ab default_new ""
  ab default_wt monolith_amd/libmhte_wt.so
done
for i in 1 2; do
  ab dlrm26_prev monolith_amd/libmhte_prev.so --config dlrm26
  ab dlrm26_new "" --config dlrm26
  ab dlrm26_wt monolith_amd/libmhte_wt.so --config dlrm26
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/new_$c && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/new_$c -o pmc -- \
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity-check --launch eager > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
  echo "pmc $c rc=$?"
  for f in $(find /tmp/new_$c -name '*counter_collection*.csv'); do python scripts/pmc_csv.py $f > $OUT/pmc_$c.md; grep "step_" $OUT/pmc_$c.md; done
done
rm -rf /tmp/dprofw && MHTE_LIBRARY=monolith_amd/libmhte_wt.so timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/dprofw -o trace -- \
  python bench.py --config dlrm26 --steps 50 --warmup 10 --no-cpu-baseline --no-parity-check > $OUT/prof_dlrm26_wt.json 2> $OUT/prof_dlrm26_wt.err
db=$(find /tmp/dprofw -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_dlrm26_wt.md | grep -E "mstep|kernel \|"; fi
