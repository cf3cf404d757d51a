#!/bin/bash
# round 5, visit p: grouping for the pooling gradients by a stable radix sort (AuxWs::group_sorted) against
# the list-building dedup (MHTE_GROUP_DD=1)
set -u
OUT=gpurun_out/r05p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "gather or reduce or layout or pool or segment" > $OUT/pytest_pool.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_pool.log
timeout 300 python scripts/next_rows_bench.py gather reduce > $OUT/pool_sorted.jsonl 2> $OUT/pool.err; echo "bench rc=$?"
cat $OUT/pool_sorted.jsonl
MHTE_GROUP_DD=1 timeout 300 python scripts/next_rows_bench.py gather reduce > $OUT/pool_dd.jsonl 2>> $OUT/pool.err
cat $OUT/pool_dd.jsonl
MHTE_POOL_ATOMICS=1 timeout 300 python scripts/next_rows_bench.py gather reduce > $OUT/pool_atomics.jsonl 2>> $OUT/pool.err
cat $OUT/pool_atomics.jsonl
rm -rf /tmp/pprof && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/pprof -o trace -- python scripts/next_rows_bench.py gather reduce > $OUT/prof_run.jsonl 2> $OUT/prof.err
db=$(find /tmp/pprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_pooling.md | head -30 | cut -c1-170; fi
