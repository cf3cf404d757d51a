#!/bin/bash
# round 5, visit y: rows reserved a launch ahead for the ids the admission filter will admit (build role peeks at
# the filter) against no reservation (MHTE_FILTER_NO_RESERVE=1) — development build, same box
set -u
OUT=gpurun_out/r05y
mkdir -p $OUT
export TMPDIR=/tmp
for r in 1 2; do
  MHTE_LIBRARY=monolith_amd/libmhte_dev_flt.so timeout 300 python scripts/next_rows_bench.py filter 2>> $OUT/err.txt | tee -a $OUT/filter_reserve.md
  MHTE_FILTER_NO_RESERVE=1 MHTE_LIBRARY=monolith_amd/libmhte_dev_flt.so timeout 300 python scripts/next_rows_bench.py filter 2>> $OUT/err.txt | tee -a $OUT/filter_no_reserve.md
done
