#!/bin/bash
# round 5, visit l: the filter's element counters one per 128-byte line
set -u
OUT=gpurun_out/r05l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "filter or Filter" > $OUT/pytest_filter.log 2>&1; echo "pytest filter rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_filter.log
timeout 300 python scripts/next_rows_bench.py filter > $OUT/filter_step.md 2> $OUT/filter_step.err; echo "filter bench rc=$?"
cat $OUT/filter_step.md
MHTE_FILTER_MAINTAIN_ALWAYS=1 timeout 300 python scripts/next_rows_bench.py filter > $OUT/filter_step_maintain_always.md 2>> $OUT/filter_step.err
cat $OUT/filter_step_maintain_always.md
