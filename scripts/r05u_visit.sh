#!/bin/bash
# round 5, visit u: the whole GPU suite three times on the final library (flakiness check), smoke
set -u
OUT=gpurun_out/r05u
mkdir -p $OUT
export TMPDIR=/tmp
fails=0
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_$i.log 2>&1 || { fails=$((fails+1)); tail -15 $OUT/pytest_$i.log; }
  grep -n "passed\|failed" $OUT/pytest_$i.log | tail -1
done
echo "pytest -m gpu: $fails failures of 3 runs" | tee $OUT/summary.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
