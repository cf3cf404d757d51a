#!/bin/bash
# round 5, visit q: the process tests of the sharded step repeated on the final library (flakiness check)
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
fails=0
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_shard_ipc_gpu.py -m gpu -x -q > $OUT/ipc_$i.log 2>&1 || { fails=$((fails+1)); tail -5 $OUT/ipc_$i.log; }
done
echo "tests/test_shard_ipc_gpu.py: $fails failures of 6 runs" | tee $OUT/summary.txt
tail -2 $OUT/ipc_1.log
