#!/bin/bash
# diagnostic: which switch makes test_processes_against_oracle[uniform-2] fail intermittently
set -u
OUT=gpurun_out/r05i
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  fails=0
  for i in 1 2 3 4 5 6; do
    env "$@" timeout 300 python -m pytest "tests/test_shard_ipc_gpu.py::test_processes_against_oracle" -m gpu -x -q > $OUT/${name}_$i.log 2>&1 || { fails=$((fails+1)); grep -n "Mismatched\|step [0-9]\|rank [0-9]:" $OUT/${name}_$i.log | head -4; }
  done
  echo "$name: $fails failures of 6" | tee -a $OUT/summary.txt
}
run default X=1
run unr2 MHTE_SHARD_LOOKUP_UNR=2
run nofold MHTE_SHARD_FOLD_SLOW=0
run nodirect MHTE_SHARD_DIRECT=0
run nofast MHTE_SHARD_NO_FAST_APPLY=1
