#!/bin/bash
# Round 5, visit f: direct peer stores (no push launches, two one-wavefront sync points per step) — the
# process tests, then two ranks sharing the GPU in both forms of the exchange.
set -u
OUT=gpurun_out/r05f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_shard_ipc_gpu.py tests/test_shard_step_gpu.py -m gpu -x -q > $OUT/pytest_shard.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_shard.log
tail -25 $OUT/pytest_shard.log
for v in direct push; do
  case $v in
    direct) ENVV="" ;;
    push) ENVV="MHTE_SHARD_DIRECT=0" ;;
  esac
  env $ENVV timeout 600 python bench.py --no-cpu-baseline --gpus 2 --steps 100 --warmup 10 > $OUT/ranks2_$v.json 2> $OUT/ranks2_$v.err; echo "ranks2 $v rc=$?"
  python - $OUT/ranks2_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["value"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d["stages"].items() if isinstance(v,dict)}, d["config"]["shard_step"], d.get("parity_check"))
PY
  tail -3 $OUT/ranks2_$v.err
done
timeout 600 python bench.py --no-cpu-baseline --force-sharded --transport ipc --no-parity-check > $OUT/sharded_n1_ipc.json 2> $OUT/sharded_n1_ipc.err; echo "n1 ipc rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05f/sharded_n1_ipc.json"))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d["stages"].items() if isinstance(v,dict)}, d["config"]["shard_step"])
PY
