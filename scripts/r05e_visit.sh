#!/bin/bash
# Round 5, visit c: owner launches sized by the counts the peers actually send, scatter ids dealt out
# over the workgroups — per-kernel durations of the sharded step under rocprofv3 + the A/Bs.
set -u
OUT=gpurun_out/r05e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_shard.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_shard.log
tail -4 $OUT/pytest_shard.log
for v in new nofast; do
  case $v in
    new) ENVV="" ;;
    nofast) ENVV="MHTE_SHARD_NO_FAST_APPLY=1" ;;
    nofuse) ENVV="MHTE_SHARD_FUSE_SCATTER=0" ;;
  esac
  env $ENVV timeout 600 python bench.py --no-cpu-baseline --force-sharded --no-parity-check > $OUT/sharded_n1_$v.json 2> $OUT/sharded_n1_$v.err; echo "sharded $v rc=$?"
  python - $OUT/sharded_n1_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d["stages"].items() if isinstance(v,dict)})
PY
done
rm -rf /tmp/sprof && timeout -k 5 400 rocprofv3 --kernel-trace --stats -d /tmp/sprof -o trace -- \
  python bench.py --force-sharded --steps 200 --warmup 20 --no-cpu-baseline --no-parity-check --no-stage-timing > $OUT/prof_sharded_n1.json 2> $OUT/prof_sharded_n1.err
echo "shardprof rc=$?"
db=$(find /tmp/sprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/sharded_n1_kernel_stats.md --by-grid --timeline 30 | head -60; fi
timeout 600 python bench.py --no-cpu-baseline --gpus 2 --steps 100 --warmup 10 > $OUT/ranks2.json 2> $OUT/ranks2.err; echo "ranks2 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05e/ranks2.json"))
print(d["ms_per_step"], d["value"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d["stages"].items() if isinstance(v,dict)})
PY
timeout 600 python bench.py --no-cpu-baseline --config dlrm26 --force-sharded --steps 100 --warmup 10 > $OUT/sharded_dlrm26.json 2> $OUT/sharded_dlrm26.err; echo "sharded dlrm rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05e/sharded_dlrm26.json"))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d.get("stages",{}).items() if isinstance(v,dict)})
PY
