#!/bin/bash
# Round 5, visit b: the rebuilt id-sharded step (one owner launch for all peers, scatter | dedup fused,
# lookup records as update hints) — tests, the three reference numbers of the sharded step with per-kernel
# stages, the A/B against the per-peer owner form, and the CPU baseline's shared map from 2^18 slots.
set -u
OUT=gpurun_out/r05b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
P=$(nproc)
for i in $(seq 1 14); do
  timeout 300 oracle/_ref/shared_map_stress $P 30 zipf 262144 > $OUT/stress_$i.log 2>&1; rc=$?
  echo "stress from 2^18 slots, run $i threads $P rc $rc" | tee -a $OUT/stress_summary.txt
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05b/bench_driver_args.json"))
print({k:d[k] for k in ("value","ms_per_step","timing_ms_per_step")}, d["roofline"]["frac"], d["cpu_baseline"]["variants"]["ii_shared_table"])
PY
tail -3 $OUT/bench_driver_args.err
for v in new perpeer nofuse; do
  case $v in
    new) ENVV="" ;;
    perpeer) ENVV="MHTE_SHARD_PER_PEER=1 MHTE_SHARD_FUSE_SCATTER=0" ;;
    nofuse) ENVV="MHTE_SHARD_FUSE_SCATTER=0" ;;
  esac
  env $ENVV timeout 600 python bench.py --no-cpu-baseline --force-sharded > $OUT/sharded_n1_$v.json 2> $OUT/sharded_n1_$v.err; echo "sharded $v rc=$?"
  python - $OUT/sharded_n1_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d["stages"].items() if isinstance(v,dict)}, d["config"]["shard_step"])
PY
done
timeout 600 python bench.py --no-cpu-baseline --gpus 2 --steps 100 --warmup 10 > $OUT/ranks2.json 2> $OUT/ranks2.err; echo "ranks2 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05b/ranks2.json"))
print(d["ms_per_step"], d["value"], d["config"]["shard_step"], d.get("parity_check"))
PY
MHTE_SHARD_PER_PEER=1 MHTE_SHARD_FUSE_SCATTER=0 timeout 600 python bench.py --no-cpu-baseline --gpus 2 --steps 100 --warmup 10 > $OUT/ranks2_perpeer.json 2> $OUT/ranks2_perpeer.err; echo "ranks2 perpeer rc=$?"
cut -c1-300 $OUT/ranks2_perpeer.json
timeout 600 python bench.py --no-cpu-baseline --config dlrm26 --force-sharded --steps 100 --warmup 10 > $OUT/sharded_dlrm26.json 2> $OUT/sharded_dlrm26.err; echo "sharded dlrm rc=$?"
cut -c1-400 $OUT/sharded_dlrm26.json
