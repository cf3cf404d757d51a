#!/bin/bash
# Round 5, visit a: baseline of this round's box (tests, the driver's bench line with the new eager_cpp /
# exact_order windows, the sharded step's three reference numbers) + the hunt for round 4's rc -11 of the
# CPU baseline's shared-map variant (256 host threads on ONE reference cuckoohash_map).
set -u
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
# ---- the shared-map stress: plain build 12 times, ASan twice, the bench's own child 3 times
P=$(nproc)
for i in $(seq 1 12); do
  timeout 300 oracle/_ref/shared_map_stress $P 40 > $OUT/stress_$i.log 2>&1; rc=$?
  echo "stress run $i threads $P rc $rc" | tee -a $OUT/stress_summary.txt
  [ $rc -ne 0 ] && tail -40 $OUT/stress_$i.log
done
for i in 1 2; do
  timeout 600 oracle/_ref/shared_map_stress_asan $P 30 > $OUT/stress_asan_$i.log 2>&1; rc=$?
  echo "stress asan run $i rc $rc" | tee -a $OUT/stress_summary.txt
  [ $rc -ne 0 ] && tail -60 $OUT/stress_asan_$i.log
done
for i in 1 2 3; do
  timeout 600 python -X faulthandler bench.py --cpu-child ii --cpu-steps 75 > $OUT/child_ii_$i.json 2> $OUT/child_ii_$i.err; rc=$?
  echo "bench --cpu-child ii run $i rc $rc" | tee -a $OUT/stress_summary.txt
  [ $rc -ne 0 ] && tail -30 $OUT/child_ii_$i.err
done
# ---- bench lines
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; echo "bench rc=$?"
cat $OUT/bench_driver_args.json | cut -c1-1500
timeout 600 python bench.py --no-cpu-baseline --force-sharded > $OUT/sharded_n1.json 2> $OUT/sharded_n1.err; echo "sharded rc=$?"
cut -c1-600 $OUT/sharded_n1.json
timeout 600 python bench.py --no-cpu-baseline --gpus 2 --steps 100 --warmup 10 > $OUT/ranks2.json 2> $OUT/ranks2.err; echo "ranks2 rc=$?"
cut -c1-600 $OUT/ranks2.json
timeout 600 python bench.py --no-cpu-baseline --config dlrm26 --force-sharded --steps 100 --warmup 10 > $OUT/sharded_dlrm26.json 2> $OUT/sharded_dlrm26.err; echo "sharded dlrm rc=$?"
cut -c1-400 $OUT/sharded_dlrm26.json
