#!/bin/bash
# GPU visit: full test suite; default bench line (+ CPU baseline); the N = 2 line on one GPU
export TMPDIR=/tmp MHTE_NO_REBUILD=1
OUT=gpurun_out/${1:-r04k}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; cut -c1-700 $OUT/bench_driver_args.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; tail -c 1800 $OUT/bench_2ranks.json; tail -3 $OUT/bench_2ranks.err
