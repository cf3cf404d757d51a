#!/bin/bash
# Copies the artifacts of one full GPU visit (scripts/gpu_round.sh <tag> tests smoke bench drv prof pmc
# calib trace dlrm dlrmsps1 dlrmdense dlrmprof dlrmpmc shard shardprof ranks2 cfg1 next) from gpurun_out/<tag>/ into
# profiles/<tag>/ under the names profiles/README.md lists, and regenerates profiles/pmc_traffic.json.
set -u
TAG=${1:?tag}; SRC=gpurun_out/$TAG; DST=profiles/$TAG
mkdir -p $DST
cpy() { [ -s "$SRC/$1" ] && cp "$SRC/$1" "$DST/$2" || echo "missing: $SRC/$1"; }
cpy bench.json bench_default.json
cpy bench_driver_args.json bench_driver_args.json
cpy prof_bench.json bench_default_under_rocprof.json
cpy kernel_stats.md kernel_stats_default.md
cpy pmc_FETCH_SIZE.md pmc_FETCH_SIZE.md
cpy pmc_WRITE_SIZE.md pmc_WRITE_SIZE.md
cpy pmc_dlrm_FETCH_SIZE.md pmc_dlrm26_FETCH_SIZE.md
cpy pmc_dlrm_WRITE_SIZE.md pmc_dlrm26_WRITE_SIZE.md
cpy calib.json pmc_calibration.json
cpy calib_FETCH_SIZE.md pmc_calibration_FETCH_SIZE.md
cpy calib_WRITE_SIZE.md pmc_calibration_WRITE_SIZE.md
cpy trace_report.md wave_timeline.md
cpy bench_dlrm26.json bench_dlrm26.json
cpy bench_dlrm26_sps1.json bench_dlrm26_new_second_every_step.json
cpy bench_dlrm26_dense.json bench_dlrm26_dense.json
cpy prof_dlrm26_bench.json bench_dlrm26_under_rocprof.json
cpy kernel_stats_dlrm26.md kernel_stats_dlrm26.md
cpy bench_sharded_n1.json sharded_n1_identity_bench.json
cpy bench_sharded_n1_dlrm26.json sharded_n1_dlrm26_bench.json
cpy prof_sharded_n1_bench.json sharded_n1_bench_under_rocprof.json
cpy kernel_stats_sharded_n1.md sharded_n1_kernel_stats.md
cpy bench_2ranks.json bench_2ranks_ipc_one_gpu.json
cpy bench_configs1.json bench_configs1.json
cpy next_rows.md next_rows.md
cpy next_rows.jsonl next_rows.jsonl
cpy pytest_gpu.log pytest_gpu.log
cpy smoke.log smoke.log
python scripts/pmc_traffic.py $SRC $TAG
