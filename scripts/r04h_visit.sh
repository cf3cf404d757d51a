#!/bin/bash
# GPU visit: the library with the dense tower: full parity suite, smoke, the tower's bench and per-kernel
# summary, the configs[4] step with the dense leg through mhte_dense_mlp_* and through torch
export TMPDIR=/tmp MHTE_NO_REBUILD=1
OUT=gpurun_out/${1:-r04h}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $OUT/pytest_gpu.log | tail -5
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 200 python scripts/gemm_bench.py 2>&1 | grep "^{" > $OUT/gemm_bench.jsonl; cut -c1-230 $OUT/gemm_bench.jsonl
rm -rf /tmp/gprof && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/gprof -o trace -- python scripts/gemm_bench.py > /dev/null 2> $OUT/prof.err
db=$(find /tmp/gprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_gemm.md --by-grid | grep "mhte::" | cut -c1-150; fi
timeout 900 python bench.py --config dlrm26 --dense --no-cpu-baseline > $OUT/bench_dlrm26_dense.json 2> $OUT/bench_dlrm26_dense.err; echo "dense rc=$?"
python -c "import json; d=json.load(open('$OUT/bench_dlrm26_dense.json')); print(d['ms_per_step'], d['dense'])"
timeout 900 python bench.py --config dlrm26 --dense --mlp-impl torch --no-cpu-baseline > $OUT/bench_dlrm26_dense_torch.json 2> $OUT/bench_dlrm26_dense_torch.err; echo "dense torch rc=$?"
python -c "import json; d=json.load(open('$OUT/bench_dlrm26_dense_torch.json')); print(d['ms_per_step'], d['dense'])"
