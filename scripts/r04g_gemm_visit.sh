#!/bin/bash
# GPU visit: the dense tower from the development build (scripts/dbg/gemm_dev.hip): parity test, bench,
# per-kernel summary
export TMPDIR=/tmp MHTE_NO_REBUILD=1 MHTE_DENSE_LIBRARY=monolith_amd/libmhte_gemm_dev.so
OUT=gpurun_out/${1:-r04gemm}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_dense_mlp_gpu.py -q 2>&1 | grep -E "AssertionError|passed|failed" | head -6
timeout 200 python scripts/gemm_bench.py 2>&1 | grep "^{" | head -1 | cut -c1-200
rm -rf /tmp/gprof && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/gprof -o trace -- python scripts/gemm_bench.py > /dev/null 2> $OUT/prof.err
db=$(find /tmp/gprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_gemm.md --by-grid | grep "gemm_nt_bf16_kernel" | cut -c1-120; fi
