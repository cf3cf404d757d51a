#!/bin/bash
# GPU visit: the dense tower, LDS-DMA main loop (dev) against the register-staged one (dev_reg): parity
# test, bench, per-kernel summary
export TMPDIR=/tmp MHTE_NO_REBUILD=1
OUT=gpurun_out/${1:-r04gemm}; mkdir -p $OUT
MHTE_DENSE_LIBRARY=monolith_amd/libmhte_gemm_dev.so timeout 300 python -m pytest tests/test_dense_mlp_gpu.py -q 2>&1 | grep -E "AssertionError|passed|failed" | head -6
for v in dev dev_reg; do
  MHTE_DENSE_LIBRARY=monolith_amd/libmhte_gemm_$v.so timeout 200 python scripts/gemm_bench.py 2>&1 | grep "^{" | head -1 | cut -c1-200
  rm -rf /tmp/gprof && MHTE_DENSE_LIBRARY=monolith_amd/libmhte_gemm_$v.so timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/gprof -o trace -- python scripts/gemm_bench.py > /dev/null 2> $OUT/prof_$v.err
  db=$(find /tmp/gprof -name '*.db' | head -1)
  echo "== $v"
  if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_gemm_$v.md --by-grid | grep "gemm_nt_bf16_kernel" | cut -c1-120; fi
done
