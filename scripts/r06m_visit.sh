#!/bin/bash
# round 6, visit m: the bit-exact mode of the MULTI-TABLE step on the pre-sum kernel (mstep_exact_sum_kernel in front
# of mstep_bwd) against round 5's walk (MHTE_EXACT_WALK=1), one binary; the suite first.
set -u
OUT=gpurun_out/r06m
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -2; grep -n "^E " $OUT/pytest_gpu.log | head -8
for v in 0 1; do
  MHTE_EXACT_WALK=$v timeout 900 python bench.py --config dlrm26 --exact-order --steps 40 --warmup 10 --no-cpu-baseline \
    > $OUT/bench_dlrm26_exact_walk$v.json 2> $OUT/bench_dlrm26_exact_walk$v.err
  python -c "
import json
d = json.load(open('$OUT/bench_dlrm26_exact_walk$v.json'))
pc = d.get('parity_check') or {}
print('walk=$v dlrm26 exact', d['ms_per_step'], {k: (v.get('rows_bit_exact'), v.get('n'), v.get('max_abs')) for k, v in list(pc.items())[:3]} if isinstance(pc, dict) else pc)" 2>&1 | cut -c1-300
done
timeout 900 python bench.py --config dlrm26 --no-cpu-baseline > $OUT/bench_dlrm26.json 2> $OUT/bench_dlrm26.err
python -c "import json; d=json.load(open('$OUT/bench_dlrm26.json')); print('dlrm26 default', d['ms_per_step'])"
