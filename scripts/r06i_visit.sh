#!/bin/bash
# round 6, visit i: the hot list (runs of the ids that occurred >= 3 times in the previous batch resolve without a
# device atomic) against MHTE_NO_HOT_LIST=1 on one development binary (dim-64 float4 shapes), interleaved twice:
# single table and 26 tables of dim 64, with the bench's parity checks against the oracle's replay.
set -u
OUT=gpurun_out/r06i
mkdir -p $OUT
export TMPDIR=/tmp
export MHTE_LIBRARY=$PWD/monolith_amd/libmhte_dev_hot.so
for rep in 1 2; do
  for v in 1 0; do
    MHTE_NO_HOT_LIST=$v timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-windows \
      > $OUT/bench_nohot${v}_$rep.json 2> $OUT/bench_nohot${v}_$rep.err
    python -c "
import json
d = json.load(open('$OUT/bench_nohot${v}_$rep.json')); st = d.get('stages', {})
print('no_hot=$v $rep single', d['ms_per_step'], {k: st[k]['avg_us'] for k in ('step_bwd_kernel', 'step_fwd_kernel') if k in st}, d['parity_check'].get('rows_bit_exact'), d['parity_check'].get('n'), d['parity_check'].get('max_abs'), d['parity_check'].get('failed'))"
    MHTE_NO_HOT_LIST=$v timeout 600 python bench.py --config dlrm26 --dims 64 --no-cpu-baseline \
      > $OUT/dlrm_nohot${v}_$rep.json 2> $OUT/dlrm_nohot${v}_$rep.err
    python -c "
import json
d = json.load(open('$OUT/dlrm_nohot${v}_$rep.json')); st = d.get('stages', {})
print('no_hot=$v $rep dlrm26(dim 64)', d['ms_per_step'], {k: st[k]['avg_us'] for k in st}, d.get('parity_check'))" 2>&1 | cut -c1-400
  done
done
tail -3 $OUT/bench_nohot0_1.err
