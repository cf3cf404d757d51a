#!/bin/bash
# round 6, visit h: how much of the run dedup's time is its device-scope atomics?  Timing-only development builds
# with 0 / 1 / 2 EXTRA atomics per run (one CAS + one add is what a run costs today: +50 % / +100 %), the
# single-table step and 26 tables of dim 64, interleaved twice.  Parity checks off (the extra adds corrupt lists of 32).
set -u
OUT=gpurun_out/r06h
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
  for v in x0 x1 x2; do
    MHTE_LIBRARY=$PWD/monolith_amd/libmhte_dev_$v.so timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-windows --no-parity-check \
      > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
    python -c "
import json
d = json.load(open('$OUT/bench_${v}_$rep.json')); st = d.get('stages', {})
print('$v $rep single', d['ms_per_step'], {k: st[k]['avg_us'] for k in ('step_bwd_kernel', 'step_fwd_kernel') if k in st})"
    MHTE_LIBRARY=$PWD/monolith_amd/libmhte_dev_$v.so timeout 600 python bench.py --config dlrm26 --dims 64 --no-cpu-baseline --no-parity-check \
      > $OUT/dlrm_${v}_$rep.json 2> $OUT/dlrm_${v}_$rep.err
    python -c "
import json
d = json.load(open('$OUT/dlrm_${v}_$rep.json')); st = d.get('stages', {})
print('$v $rep dlrm26(dim 64)', d['ms_per_step'], {k: st[k]['avg_us'] for k in st})"
  done
done
