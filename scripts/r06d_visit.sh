#!/bin/bash
# round 6, visit d: A/B of the cooperative long-list sum in step_bwd's id-major groups (lists of 9..32 occurrences
# loaded by all lane groups of the wavefront at once, added in list order by the owner) on development builds
# of one source tree (-DMHTE_NO_COOP_LISTS for the old form), interleaved twice.
set -u
OUT=gpurun_out/r06d
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
  for v in nocoop coop; do
    MHTE_LIBRARY=$PWD/monolith_amd/libmhte_dev_$v.so timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-windows \
      > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
    python - <<EOF2
import json
d = json.load(open("$OUT/bench_${v}_$rep.json"))
st = d.get("stages", {})
print("$v $rep", d["ms_per_step"], {k: st[k]["avg_us"] for k in ("step_bwd_kernel", "step_fwd_kernel") if k in st}, d["parity_check"]["rows_bit_exact"], d["parity_check"]["max_abs"])
EOF2
  done
done
# parity of the cooperative form, bit for bit: the step tests on the dev build (dim-64 float4 shapes only)
MHTE_LIBRARY=$PWD/monolith_amd/libmhte_dev_coop.so timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pipelined or training_loop or full_batch or duplicate" > $OUT/pytest_coop.log 2>&1; echo "pytest coop rc=$?"
tail -3 $OUT/pytest_coop.log
