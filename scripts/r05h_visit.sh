#!/bin/bash
# Round 5, visit h: per-optimizer FULL instances of step_bwd (OPTK) with and without the row fetched ahead,
# against the one FULL instance; the sharded step with the lookup's UNR choice; whole suite.
set -u
OUT=gpurun_out/r05h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
NEXT_ROWS_MD=$OUT/step_optimizers_optk_prefetch.md timeout 600 python scripts/next_rows_bench.py step_optimizers > $OUT/step_optimizers_optk_prefetch.jsonl 2> $OUT/so1.err; cat $OUT/step_optimizers_optk_prefetch.md
MHTE_NO_OPTK=1 NEXT_ROWS_MD=$OUT/step_optimizers_one_full_instance.md timeout 600 python scripts/next_rows_bench.py step_optimizers > $OUT/step_optimizers_one_full_instance.jsonl 2> $OUT/so2.err; cat $OUT/step_optimizers_one_full_instance.md
MHTE_LIBRARY=monolith_amd/libmhte_nopf.so NEXT_ROWS_MD=$OUT/step_optimizers_optk_no_prefetch.md timeout 600 python scripts/next_rows_bench.py step_optimizers > $OUT/step_optimizers_optk_no_prefetch.jsonl 2> $OUT/so3.err; cat $OUT/step_optimizers_optk_no_prefetch.md; tail -3 $OUT/so3.err
timeout 600 python bench.py --no-cpu-baseline --config dlrm26 --force-sharded --steps 100 --warmup 10 > $OUT/sharded_dlrm26.json 2> $OUT/sharded_dlrm26.err; echo "sharded dlrm rc=$?"
python - $OUT/sharded_dlrm26.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d.get("stages",{}).items() if isinstance(v,dict)})
PY
MHTE_SHARD_LOOKUP_UNR=2 timeout 600 python bench.py --no-cpu-baseline --config dlrm26 --force-sharded --steps 100 --warmup 10 > $OUT/sharded_dlrm26_unr2.json 2> $OUT/sharded_dlrm26_unr2.err; echo "sharded dlrm unr2 rc=$?"
python - $OUT/sharded_dlrm26_unr2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d.get("stages",{}).items() if isinstance(v,dict)})
PY
timeout 600 python bench.py --no-cpu-baseline --force-sharded > $OUT/sharded_n1.json 2> $OUT/sharded_n1.err; echo "sharded rc=$?"
python - $OUT/sharded_n1.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d["stages"].items() if isinstance(v,dict)})
PY
