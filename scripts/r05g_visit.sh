#!/bin/bash
# Round 5, visit g: whole GPU suite on the current library + the sharded step's reference numbers
set -u
OUT=gpurun_out/r05g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --force-sharded > $OUT/sharded_n1.json 2> $OUT/sharded_n1.err; echo "sharded rc=$?"
python - $OUT/sharded_n1.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d["stages"].items() if isinstance(v,dict)}, d["config"]["shard_step"], d.get("parity_check"))
PY
timeout 600 python bench.py --no-cpu-baseline --config dlrm26 --force-sharded --steps 100 --warmup 10 > $OUT/sharded_dlrm26.json 2> $OUT/sharded_dlrm26.err; echo "sharded dlrm rc=$?"
python - $OUT/sharded_dlrm26.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d.get("stages",{}).items() if isinstance(v,dict)})
PY
timeout 600 python bench.py --no-cpu-baseline --config dlrm26 --steps 100 --warmup 10 > $OUT/dlrm26.json 2> $OUT/dlrm26.err; echo "dlrm rc=$?"
python - $OUT/dlrm26.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], {k:(v.get("avg_us"),v.get("launches_per_step")) for k,v in d.get("stages",{}).items() if isinstance(v,dict)})
PY
timeout 600 python bench.py --no-cpu-baseline --gpus 2 --steps 100 --warmup 10 > $OUT/ranks2.json 2> $OUT/ranks2.err; echo "ranks2 rc=$?"
python - $OUT/ranks2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["value"], d["config"]["shard_step"])
PY
