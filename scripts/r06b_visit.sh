#!/bin/bash
# round 6, visit b: the GPU suite after the fix of visit a's hang (mstep_bwd built its ApplyCtl field by field: the
# new pre_summed flag was an uninitialised register there, lanes disagreed and a workgroup barrier never filled),
# the packed one-pair-per-peer RCCL exchange, the grouping with 4 096-key tiles and 8-byte words between passes;
# then the eager C loop and the graph replay under a kernel trace: where the eager step's extra 13 us are.
set -u
OUT=gpurun_out/r06b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -4 $OUT/pytest_gpu.log
timeout 300 python scripts/next_rows_bench.py gather reduce > $OUT/pool_sorted.jsonl 2> $OUT/pool.err; echo "pool rc=$?"
cut -c1-200 $OUT/pool_sorted.jsonl | grep -i "gradient\|unsorted"
rm -rf /tmp/pprof && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/pprof -o trace -- python scripts/next_rows_bench.py gather reduce > $OUT/prof_run.jsonl 2> $OUT/prof.err
db=$(find /tmp/pprof -name '*.db' | head -1)
if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/kernel_stats_pooling.md | head -16 | cut -c1-150; fi
# eager C loop vs graph replay: kernel durations and the gaps between them (timeline of 40 dispatches each)
for mode in eager graph; do
  rm -rf /tmp/tl_$mode && timeout -k 5 600 rocprofv3 --kernel-trace -d /tmp/tl_$mode -o trace -- \
    python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-parity-check --no-stage-timing --launch $mode > $OUT/tl_${mode}_bench.json 2> $OUT/tl_$mode.err
  echo "tl $mode rc=$?"
  db=$(find /tmp/tl_$mode -name '*.db' | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_stats.py $db $OUT/timeline_$mode.md --by-grid --timeline 60 > /dev/null; grep -c . $OUT/timeline_$mode.md; fi
done
python - <<'EOF'
import re, sys
for mode in ("eager", "graph"):
  try:
    rows = [l.split("|") for l in open("gpurun_out/r06b/timeline_%s.md" % mode) if re.match(r"\| *[0-9.]+ *\|", l)]
  except OSError:
    continue
  ks = [(float(r[1]), float(r[2]), r[6].strip()) for r in rows if "step_" in r[6]]
  gaps = [b[0] - a[1] for a, b in zip(ks, ks[1:])]
  durs = {}
  for s, e, k in ks:
    durs.setdefault(k.split("<")[0], []).append(e - s)
  if gaps:
    gaps.sort()
    print(mode, "gaps between step kernels us: median %.2f p90 %.2f" % (gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)]),
          {k: round(sum(v) / len(v), 2) for k, v in durs.items()})
EOF
# the HIP runtime keeps the kernel arguments of an EAGER launch in host memory unless told otherwise (graph nodes get
# theirs in device memory): the step kernels read ~1.5 KB of arguments each — A/B of the driver's line
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check > $OUT/bench_devkernarg_$v.json 2> $OUT/bench_devkernarg_$v.err
  python - <<EOF2
import json
d = json.load(open("$OUT/bench_devkernarg_$v.json"))
print("HIP_FORCE_DEV_KERNARG=$v", d.get("ms_per_step"), d.get("timing_ms_per_step"), d["roofline"].get("avg_launch_us"))
EOF2
done
