#!/usr/bin/env python3
"""Table of the bench lines of one A/B GPU visit (gpurun_out/<visit>/*.json): us per step and the
HIP-event time of the tagged kernels.  Usage: ab_summary.py <dir> [title...] > profiles/rNN/<name>.md"""
import glob, json, os, sys

d = sys.argv[1]
print("# " + (" ".join(sys.argv[2:]) or d))
print()
print("| run | us/step | kernels (avg us, HIP events on the launch stream) |")
print("|---|---|---|")
for f in sorted(glob.glob(os.path.join(d, "*.json"))):
  try:
    line = open(f).read().strip().splitlines()[-1]
    j = json.loads(line)
  except Exception:
    continue
  if "ms_per_step" not in j:
    continue
  st = j.get("stages", {})
  ks = ", ".join("%s %.2f" % (k.replace("_kernel", ""), v["avg_us"]) for k, v in st.items()
                 if isinstance(v, dict) and "avg_us" in v and not k.startswith("unzipped"))
  print("| %s | %.2f | %s |" % (os.path.basename(f)[:-5], j["ms_per_step"] * 1e3, ks))
