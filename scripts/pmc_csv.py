#!/usr/bin/env python
"""Aggregate a rocprofv3 counter_collection CSV (--pmc ... --output-format csv) per kernel:
calls, mean counter value per dispatch.  Usage: python scripts/pmc_csv.py file.csv [counter]"""
import csv
import re
import sys


def main():
  rows = list(csv.DictReader(open(sys.argv[1])))
  if not rows:
    print("empty")
    return
  kcol = [c for c in rows[0] if c.lower() in ("kernel_name", "kernel-name", "name")][0]
  ccol = [c for c in rows[0] if c.lower() in ("counter_name", "counter-name")][0]
  vcol = [c for c in rows[0] if c.lower() in ("counter_value", "counter-value", "value")][0]
  agg = {}
  for r in rows:
    if len(sys.argv) > 2 and r[ccol] != sys.argv[2]:
      continue
    k = re.sub(r"\(.*", "", r[kcol]).replace("void ", "")
    a = agg.setdefault((k, r[ccol]), [0, 0.0])
    a[0] += 1
    a[1] += float(r[vcol])
  print("| kernel | counter | dispatches | mean_per_dispatch | total |")
  print("|---|---|---|---|---|")
  for (k, c), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("| %s | %s | %d | %.2f | %.1f |" % (k, c, a[0], a[1] / a[0], a[1]))


if __name__ == "__main__":
  main()
