#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite (…_results.db) into a per-kernel stats table
(name, calls, total/avg/min/max duration in us, % of GPU kernel time) — the same content as
`rocprofv3 --stats` CSV output.  Usage: python scripts/rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def main():
  db = sqlite3.connect(sys.argv[1])
  cur = db.cursor()
  cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
  name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
  rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
  agg = {}
  for name, s, e in rows:
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^void ", "", short)
    a = agg.setdefault(short, [0, 0, 10**18, 0])
    d = e - s
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
  total = sum(a[1] for a in agg.values()) or 1
  lines = ["| kernel | calls | total_us | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
  for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" %
                 (k, a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
  out = "\n".join(lines)
  if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
  print(out)


if __name__ == "__main__":
  main()
