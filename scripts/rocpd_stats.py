#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite (…_results.db) into a per-kernel stats table
(name, calls, total/avg/min/max duration in us, % of GPU kernel time) — the same content as
`rocprofv3 --stats` CSV output.
Usage: python scripts/rocpd_stats.py results.db [out.md] [--by-grid] [--timeline N]
  --by-grid     split rows by launch grid size (tells apart two roles of one kernel)
  --timeline N  also print the last N dispatches: start offset, duration, stream, kernel"""
import re
import sqlite3
import sys


def short(name):
  s = re.sub(r"\(.*", "", name)
  return re.sub(r"^void ", "", s)


def main():
  args = [a for a in sys.argv[1:] if not a.startswith("--")]
  by_grid = "--by-grid" in sys.argv
  tl = 0
  if "--timeline" in sys.argv:
    tl = int(sys.argv[sys.argv.index("--timeline") + 1])
    args = [a for a in args if a != str(tl)]
  db = sqlite3.connect(args[0])
  cur = db.cursor()
  rows = cur.execute("select name, start, end, grid_x, workgroup_x, stream_id from kernels "
                     "order by start").fetchall()
  agg = {}
  for name, s, e, gx, wx, _ in rows:
    key = short(name) + ((" [grid %d x %d]" % (gx // max(wx, 1), wx)) if by_grid else "")
    a = agg.setdefault(key, [0, 0, 10**18, 0])
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
  if tl:
    lines += ["", "last %d dispatches (us since the first of them):" % tl, "",
              "| start | dur | stream | kernel | blocks x threads |", "|---|---|---|---|---|"]
    t0 = rows[-tl][1]
    for name, s, e, gx, wx, st in rows[-tl:]:
      lines.append("| %.2f | %.2f | %s | %s | %d x %d |" % ((s - t0) / 1e3, (e - s) / 1e3, st,
                                                         short(name), gx // max(wx, 1), wx))
  out = "\n".join(lines)
  if len(args) > 1:
    open(args[1], "w").write(out + "\n")
  print(out)


if __name__ == "__main__":
  main()
