#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite (…_results.db) into a per-kernel stats table
(name, calls, total/avg/min/max duration in us, % of GPU kernel time) — the same content as
`rocprofv3 --stats` CSV output.
Usage: python scripts/rocpd_stats.py results.db [out.md] [--by-grid] [--timeline N] [--around NAME]
  --by-grid     split rows by launch grid size (tells apart two roles of one kernel)
  --timeline N  also print N dispatches: start offset, END offset, duration, queue, kernel — the last N,
                or (--around NAME) the N around the 5th-from-last dispatch whose name contains NAME.
                A dispatch that starts before the previous one ended ran beside it (another queue)."""
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
  around = None
  if "--timeline" in sys.argv:
    tl = int(sys.argv[sys.argv.index("--timeline") + 1])
    args = [a for a in args if a != str(tl)]
  if "--around" in sys.argv:
    around = sys.argv[sys.argv.index("--around") + 1]
    args = [a for a in args if a != around]
  db = sqlite3.connect(args[0])
  cur = db.cursor()
  cols = [c[1] for c in cur.execute("pragma table_info(kernels)").fetchall()]
  qcol = "queue_id" if "queue_id" in cols else "stream_id"
  rows = cur.execute("select name, start, end, grid_x, workgroup_x, %s from kernels order by start" % qcol).fetchall()
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
    sel = rows[-tl:]
    if around:
      hits = [i for i, r in enumerate(rows) if around in r[0]]
      if hits:
        c = hits[max(0, len(hits) - 5)]
        sel = rows[max(0, c - tl // 3):c + tl - tl // 3]
    lines += ["", "%d dispatches (us since the first of them; `beside` = started before the previous one ended):"
              % len(sel), "", "| start | end | dur | queue | beside | kernel | blocks x threads |",
              "|---|---|---|---|---|---|---|"]
    t0 = sel[0][1]
    prev_end = 0
    for name, s, e, gx, wx, st in sel:
      nm = short(name)
      if len(nm) > 60:
        nm = nm[:57] + "..."
      lines.append("| %.2f | %.2f | %.2f | %s | %s | %s | %d x %d |" % (
          (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, st, "yes" if s < prev_end else "", nm,
          gx // max(wx, 1), wx))
      prev_end = max(prev_end, e)
  out = "\n".join(lines)
  if len(args) > 1:
    open(args[1], "w").write(out + "\n")
  print(out)


if __name__ == "__main__":
  main()
