"""Per-launch duration histogram of one kernel from a rocprofv3 --kernel-trace database.
Usage: python scripts/kernel_hist.py <results.db> <kernel name substring>"""
import sqlite3
import sys

import numpy as np


def main(db, pat):
  con = sqlite3.connect(db)
  cur = con.cursor()
  tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
  kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
  ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
  rows = cur.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id" % (kd, ks)).fetchall()
  d = np.array([(e - s) / 1e3 for s, e, n in rows if pat in n])
  print("%s: %d launches, min %.2f med %.2f mean %.2f p90 %.2f p99 %.2f max %.2f us" % (
      pat, d.size, d.min(), np.median(d), d.mean(), np.percentile(d, 90), np.percentile(d, 99), d.max()))
  edges = [0, 14, 15, 16, 17, 18, 20, 25, 30, 40, 1e9]
  h, _ = np.histogram(d, edges)
  for lo, hi, c in zip(edges[:-1], edges[1:], h):
    print("  [%g, %g) us: %d launches, %.1f %% of the time" % (lo, hi, c, 100 * d[(d >= lo) & (d < hi)].sum() / d.sum()))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2])
