#!/usr/bin/env python
"""Summarise a per-wavefront timeline written by `bench.py --trace-out x.npz` (mhte_trace_begin).

For every traced launch and every role inside it: number of wavefronts, when they started and
ended relative to the launch's first wavefront (min / median / p95 / max, microseconds) and how
long they lived.  The 100 MHz wall clock gives 10 ns resolution.  Output is markdown."""
import sys

import numpy as np

ROLES = {3: "run_dedup", 4: "displacement", 5: "lookup", 6: "work_list", 7: "apply_items",
         8: "apply_ids", 9: "reserve_rows", 0: "(no record)"}


def q(a, p):
  return float(np.percentile(a, p)) if a.size else float("nan")


def main(path):
  z = np.load(path, allow_pickle=True)
  rec = z["records"].astype(np.int64)
  launches = z["launches"]
  print("| launch | kernel | grid x block | role | waves | start min/med/p95/max us | "
        "end med/p95/max us | life med/p95/max us |")
  print("|---|---|---|---|---|---|---|---|")
  for li, (name, grid, block, off) in enumerate(launches):
    grid, block, off = int(grid), int(block), int(off)
    waves = grid * ((block + 63) // 64)
    r = rec[off:off + waves]
    live = r[:, 0] != 0
    if not live.any():
      continue
    t0 = r[live, 0].min()
    span = (r[live, 1].max() - t0) / 100.0
    print("| %d | %s | %d x %d | ALL | %d | span %.2f us | | |" % (li, name, grid, block,
                                                                   int(live.sum()), span))
    for role in sorted(set(r[live, 2].tolist())):
      m = live & (r[:, 2] == role)
      st = (r[m, 0] - t0) / 100.0
      en = (r[m, 1] - t0) / 100.0
      life = en - st
      marks = ""
      if r.shape[1] > 3:
        parts = []
        for k in range(3, r.shape[1]):
          mk = r[m, k]
          ok = mk != 0
          if ok.any():
            rel = (mk[ok] - r[m, 0][ok]) / 100.0
            parts.append("m%d %.2f/%.2f/%.2f (%d)" % (k - 3, q(rel, 50), q(rel, 95), rel.max(), int(ok.sum())))
        marks = " marks since wave start med/p95/max us: " + ", ".join(parts) if parts else ""
      print("| %d | | | %s | %d | %.2f / %.2f / %.2f / %.2f | %.2f / %.2f / %.2f | "
            "%.2f / %.2f / %.2f |%s" % (li, ROLES.get(int(role), str(role)), int(m.sum()),
                                      st.min(), q(st, 50), q(st, 95), st.max(),
                                      q(en, 50), q(en, 95), en.max(),
                                      q(life, 50), q(life, 95), life.max(), marks))
  # gaps between consecutive launches (end of one to first wave of the next)
  prev_end = None
  gaps = []
  for (name, grid, block, off) in launches:
    waves = int(grid) * ((int(block) + 63) // 64)
    r = rec[int(off):int(off) + waves]
    live = r[:, 0] != 0
    if not live.any():
      continue
    if prev_end is not None:
      gaps.append((r[live, 0].min() - prev_end) / 100.0)
    prev_end = r[live, 1].max()
  if gaps:
    print("\ngaps between launches (last wave end -> next first wave start), us: " +
          ", ".join("%.2f" % g for g in gaps))


if __name__ == "__main__":
  main(sys.argv[1])
