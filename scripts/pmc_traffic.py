#!/usr/bin/env python
"""profiles/pmc_traffic.json from the PMC passes of a GPU round (scripts/gpu_round.sh pmc calib).

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB per dispatch.  MI355X_MICROARCH.md (HBM section):
on gfx950 FETCH_SIZE is exactly 1/2 of the bytes of a wide coalesced streaming read; other access
widths and WRITE_SIZE must be calibrated on a known byte count in the kernel's own access pattern.
scripts/pmc_calibrate.py supplies two launches with known byte counts (random 256-B row gather,
all-distinct table lookup), from which this script derives the counter-per-byte factors

  c_stream = 0.5 (guide), c_rows (random 256-B rows), c_buckets (random 64-B bucket lines), c_write

A step kernel mixes the three read classes, so its read traffic is estimated as
  traffic_read = FETCH_SIZE * sum(alg_i) / sum(alg_i * c_i)
(alg_i = algorithmic bytes of class i: a uniform over-fetch ratio across classes), and
  traffic_write = WRITE_SIZE / c_write.

Usage: python scripts/pmc_traffic.py <round dir with pmc_*.md, calib_*.md, calib.json, bench.json> <tag>"""
import json
import os
import sys


def table(path):
  out = {}
  if not os.path.exists(path):
    return out
  for line in open(path):
    c = [x.strip() for x in line.strip().strip("|").split("|")]
    if len(c) < 5 or c[0] in ("kernel", "---") or c[0].startswith("-"):
      continue
    try:
      out[c[0]] = (int(c[2]), float(c[3]) * 1024.0)  # dispatches, bytes per dispatch
    except ValueError:
      pass
  return out


def find(tab, sub):
  for k, v in tab.items():
    if sub in k:
      return v
  return None


def main():
  d, tag = sys.argv[1], sys.argv[2]
  fetch, write = table(os.path.join(d, "pmc_FETCH_SIZE.md")), table(os.path.join(d, "pmc_WRITE_SIZE.md"))
  cf, cw = table(os.path.join(d, "calib_FETCH_SIZE.md")), table(os.path.join(d, "calib_WRITE_SIZE.md"))
  method = {"c_stream": 0.5}
  c_rows, c_buckets, c_write = 0.5, 1.0, 1.0
  try:
    calib = json.loads(open(os.path.join(d, "calib.json")).read().strip().splitlines()[-1])
    g, l = calib["gather_rows_kernel"], calib["lookup_kernel"]
    fg, fl = find(cf, "gather_rows_kernel")[1], find(cf, "lookup_kernel")[1]
    c_rows = (fg - 0.5 * g["read_stream"]) / g["read_rows"]
    c_buckets = (fl - 0.5 * l["read_stream"] - c_rows * l["read_rows"]) / l["read_buckets"]
    wg, wl = find(cw, "gather_rows_kernel")[1], find(cw, "lookup_kernel")[1]
    c_write = 0.5 * (wg / g["write"] + wl / l["write"])
    method.update({"calibrated": True, "calib_fetch_bytes": {"gather_rows": fg, "lookup": fl},
                   "calib_write_bytes": {"gather_rows": wg, "lookup": wl}, "calib_expected": calib})
  except Exception as e:  # pylint: disable=broad-except
    method.update({"calibrated": False, "why": repr(e)})
  method.update({"c_rows": round(c_rows, 4), "c_buckets": round(c_buckets, 4), "c_write": round(c_write, 4)})

  bench = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
  cfgb = bench["config"]
  B, D, U = cfgb["batch_per_gpu"], cfgb["dim"], cfgb["unique_ids_per_batch"]
  S = D if cfgb["optimizer"] == "adagrad" else 0
  mixes = {
      # (stream, rows, buckets) algorithmic read bytes
      "lookup": (8 * B, 4 * D * U, 128 * U),
      "update": (8 * B, 4 * D * B + (4 * D + 4 * S) * U, 128 * U),
  }
  # (the step kernels also carry the dedup / build roles of the neighbouring batch: their scratch
  # traffic is implementation overhead on top of the algorithmic bytes and shows up here)
  role = {"step_fwd_kernel": "lookup", "lookup_kernel": "lookup", "step_bwd_kernel": "update",
          "sum_apply_kernel": "update"}
  kernels = {}
  for name, r in role.items():
    f, w = find(fetch, name), find(write, name)
    if not f or not w:
      continue
    st, rows, bk = mixes[r]
    scale = (st + rows + bk) / (0.5 * st + c_rows * rows + c_buckets * bk)
    rd = f[1] * scale
    wr = w[1] / c_write
    kernels[name] = {"fetch_size_bytes": round(f[1]), "write_size_bytes": round(w[1]),
                     "dispatches": f[0], "read_bytes_corrected": round(rd),
                     "write_bytes_corrected": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
  # ---- the multi-table step (configs[4]'s shape), when the visit has its PMC passes too
  dl_f, dl_w = table(os.path.join(d, "pmc_dlrm_FETCH_SIZE.md")), table(os.path.join(d, "pmc_dlrm_WRITE_SIZE.md"))
  dl_bench = os.path.join(d, "bench_dlrm26.json")
  workload_dlrm = None
  if dl_f and dl_w and os.path.exists(dl_bench):
    db = json.loads(open(dl_bench).read().strip().splitlines()[-1])
    dims, Bt, Ut = db["config"]["dims"], db["config"]["batch_per_table"], db["config"]["unique_ids_per_batch_mean"]
    workload_dlrm = db["config"]["workload"]

    def cls(nbytes):  # counter per byte of a random access of that width (64-B lines vs >= 128-B rows)
      return c_buckets if nbytes <= 64 else c_rows

    mix = {"lookup": [0.0, 0.0], "update": [0.0, 0.0]}   # [algorithmic read bytes, counter value expected]
    for D_ in dims:
      rowb, gradb = 8 * D_, 4 * D_                       # Adagrad row (w + accumulator), gradient row
      look = [(8 * Ut, 0.5), (128 * Ut, c_buckets), (4 * D_ * Ut, cls(4 * D_))]
      upd = [(24 * Ut, 0.5), (gradb * Bt, cls(gradb)), (rowb * Ut, cls(rowb))]   # hints: no bucket read
      for k, parts in (("lookup", look), ("update", upd)):
        for nbytes, c in parts:
          mix[k][0] += nbytes
          mix[k][1] += nbytes * c
    # (round 3: the lookup and the next batch's run dedup are one launch; the dedup's own reads are
    # the ids, streamed, and its scratch, which lives in L2 — the ids are added to the lookup's mix)
    for D_ in dims:
      mix["lookup"][0] += 8 * Bt
      mix["lookup"][1] += 8 * Bt * 0.5
    for name, r in (("mstep_fwd_dedup_kernel", "lookup"), ("mstep_bwd_kernel", "update")):
      f, w = find(dl_f, name), find(dl_w, name)
      if not f or not w:
        continue
      rd = f[1] * mix[r][0] / mix[r][1]
      wr = w[1] / c_write
      kernels[name] = {"fetch_size_bytes": round(f[1]), "write_size_bytes": round(w[1]),
                       "dispatches": f[0], "read_bytes_corrected": round(rd),
                       "write_bytes_corrected": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
  out = {"source": tag, "method": method, "workload": cfgb.get("workload"),
         "workload_multi_table": workload_dlrm, "kernels": kernels}
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as fp:
    json.dump(out, fp, indent=1)
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main()
