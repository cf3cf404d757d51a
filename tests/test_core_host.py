"""CPU suite: the engine's serial cuckoo displacement (mhte_core.h, shared by host and device)
reproduces the REFERENCE map's physical placement for sequential inserts, including BFS
displacement — checked against tests/golden/placement_seq.npz (generated from the reference's own
cuckoohash_map.hpp)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_serial_insert_matches_reference_placement(tmp_path):
  z = np.load(os.path.join(ROOT, "tests", "golden", "placement_seq.npz"))
  exe = str(tmp_path / "core_host")
  subprocess.check_call(["g++", "-O2", "-std=c++17", "-DMHTE_HOST_ONLY",
                         "-I" + os.path.join(ROOT, "monolith_amd", "csrc"),
                         os.path.join(ROOT, "tests", "core_host_driver.cc"), "-o", exe])
  idf = str(tmp_path / "ids.bin")
  z["ids"].astype(np.int64).tofile(idf)
  out = subprocess.run([exe, "10", idf], capture_output=True, text=True, check=True).stdout
  got = np.array([[int(a) for a in line.split()] for line in out.strip().splitlines()],
                 dtype=np.int64)
  np.testing.assert_array_equal(got[:, 0], z["dump_ids"])
  np.testing.assert_array_equal(got[:, 1], z["dump_pos"])
