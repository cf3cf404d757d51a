import os, sys, time, tempfile, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import torch
import next_rows_bench as N
from monolith_amd import entry
for d in ("/dev/shm", "/tmp"):
  os.environ["MHTE_CKPT_DIR"] = d
  print("dir", d, flush=True)
  N.bench_checkpoint()
