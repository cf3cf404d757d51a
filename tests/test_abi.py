"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol the header
declares; without a GPU the product refuses to run (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

from monolith_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "monolith_amd_hash_table.h")


def _declared_symbols():
  src = open(HEADER).read()
  src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
  return sorted(set(re.findall(r"\b(mhte_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
  so = _lib.build_library()
  assert os.path.exists(so)
  L = C.CDLL(so)
  declared = _declared_symbols()
  assert len(declared) >= 30
  missing = [s for s in declared if not hasattr(L, s)]
  assert not missing, missing
  assert set(declared) == set(_lib.EXPORTS), set(declared) ^ set(_lib.EXPORTS)
  assert L.mhte_abi_version() == _lib.ABI_VERSION


def test_library_contains_gfx950_code_object():
  so = _lib.build_library()
  out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", so], capture_output=True,
                       text=True).stdout
  assert ".hip_fatbin" in out
  raw = open(so, "rb").read()
  assert b"gfx950" in raw


def test_header_is_plain_c():
  # the boundary must compile as C: no C++ / torch types in the signatures
  r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", HEADER], capture_output=True,
                     text=True)
  assert r.returncode == 0, r.stderr


def build_c_client(out_dir):
  """gcc (C99, no hipcc, no C++) compiles tests/c_client.c against the public header and links
  libmhte.so: the boundary is usable from plain C."""
  so = _lib.build_library()
  exe = os.path.join(str(out_dir), "mhte_c_client")
  r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                      "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                      os.path.join(ROOT, "tests", "c_client.c"), "-o", exe,
                      "-L" + os.path.dirname(so), "-lmhte", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                      "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"],
                     capture_output=True, text=True)
  assert r.returncode == 0, r.stderr
  return exe


def test_c_client_compiles_and_links(tmp_path):
  exe = build_c_client(tmp_path)
  assert os.path.exists(exe)
  out = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
  for sym in ("mhte_multi_table_create", "mhte_lookup", "mhte_optimize", "mhte_multi_table_save",
              "mhte_multi_table_restore", "mhte_lookup_entry", "mhte_feature_stat"):
    assert sym in out, sym


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_gpu_means_loud_failure():
  L = _lib.lib()
  seg = _lib.SegmentConfig()
  seg.dim_size = 4
  cfg = _lib.TableConfig()
  cfg.name = b"t"
  cfg.n_segments = 1
  cfg.segments = C.pointer(seg)
  cfg.initial_capacity = 1
  h = C.c_void_p()
  st = L.mhte_multi_table_create(C.byref(cfg), 1, 0, b"x", C.byref(h))
  assert st == _lib.MHTE_UNAVAILABLE
  assert b"no CPU fallback" in L.mhte_last_error()
  from monolith_amd.multi_hash_table_ops import MultiHashTable
  from monolith_amd import entry
  cfgs = {"t": entry.make_table_config(
      [entry.CombineAsSegment(4, entry.ZerosInitializer(), entry.SgdOptimizer())])}
  with pytest.raises(_lib.MhteError):
    MultiHashTable.from_configs(cfgs)


def test_product_never_imports_oracle():
  pkg = os.path.join(ROOT, "monolith_amd")
  for dp, _, fns in os.walk(pkg):
    for fn in fns:
      if fn.endswith((".py", ".hip", ".h", ".cc")):
        txt = open(os.path.join(dp, fn)).read()
        assert "import oracle" not in txt and "from oracle" not in txt, fn
        assert "liboracle" not in txt and "libmonolith_ref" not in txt, fn
