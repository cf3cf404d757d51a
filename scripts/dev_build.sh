#!/bin/bash
# Development build of the library: -DMHTE_DEV_FAST instantiates the dim-64 float4 shape only
# (G = 16, VEC = 4), so a kernel change compiles in under a minute instead of three.  The result
# (monolith_amd/libmhte_dev.so, or $1) is for bench A/B runs through MHTE_LIBRARY — never shipped.
OUT=${1:-monolith_amd/libmhte_dev.so}; shift || true
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -DMHTE_DEV_FAST "$@" \
  -o "$OUT" monolith_amd/csrc/mhte.hip
