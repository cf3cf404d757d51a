#!/bin/bash
# kernel resource usage (VGPRs, spills, LDS, occupancy) of the development build's step kernels
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -DMHTE_DEV_FAST "$@" -Rpass-analysis=kernel-resource-usage -o /tmp/dev_res.so monolith_amd/csrc/mhte.hip 2> /tmp/res.txt
python3 - "${KRES_FILTER:-step_bwd|step_fwd|rd_build|rd_probe}" <<'PY'
import re,sys
t=open('/tmp/res.txt').read()
pat=re.compile(sys.argv[1])
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name=b.split(' ')[0]
    if not pat.search(name): continue
    f=lambda k:(re.search(k+r": (\d+)", b) or [0,'?'])[1]
    print(name[:70].ljust(70), 'VGPR',f('    VGPRs'),'vspill',f('VGPRs Spill'),'sspill',f('SGPRs Spill'),'scratch',f(r'ScratchSize \[bytes/lane\]'),'LDS',f(r'LDS Size \[bytes/block\]'),'occ',f(r'Occupancy \[waves/SIMD\]'))
PY
