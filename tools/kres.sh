#!/bin/bash
# kernel resource table of one source file: tools/kres.sh rasterize_mfma.hip [extra flags]
cd "$(dirname "$0")/../pipeline-pointcloud_amd/csrc"
F=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast "$@" -Rpass-analysis=kernel-resource-usage -c $F -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d={}; continue
    m=re.search(r'remark:\s+([A-Za-z ]+)(?: \[[^\]]*\])?: (\d+)',l)
    if m and cur: d[m.group(1).strip()]=int(m.group(2))
    if 'LDS Size' in l and cur:
        print(f\"{cur[:90]:90s} vgpr {d.get('VGPRs',0):4d} occ {d.get('Occupancy',0)} lds {d.get('LDS Size',0):6d} spill {d.get('VGPRs Spill',0)}\")
"
