#!/bin/bash
# usage: scripts/ru.sh "<extra flags>" <substring of the mangled kernel name>  -> registers / spills / occupancy of the matching kernels (hipcc remarks)
cd "$(dirname "$0")/../floria_amd/csrc"
/opt/rocm/bin/hipcc $1 -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c -o /dev/null floria_hip.hip 2>&1 | python3 -c "
import sys,re
cur=None; d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)',l)
    if m and cur: d[cur][m.group(1).strip()]=int(m.group(2))
for k,v in d.items():
    if '$2' in k: print(k[:72], 'V',v.get('VGPRs'),'S',v.get('TotalSGPRs'),'Sspill',v.get('SGPRs Spill'),'Vspill',v.get('VGPRs Spill'),'occ',v.get('Occupancy [waves/SIMD]'),'scr',v.get('ScratchSize [bytes/lane]'))
"
