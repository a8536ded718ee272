#!/bin/bash
# (GPU box) cycle counters of the optimise kernel's phases in the reference-arithmetic mode (-DFLORIA_PROF build, thread 0 of each workgroup)
make -C floria_amd/csrc -B EXTRA="-DFLORIA_PROF" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED"; exit 1; }
python scripts/arith_timing.py ${1:-250} 0.04 0 2>&1 | grep -E "^\[prof\]|^arith" | python -c "
import sys,re
for l in sys.stdin:
    if l.startswith('arith'): print(l.strip()); continue
    d={int(a):float(b) for a,b in re.findall(r'(\d+):([0-9.]+)M',l)}
    names=['build','stats0','dist','cand','sort','serial','moves','stats','undo','final','fill','asort','replay+walk','replay(p0)','clear','atomicMin']
    print('  opt phases (Mcyc): '+' '.join('%s %.0f'%(n,d.get(i,0)) for i,n in enumerate(names)))"
make -C floria_amd/csrc -B libfloria_hip.so > /dev/null 2>&1
