#!/bin/bash
# (GPU box) cycle counters of the optimise kernel's phases in the reference-arithmetic mode (-DFLORIA_PROF build, thread 0 of each workgroup)
# (a library prebuilt on the build host as floria_amd/csrc/variants/libfloria_hip_prof.so is used if there is one: no rebuild on the GPU box)
D=floria_amd/csrc
if [ -f $D/variants/libfloria_hip_prof.so ]; then cp $D/libfloria_hip.so $D/variants/libfloria_hip_base.so; cp $D/variants/libfloria_hip_prof.so $D/libfloria_hip.so
else make -C $D -B EXTRA="-DFLORIA_PROF" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED"; exit 1; }; fi
python scripts/arith_timing.py ${1:-250} 0.04 0 2>&1 | grep -E "^\[prof\]|^arith" | python -c "
import sys,re
for l in sys.stdin:
    if l.startswith('arith'): print(l.strip()); continue
    d={int(a):float(b) for a,b in re.findall(r'(\d+):([0-9.]+)M',l)}
    names=['build','stats0','dist','cand','sort','serial','moves','stats','undo','final','fill','asort','replay+walk','replay(p0)','clear','atomicMin']
    print('  opt phases (Mcyc): '+' '.join('%s %.0f'%(n,d.get(i,0)) for i,n in enumerate(names)))"
if [ -f $D/variants/libfloria_hip_base.so ]; then cp $D/variants/libfloria_hip_base.so $D/libfloria_hip.so; else make -C $D -B libfloria_hip.so > /dev/null 2>&1; fi
